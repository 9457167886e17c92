#!/bin/bash
# round-3 GPU session 1: A/B of (old = zero-initialised staging registers) / (base) / (w4 = N=10 at 4 waves/SIMD, spills)
set -e
OUT=gpurun_out/r3b; mkdir -p $OUT
export REPS=300
for rep in 1 2; do
  SHAPES=7:100:10000,10:100:10000 scripts/run_variants.sh scripts/ubench/lib_r3_old.so scripts/ubench/lib_r3_base.so >> $OUT/ab_n7.txt 2>&1
  SHAPES=10:100:10000 KBENCH_ARGS=--xxz scripts/run_variants.sh scripts/ubench/lib_r3_base.so scripts/ubench/lib_r3_w4.so >> $OUT/ab_n10.txt 2>&1
done
cat $OUT/ab_n7.txt $OUT/ab_n10.txt
scripts/pmc_quick.sh r3b/pmc_base_n7 scripts/ubench/lib_r3_base.so 7:100:10000 > $OUT/pmc_n7.txt 2>&1
scripts/pmc_quick.sh r3b/pmc_base_n10 scripts/ubench/lib_r3_base.so 10:100:10000 --xxz > $OUT/pmc_n10.txt 2>&1
cat $OUT/pmc_n7.txt $OUT/pmc_n10.txt
