#!/bin/bash
# round-3 GPU session 2: full GPU suite on the new build, A/B base vs new (mixed_refine + repair path), instruction counts
OUT=gpurun_out/r3c; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
export REPS=300
for rep in 1 2; do
  SHAPES=5:100:10000,7:100:10000 scripts/run_variants.sh scripts/ubench/lib_r3_base.so scripts/ubench/lib_r3_new.so >> $OUT/ab.txt 2>&1
  SHAPES=10:100:10000 KBENCH_ARGS=--xxz scripts/run_variants.sh scripts/ubench/lib_r3_base.so scripts/ubench/lib_r3_new.so >> $OUT/ab.txt 2>&1
  SHAPES=7:1000:10000 KBENCH_ARGS="--out mid --device-draws" scripts/run_variants.sh scripts/ubench/lib_r3_base.so scripts/ubench/lib_r3_new.so >> $OUT/ab.txt 2>&1
done
cat $OUT/ab.txt
scripts/pmc_quick.sh r3c/pmc_new_n7 scripts/ubench/lib_r3_new.so 7:100:10000 > $OUT/pmc_n7.txt 2>&1
scripts/pmc_quick.sh r3c/pmc_new_n10 scripts/ubench/lib_r3_new.so 10:100:10000 --xxz > $OUT/pmc_n10.txt 2>&1
cat $OUT/pmc_n7.txt $OUT/pmc_n10.txt
ROBCHAR_HIP_LIB=$PWD/scripts/ubench/lib_r3_new.so python scripts/polish_rate.py > $OUT/polish_rate.txt 2>&1; cat $OUT/polish_rate.txt
