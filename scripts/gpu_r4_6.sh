#!/bin/bash
# round 4, call 6: what the stepping path costs - code present but never run (stepnotrun), stepping runs but the tile-wide fp64
# QL never does (fbnotrun), against the shipped build (few) and the build without either (nostep); two fuzz campaign blocks
R=$PWD; OUT=$R/gpurun_out/r4f; mkdir -p $OUT/dump
kb() { timeout -k 10 200 python scripts/kbench.py --reps 300 "$@" 2>&1 | grep "N="; }
for round in 1 2; do
  for v in few nostep stepnotrun fbnotrun; do
    export ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so
    echo "== $v" >> $OUT/ab_step_cost.txt
    kb --shapes 7:100:10000 >> $OUT/ab_step_cost.txt
    kb --shapes 10:100:10000 --xxz >> $OUT/ab_step_cost.txt
    kb --shapes 7:100:10000 --shipped >> $OUT/ab_step_cost.txt
  done
done
unset ROBCHAR_HIP_LIB
cat $OUT/ab_step_cost.txt
for r in 4000:4099 4100:4199; do
  FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1e-11 SEED=$r NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
done
ls $OUT/dump
