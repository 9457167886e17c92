#!/bin/bash
OUT=gpurun_out/r3l; mkdir -p $OUT
for seed in 34 44; do
  FUZZ_DUMP_ABOVE=3e-11 FUZZ_DUMP=$OUT SEED=$seed NCFG=2500 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "^auto"
done
ls $OUT
