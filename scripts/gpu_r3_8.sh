#!/bin/bash
OUT=gpurun_out/r3j; mkdir -p $OUT
for seed in 33 34 36; do
  FUZZ_DUMP=$OUT SEED=$seed NCFG=2500 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "auto\|tridiag" 
done
ls -la $OUT
