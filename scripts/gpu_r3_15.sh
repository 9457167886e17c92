#!/bin/bash
OUT=gpurun_out/r3p; mkdir -p $OUT
for seed in 71 72 73 74 75 76 77 78 79 80 81 82; do
  FUZZ_DUMP_ABOVE=3e-11 FUZZ_DUMP=$OUT SEED=$seed NCFG=2500 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "worst\|configurations" | tee -a $OUT/fuzz.txt
  echo "seed $seed rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz.txt
done
ls $OUT
