#!/bin/bash
# second adversarial fuzz campaign on the final build (after the Aberth first step, the 512-block sub-streams and the
# level batching): 4 x 100 seeds x 150 configurations
OUT=gpurun_out/r3y; mkdir -p $OUT/dump
for r in 2000:2099 2100:2199 2200:2299 2300:2399; do
  FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=5e-11 SEED=$r NCFG=150 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
done
ls $OUT/dump | head
