#!/bin/bash
# round 4, call 17: fuzz campaign on the round's last build, seeds 4800-5399 (6 x 100 seeds x 150 configurations = 90 000),
# each block with 100 random cases of the directional fidelity entry
R=$PWD; OUT=$R/gpurun_out/r4r; mkdir -p $OUT/dump
for r in 4800:4899 4900:4999 5000:5099 5100:5199 5200:5299 5300:5399; do
  FUZZ_DIR=100 FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1.3e-11 SEED=$r NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
done
ls $OUT/dump
