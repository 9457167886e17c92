#!/bin/bash
# round 4, call 20: fuzz campaign on the final build, seeds 5200-5499 (3 x 100 seeds x 150 configurations), each block with 100
# random cases of the fused Philox fidelity kernel (bit-identical to the draw-tensor route) and 60 of the directional entry
R=$PWD; OUT=$R/gpurun_out/r4y; mkdir -p $OUT/dump
for r in 5200:5299 5300:5399 5400:5499; do
  FUZZ_FUSED=100 FUZZ_DIR=60 FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1.3e-11 SEED=$r NCFG=150 timeout -k 10 380 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
  echo "block $r rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz.txt
done
ls $OUT/dump
