#!/bin/bash
# round 4, call 21: the fuzz script's auxiliary blocks (legacy stream, directional draws, directional entry, fused Philox kernel)
# under four further seeds of their own (FUZZ_AUX_SEED; the campaign so far ran them with one fixed seed), 300 fused / 100
# directional cases each, beside a short main block
R=$PWD; OUT=$R/gpurun_out/r4z; mkdir -p $OUT/dump
for a in 1 2 3 4; do
  FUZZ_AUX_SEED=$a FUZZ_FUSED=300 FUZZ_DIR=100 FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1.3e-11 SEED=$((5500 + 10 * a)):$((5509 + 10 * a)) NCFG=150 timeout -k 10 280 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
  echo "aux seed $a rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz.txt
done
ls $OUT/dump
