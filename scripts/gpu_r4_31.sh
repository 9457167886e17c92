#!/bin/bash
# round 4, call 31: product-level fuzz with the single-process multi-device route in it (the same device listed 1 / 2 / 3 times)
R=$PWD; OUT=$R/gpurun_out/r4an; mkdir -p $OUT
export RC_ALLOW_DUPLICATE_DEVICES=1
SEED=1 NCFG=3 timeout -k 10 200 python scripts/fuzz_mcdatasim.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_mcdatasim.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
for r in 200:209 210:219 220:229; do
  SEED=$r NCFG=20 timeout -k 10 420 python scripts/fuzz_mcdatasim.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz_mcdatasim.txt
  echo "block $r rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz_mcdatasim.txt
done
