#!/bin/bash
# round 4, call 29: product-level fuzz of MCDataSim (scripts/fuzz_mcdatasim.py): GPU vs the oracle-backed host route, host-drawn vs
# device-continued legacy stream, fused vs draw-tensor Philox route
R=$PWD; OUT=$R/gpurun_out/r4al; mkdir -p $OUT
SEED=1 NCFG=3 timeout -k 10 200 python scripts/fuzz_mcdatasim.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_mcdatasim.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
for r in 10:19 20:29; do
  SEED=$r NCFG=20 timeout -k 10 420 python scripts/fuzz_mcdatasim.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz_mcdatasim.txt
  echo "block $r rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz_mcdatasim.txt
done
