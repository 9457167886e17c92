#!/bin/bash
# round 4, call 30: product-level fuzz of MCDataSim, 2000 further configurations (8 % of them with rows of 2049 .. 5000 draws)
R=$PWD; OUT=$R/gpurun_out/r4am; mkdir -p $OUT
for r in 100:109 110:119 120:129 130:139 140:149 150:159 160:169 170:179 180:189 190:199; do
  SEED=$r NCFG=20 timeout -k 10 420 python scripts/fuzz_mcdatasim.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz_mcdatasim.txt
  echo "block $r rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz_mcdatasim.txt
done
