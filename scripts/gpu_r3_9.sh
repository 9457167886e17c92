#!/bin/bash
OUT=gpurun_out/r3k; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for seed in 31 32 33 34 35 36 41 42 43 44; do
  FUZZ_DUMP=$OUT SEED=$seed NCFG=2500 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "worst\|configurations" | tee -a $OUT/fuzz.txt
  echo "seed $seed rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz.txt
done
ls $OUT | head -30
