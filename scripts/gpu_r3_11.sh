#!/bin/bash
OUT=gpurun_out/r3m; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $OUT/pytest.log; echo "pytest rc=${PIPESTATUS[0]}"; grep "repaired\|passed\|failed" $OUT/pytest.log | tail -8
export REPS=200
for rep in 1 2; do
  SHAPES=7:1000:10000 KBENCH_ARGS="--out mid --device-draws" scripts/run_variants.sh scripts/ubench/lib_r3_new.so scripts/ubench/lib_r3_new2.so 2>&1 | grep -v amdgpu | tee -a $OUT/ab_c4.txt
  SHAPES=7:100:10000,10:100:10000 scripts/run_variants.sh scripts/ubench/lib_r3_new.so scripts/ubench/lib_r3_new2.so 2>&1 | grep -v amdgpu | tee -a $OUT/ab_c4.txt
done
for seed in 34 44 51 52 53 54; do
  FUZZ_DUMP_ABOVE=3e-11 FUZZ_DUMP=$OUT SEED=$seed NCFG=2500 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "^auto\|^tridiag_adj\|^ring:auto" | tee -a $OUT/fuzz.txt
  echo "seed $seed rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz.txt
done
ls $OUT
