#!/bin/bash
OUT=gpurun_out/r3o; mkdir -p $OUT
export REPS=300
for rep in 1 2; do
  SHAPES=5:100:10000,7:100:10000 scripts/run_variants.sh scripts/ubench/lib_acc_1e-14.so scripts/ubench/lib_acc_1e-13.so 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt
  SHAPES=10:100:10000 KBENCH_ARGS=--xxz scripts/run_variants.sh scripts/ubench/lib_acc_1e-14.so scripts/ubench/lib_acc_1e-13.so 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt
done
for v in 1e-14 1e-13; do echo "== polish rate $v"; ROBCHAR_HIP_LIB=$PWD/scripts/ubench/lib_acc_$v.so python scripts/polish_rate.py 2>&1 | grep -v amdgpu; done | tee -a $OUT/ab.txt
