#!/bin/bash
# rational QL (tqlrat) vs rotations for the fp32 starting values, x Aberth / Halley first step: same-box A/B
OUT=gpurun_out/r3t; mkdir -p $OUT
LIBS="scripts/ubench/lib_halley.so scripts/ubench/lib_aberth_rot.so scripts/ubench/lib_halley_rat.so code-robchar_amd/csrc/librobchar_hip.so"
for rep in 1 2; do
  REPS=300 SHAPES=5:100:10000,7:100:10000,10:100:10000,13:100:10000 scripts/run_variants.sh $LIBS
done 2>&1 | tee $OUT/ab_chain.txt
REPS=300 SHAPES=10:100:10000 KBENCH_ARGS=--xxz scripts/run_variants.sh $LIBS 2>&1 | tee $OUT/ab_c5.txt
REPS=200 SHAPES=5:100:10000,7:100:10000,10:100:10000 KBENCH_ARGS=--ring scripts/run_variants.sh $LIBS 2>&1 | tee $OUT/ab_ring.txt
REPS=20 SHAPES=7:1000:100000 KBENCH_ARGS="--out mid --device-draws" scripts/run_variants.sh $LIBS 2>&1 | tee $OUT/ab_c4.txt
scripts/pmc_quick.sh r3t_7 code-robchar_amd/csrc/librobchar_hip.so 7:100:10000 2>&1 | tail -2
scripts/pmc_quick.sh r3t_10 code-robchar_amd/csrc/librobchar_hip.so 10:100:10000 --xxz 2>&1 | tail -2
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -6
for s in 41 42 43 44; do SEED=$s NCFG=150 timeout -k 10 300 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "worst\|configurations" ; done | tee $OUT/fuzz.txt
