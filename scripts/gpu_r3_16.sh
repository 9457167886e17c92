#!/bin/bash
OUT=gpurun_out/r3q; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $OUT/pytest.log; echo "pytest rc=${PIPESTATUS[0]}"; grep "repaired\|passed\|failed\|Error" $OUT/pytest.log | tail -8
python3 scripts/kbench.py --ring --reps 200 --shapes 5:100:10000,7:100:10000,10:100:10000 2>&1 | grep -v amdgpu.ids | tee $OUT/ring_kbench.txt
scripts/collect_profiles_cfg.sh r3q_ring 7:100:10000 --ring > /dev/null 2>&1
head -4 gpurun_out/r3q_ring/kt/p_kernel_stats.csv | cut -c1-150
for seed in 72 77 80 81 91 92; do
  FUZZ_DUMP_ABOVE=3e-11 FUZZ_DUMP=$OUT SEED=$seed NCFG=2500 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "^auto\|^ring:auto\|^tridiag_adj" | tee -a $OUT/fuzz.txt
  echo "seed $seed rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz.txt
done
