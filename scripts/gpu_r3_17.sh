#!/bin/bash
OUT=gpurun_out/r3r; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_mcsim.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -4
python3 scripts/kbench.py --ring --reps 300 --shapes 5:100:10000,7:100:10000,10:100:10000,7:1000:10000 2>&1 | grep -v amdgpu.ids | tee $OUT/ring_kbench.txt
python3 scripts/kbench.py --ring --kernel ring_hh --reps 100 --shapes 7:100:10000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ring_kbench.txt
scripts/collect_profiles_cfg.sh r3r_ring 7:100:10000 --ring > /dev/null 2>&1
head -5 gpurun_out/r3r_ring/kt/p_kernel_stats.csv | cut -c1-230
