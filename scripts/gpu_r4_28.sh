#!/bin/bash
# round 4, call 28: two more edges of the fast path - chain sizes beyond the register-resident kernels (N = 17 .. 32: the LDS
# any-N kernel) and draw tensors whose rows are not 16-byte aligned (N and K odd: the staging path without LDS-DMA)
R=$PWD; OUT=$R/gpurun_out/r4ak; mkdir -p $OUT
echo "== chain, N = 16 .. 32 (end-to-end pairs)" | tee $OUT/edges.txt
timeout -k 10 500 python scripts/kbench.py --reps 20 --shapes 16:100:10000,17:100:10000,20:100:10000,24:100:10000,32:100:10000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/edges.txt
echo "== N = 7, K = 10000 / 10001 (rows of 210 000 / 210 021 doubles: 16-byte aligned / not)" | tee -a $OUT/edges.txt
timeout -k 10 200 python scripts/kbench.py --reps 200 --shapes 7:100:10000,7:100:10001,5:100:10001,9:100:10001 2>&1 | grep -v amdgpu.ids | tee -a $OUT/edges.txt
