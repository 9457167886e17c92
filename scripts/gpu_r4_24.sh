#!/bin/bash
# round 4, call 24: the chain kernels over the whole size range of the fast path (N = 2 .. 16, 100 x 10 000, sigma 0.05, uniform
# random controllers as in the bench): end-to-end weights (0 -> N-1) and general adjugate weights (0 -> N/2); the same for the
# fidelity kernel with the draws generated inside (philox_fused_probe covers N = 7 / 10 only)
R=$PWD; OUT=$R/gpurun_out/r4ad; mkdir -p $OUT
S=$(python3 -c "print(','.join(f'{n}:100:10000' for n in range(2, 17)))")
echo "== end-to-end weights (in = 0, out = N - 1)" | tee $OUT/size_sweep.txt
timeout -k 10 400 python scripts/kbench.py --reps 200 --shapes $S 2>&1 | grep -v amdgpu.ids | tee -a $OUT/size_sweep.txt
echo "== general adjugate weights (in = 0, out = N / 2)" | tee -a $OUT/size_sweep.txt
timeout -k 10 400 python scripts/kbench.py --reps 200 --out mid --shapes $S 2>&1 | grep -v amdgpu.ids | tee -a $OUT/size_sweep.txt
