#!/bin/bash
# round 4, call 18: the fidelity kernel with the Philox draws generated inside it: parity tests (bit-identical to the two-kernel
# route), kernel time against philox_normal_kernel + mc_fid_chain_kernel at BASELINE config 4's size
R=$PWD; OUT=$R/gpurun_out/r4s; mkdir -p $OUT
python -m pytest tests/test_gpu_round4.py -m gpu -q -x -k "philox" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -15
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/fused_bench.txt
import importlib, time, numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
rng = np.random.default_rng(0)
def ev(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (N, a, b, C, K) in ((7, 0, 3, 1000, 100000), (7, 0, 6, 1000, 100000), (5, 0, 2, 11000, 100), (10, 0, 9, 100, 100000)):
    x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
    ct = torch.from_numpy(x).cuda()
    d = torch.empty((C, K, N, 3), dtype=torch.float64, device="cuda")
    out = torch.empty((C, K), dtype=torch.float64, device="cuda")
    tg = ev(lambda: be.philox_normal(d.shape, 11, scale=0.05, out=d))
    tf = ev(lambda: be.mc_fidelity(ct, d, N, a, b, out=out))
    tb = ev(lambda: (be.philox_normal(d.shape, 11, scale=0.05, out=d), be.mc_fidelity(ct, d, N, a, b, out=out)))
    tx = ev(lambda: be.mc_fidelity_philox(ct, K, N, a, b, 11, sigma=0.05, out=out))
    print(f"N={N} {a}->{b} {C} x {K}: generator {tg:.3f} ms + fidelity {tf:.3f} ms = two kernels {tb:.3f} ms;  fused {tx:.3f} ms  ({100 * (tx / tb - 1):+.1f} %)")
    del d
PY
