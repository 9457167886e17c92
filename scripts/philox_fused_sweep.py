#!/usr/bin/env python3
"""Fused Philox fidelity kernel against the two-kernel route (philox_normal_kernel + mc_fid_chain_kernel) at every size the
fused kernel covers (N = 2 .. 16, 100 x 10 000, end-to-end and general adjugate weights): time per 1e6 evaluations of both, and
bit-identity.  Development aid (decides `philox_fused_supported`)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
dev = torch.device("cuda", 0)
C, K, reps = 100, 10000, 60
def timed(f):
    for _ in range(3): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
for N in range(2, 17):
    rng = np.random.default_rng(N)
    x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
    ct = torch.from_numpy(x).to(dev)
    d = torch.empty((C, K, N, 3), dtype=torch.float64, device=dev)
    o1 = torch.empty((C, K), dtype=torch.float64, device=dev); o2 = torch.empty_like(o1)
    for (a, b) in ((0, N - 1), (0, N // 2)):
        def two():
            be.philox_normal(d.shape, 99, scale=0.05, offset=7, out=d)
            be.mc_fidelity(ct, d, N, a, b, out=o1)
        def fused():
            be.mc_fidelity_philox(ct, K, N, a, b, 99, offset=7, sigma=0.05, out=o2)
        t2, tf = timed(two), timed(fused)
        print(f"N={N:2d} {a}->{b:2d}: two kernels {t2:7.1f} us  fused {tf:7.1f} us  ratio {tf / t2:.2f}  identical {bool(torch.equal(o1, o2))}")
