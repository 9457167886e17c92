#!/usr/bin/env python3
"""How far is the fused Philox fidelity kernel from the two-kernel route bit for bit?  (development aid)"""
import importlib, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
be = importlib.import_module("code-robchar_amd.backend")
rng = np.random.default_rng(1)
dev = torch.device("cuda", 0)
for sigma in (0.0, 0.05):
    for N in (2, 5, 7, 10):
        for (a, b) in ((0, N - 1), (N // 2, 0)):
            C, K, off = 6, 1000, 7
            x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
            ct = torch.from_numpy(x).to(dev)
            d = be.philox_normal((C, K, N, 3), 99, scale=sigma, offset=off, device=dev, as_torch=True)
            w = be.mc_fidelity(ct, d, N, a, b)
            g = be.mc_fidelity_philox(ct, K, N, a, b, 99, offset=off, sigma=sigma)
            diff = (g - w).abs()
            bad = torch.nonzero(diff > 0)
            lanes = sorted(set(int(k) % 64 for _, k in bad[:200].tolist()))
            print(f"sigma={sigma} N={N} {a}->{b}: differing {int((diff > 0).sum())} of {C * K}, max |diff| {float(diff.max()):.2e}; lanes {lanes[:20]}")
