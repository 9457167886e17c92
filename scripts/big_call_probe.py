#!/usr/bin/env python3
"""One very large call through the blocking entry (round 5): C = 4 000 controllers x K = 250 000 counter-based draws = 1e9
evaluations, 2.1e10 draws (element offsets beyond 2^32, chunked through the library's workspace, rows of 250 000 values through
the long-row reduction): the metric rows of a few controllers must equal, bit for bit, what the same controllers give when they are
computed alone from the matching stream offset; a second shape with K NOT a multiple of 64 and general in/out.
usage: python3 scripts/big_call_probe.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
be = importlib.import_module("code-robchar_amd.backend")
rng = np.random.default_rng(77)
for (N, a, b, C, K) in ((7, 0, 6, 4000, 250000), (5, 1, 3, 9000, 100003)):
    ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
    t0 = time.time()
    big = be.mc_metrics_sharded(ctrl, K, N, a, b, seed=5, offset=123, sigma=0.05, devices=[0], dkw_eps=0.003)
    dt = time.time() - t0
    G = 3 * N
    for row in (0, 1, C // 3, C - 2, C - 1):
        one = be.mc_metrics_sharded(ctrl[row:row + 1], K, N, a, b, seed=5, offset=123 + row * K * G, sigma=0.05, devices=[0], dkw_eps=0.003)
        for k in ("rim1", "std", "min", "q"):
            assert np.array_equal(big[k][..., row], one[k][..., 0]), (N, row, k, big[k][..., row], one[k][..., 0])
    assert np.isfinite(big["rim1"]).all() and (big["rim1"] >= 0).all() and (big["rim1"] <= 1).all()
    print(f"N={N} {a}->{b} C={C} K={K}: {C * K / 1e9:.2f}e9 evaluations in {dt:.2f} s ({C * K / dt / 1e9:.2f}e9 per s through the blocking entry), "
          f"last element offset {123 + C * K * G:.3e}; rows 0, 1, {C // 3}, {C - 2}, {C - 1} identical to their stand-alone runs", flush=True)
print("ok")
