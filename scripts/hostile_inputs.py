#!/usr/bin/env python3
"""Hostile inputs (round 5): infinities, NaNs, 1e300s and denormals in controllers and draws, every kernel route, N = 2 ... 24.
Every launch must RETURN (all device loops are capped), the hostile samples may hold anything (NaN expected), and the clean
samples of the same launch - same tiles - must still agree with the oracle to 1e-10.  Run under `timeout`.
usage: python3 scripts/hostile_inputs.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
be = importlib.import_module("code-robchar_amd.backend")
from oracle import robchar_oracle as orc
rng = np.random.default_rng(2024)
bad_vals = [np.inf, -np.inf, np.nan, 1e300, -1e300, 1e-310, 1e200]
worst, nlaunch, t0 = 0.0, 0, time.time()
for N in (2, 3, 5, 7, 10, 13, 16, 20, 24):
    C, K = 6, 200
    ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    ctrl[2, N], ctrl[3, N] = 0.0, 1e-300                   # LEGITIMATE edge cases among the clean rows: T = 0 (F = [in == out]) and a denormal-scale T
    clean_ctrl, clean_draws = ctrl.copy(), draws.copy()
    # hostile DRAWS: 12 samples per controller get one poisoned entry each
    hostile = np.zeros((C, K), dtype=bool)
    for c in range(C):
        for j, k in enumerate(rng.choice(K, 12, replace=False)):
            draws[c, k, rng.integers(0, N), rng.integers(0, 3)] = bad_vals[j % len(bad_vals)]
            hostile[c, k] = True
    # hostile CONTROLLERS (rows 4, 5): T = inf / a 1e300 bias - every sample of those rows is hostile
    ctrl[4, N] = np.inf
    ctrl[5, 0] = 1e300
    hostile[4:] = True
    for ring in (False, True):
        if ring and (N > 16 or N < 3):
            continue
        a, b = 0, (N - 1 if not ring else N // 2)
        want = orc.fidelity_eigh(clean_ctrl[:4], clean_draws[:4], N, a, b, ring=ring)
        kernels = (("auto", "ring_hh", "jacobi") if ring else (("auto", "tridiag_ql", "tridiag_adj") + (("jacobi", "expm") if N <= 16 else ())))
        for kern in kernels:
            got = np.asarray(be.mc_fidelity(ctrl, draws, N, a, b, ring=ring, kernel=kern))
            nlaunch += 1
            ok = ~hostile[:4]
            e = float(np.abs(got[:4][ok] - want[ok]).max())
            worst = max(worst, e)
            assert e < 1e-10, (N, ring, kern, e)
            assert got.shape == (C, K)
            print(f"N={N:2d} ring={int(ring)} {kern:12s}: clean samples max|dF| {e:.1e}; hostile samples: {int(np.isnan(got[hostile]).sum())} NaN of {int(hostile.sum())}, "
                  f"finite ones in [{np.nanmin(got[hostile]):.2g}, {np.nanmax(got[hostile]):.2g}]", flush=True)
print(f"{nlaunch} launches in {time.time() - t0:.1f} s, clean samples worst {worst:.1e}")
print("ok")
