#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep over wide parameter ranges (noise up to 0.5, times up to 100, biases up to
+-100, XXZ offsets, every in/out pair class, all chain kernels).  Prints the worst error per kernel; exits non-zero
above 1e-10.  Development aid - the pytest suite holds the fixed cases."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
be = importlib.import_module("code-robchar_amd.backend")
from oracle import robchar_oracle as orc
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
worst = {}
t0 = time.time()
ncfg = int(os.environ.get("NCFG", "150"))
for it in range(ncfg):
    N = int(rng.integers(2, 17))
    C, K = int(rng.integers(1, 6)), int(rng.integers(1, 700))
    amp = float(rng.choice([1.0, 10.0, 100.0]))
    sig = float(rng.choice([0.0, 1e-3, 0.05, 0.2, 0.5]))
    ctrl = np.empty((C, N + 1))
    ctrl[:, :N] = rng.uniform(-amp, amp, (C, N))
    ctrl[:, N] = rng.uniform(0.0, float(rng.choice([1.0, 30.0, 100.0])), C) * rng.choice([-1, 1], C)
    draws = sig * rng.standard_normal((C, K, N, 3))
    h0 = orc.xxz_delta(N) if rng.random() < 0.3 else None
    a, b = int(rng.integers(0, N)), int(rng.integers(0, N))
    if rng.random() < 0.4:
        a, b = 0, N - 1
    want = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0)
    for kern in ("auto", "tridiag_ql", "tridiag_adj"):
        got = be.mc_fidelity(ctrl, draws, N, a, b, h0_diag=h0, kernel=kern)
        e = float(np.abs(got - want).max())
        if e > worst.get(kern, (0,))[0]:
            worst[kern] = (e, dict(N=N, C=C, K=K, amp=amp, sig=sig, a=a, b=b, xxz=h0 is not None, Tmax=float(np.abs(ctrl[:, N]).max())))
print(f"{ncfg} configurations in {time.time() - t0:.1f}s; general-path tiles seen: {be.general_path_tiles()}")
bad = False
for k, (e, cfg) in worst.items():
    print(f"{k:12s} worst |dF| = {e:.2e} at {cfg}")
    bad |= e > 1e-10
sys.exit(1 if bad else 0)
