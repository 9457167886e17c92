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
    # ring topology: the lane-per-sample Householder + QL kernel (N = 3..10) and the Jacobi kernel
    if N >= 3:
        want_r = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0, ring=True)
        for kern in (("ring_hh", "jacobi") if N <= 10 else ("jacobi",)):
            got = be.mc_fidelity(ctrl, draws, N, a, b, h0_diag=h0, ring=True, kernel=kern)
            e = float(np.abs(got - want_r).max())
            if e > worst.get(kern, (0,))[0]:
                worst[kern] = (e, dict(N=N, C=C, K=K, amp=amp, sig=sig, a=a, b=b, xxz=h0 is not None, Tmax=float(np.abs(ctrl[:, N]).max())))
print(f"{ncfg} configurations in {time.time() - t0:.1f}s; general-path tiles seen: {be.general_path_tiles()}")
# the reference's RNG on the device against NumPy itself: random stream positions, sizes, period patterns
rng2 = np.random.default_rng(99)
for it in range(40):
    np.random.seed(int(rng2.integers(0, 2 ** 31)))
    np.random.standard_normal(int(rng2.integers(0, 2000)))          # arbitrary position, cached normal or not
    periods, period = int(rng2.integers(1, 40)), int(rng2.integers(1, 5000))
    skip = int(rng2.integers(0, min(period, 3) + 1))
    scales = rng2.uniform(0, 0.3, periods)
    st0 = np.random.get_state()
    got = be.legacy_normal_periods(periods, period, skip, scales).cpu().numpy()
    st1 = np.random.get_state()
    np.random.set_state(st0)
    want = np.stack([np.random.normal(scale=scales[p], size=period)[skip:] for p in range(periods)])
    st2 = np.random.get_state()
    assert np.array_equal(st1[1], st2[1]) and st1[2:] == st2[2:], "generator state differs from NumPy's"
    if want.size:
        assert np.abs(got - want).max() <= 8 * 2.2e-16 * max(np.abs(want).max(), 1e-300), "normals differ"
print("legacy stream: 40 random (position, periods, period, skip) cases identical in state, <= 2 ulp in value")
bad = False
for k, (e, cfg) in worst.items():
    print(f"{k:12s} worst |dF| = {e:.2e} at {cfg}")
    bad |= e > 1e-10
sys.exit(1 if bad else 0)
