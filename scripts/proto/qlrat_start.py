#!/usr/bin/env python3
"""Prototype (numpy, fp32, the kernel's lock-step semantics: a 64-sample tile sweeps until ALL its samples have converged):
the fp32 starting values of the mixed-precision path from (a) the implicit QL rotation of tridiag_ql_f32 and (b) the
rational QL of Reinsch (EISPACK tqlrat: squares of the couplings, two reciprocals, no square root), (c) Pal-Walker-Kahan.
Reports sweeps per tile, inner steps per tile, the error of the starts and the share of tiles whose every sample passes the
one-step acceptance  err^3 <= 1e-14 (gap - 3 ulp32 scale)^2."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import robchar_oracle as orc
f32 = np.float32
TOL = f32(float(os.environ.get("TOL", "1e-4")))      # kF32SplitTol
CAP = 10

def ql_implicit(d, e):
    """d (T,64,N), e (T,64,N-1) fp32 -> eigenvalues (unsorted), sweeps per tile, steps per tile, failed samples"""
    T, W, N = d.shape
    d = d.copy(); e = np.concatenate([e, np.zeros((T, W, 1), f32)], axis=2)
    scale = np.maximum(np.abs(d).max(axis=2), np.abs(e).max(axis=2))
    thr = TOL * scale
    sweeps = np.zeros(T, int); steps = np.zeros(T, int); bad = np.zeros((T, W), bool)
    for l in range(N - 1):
        if l == N - 2:
            el = e[:, :, l]; delta = f32(0.5) * (d[:, :, l + 1] - d[:, :, l])
            h = delta * delta + (el * el + f32(1e-30))
            t = np.copysign(np.sqrt(h) - np.abs(delta), delta)
            d[:, :, l] -= t; d[:, :, l + 1] += t
            break
        it = np.zeros(T, int)
        while True:
            done = (np.abs(e[:, :, l]) <= thr) | bad
            act = ~done.all(axis=1)
            if not act.any():
                break
            A = np.where(act)[0]
            dd = d[A]; ee = e[A]
            el = ee[:, :, l]; delta = f32(0.5) * (dd[:, :, l + 1] - dd[:, :, l])
            h0 = delta * delta + (el * el + f32(1e-30))
            g = dd[:, :, N - 1] - dd[:, :, l] + np.copysign(np.sqrt(h0) - np.abs(delta), delta)
            sn = np.ones_like(g); cs = np.ones_like(g); p = np.zeros_like(g)
            for i in range(N - 2, l - 1, -1):
                f = sn * ee[:, :, i]; b = cs * ee[:, :, i]; gn = g + f32(1e-15)
                h = f * f + gn * gn
                rinv = f32(1.0) / np.sqrt(h)
                if i + 1 <= N - 2:
                    ee[:, :, i + 1] = h * rinv
                sn = f * rinv; cs = gn * rinv
                g = dd[:, :, i + 1] - p
                r = (dd[:, :, i] - g) * sn + f32(2.0) * cs * b
                p = sn * r
                dd[:, :, i + 1] = g + p
                g = cs * r - b
            dd[:, :, l] -= p; ee[:, :, l] = g
            d[A] = dd; e[A] = ee
            it[A] += 1; sweeps[A] += 1; steps[A] += N - 1 - l
            capped = A[it[A] >= CAP]
            if len(capped):
                bad[capped] |= ~(np.abs(e[capped][:, :, l]) <= thr[capped])
    return d, sweeps, steps, bad

def ql_rational(d, e, pwk=False):
    """Reinsch's rational QL (or PWK) with explicit shifts, lock-step per tile.  e -> squares."""
    T, W, N = d.shape
    d = d.copy(); e2 = np.concatenate([e * e, np.zeros((T, W, 1), f32)], axis=2)
    scale = np.maximum(np.abs(d).max(axis=2), np.abs(e).max(axis=2))
    thr2 = (TOL * scale) ** 2
    tiny = f32(1e-30)
    lam = np.zeros_like(d)
    fshift = np.zeros((T, W), f32)
    sweeps = np.zeros(T, int); steps = np.zeros(T, int); bad = np.zeros((T, W), bool)
    with np.errstate(all="ignore"):
        for l in range(N - 1):
            if l == N - 2:
                el2 = e2[:, :, l]; delta = f32(0.5) * (d[:, :, l + 1] - d[:, :, l])
                h = delta * delta + (el2 + tiny)
                t = np.copysign(np.sqrt(h) - np.abs(delta), delta)
                lam[:, :, l] = d[:, :, l] - t + fshift; lam[:, :, l + 1] = d[:, :, l + 1] + t + fshift
                break
            it = np.zeros(T, int)
            while True:
                done = (e2[:, :, l] <= thr2) | bad
                act = ~done.all(axis=1)
                if not act.any():
                    break
                A = np.where(act)[0]
                dd = d[A]; ee = e2[A]
                el2 = ee[:, :, l]; delta = f32(0.5) * (dd[:, :, l + 1] - dd[:, :, l])
                h0 = delta * delta + (el2 + tiny)
                sig = dd[:, :, l] - np.copysign(np.sqrt(h0) - np.abs(delta), delta)      # Wilkinson shift
                dd -= sig[:, :, None]                     # explicit shift (only i >= l matter)
                fshift[A] += sig
                if not pwk:
                    g = dd[:, :, N - 1].copy(); g = np.where(g == 0, tiny, g)
                    h = g.copy(); s = np.zeros_like(g)
                    for i in range(N - 2, l - 1, -1):
                        p = g * h
                        r = p + ee[:, :, i]
                        if i + 1 <= N - 2:
                            ee[:, :, i + 1] = s * r
                        rinv = f32(1.0) / r
                        s = ee[:, :, i] * rinv
                        dd[:, :, i + 1] = h + s * (h + dd[:, :, i])
                        g = dd[:, :, i] - ee[:, :, i] * (f32(1.0) / g)
                        g = np.where(g == 0, tiny, g)
                        h = g * p * rinv
                    ee[:, :, l] = s * g * h
                    dd[:, :, l] = h
                else:
                    c = np.ones_like(sig); s = np.zeros_like(sig)
                    gamma = dd[:, :, N - 1].copy(); p = gamma * gamma
                    for i in range(N - 2, l - 1, -1):
                        bb = ee[:, :, i]
                        r = p + bb
                        if i + 1 <= N - 2:
                            ee[:, :, i + 1] = s * r
                        oldc = c
                        rinv = f32(1.0) / r
                        c = p * rinv; s = bb * rinv
                        oldgam = gamma
                        alpha = dd[:, :, i]
                        gamma = c * alpha - s * oldgam
                        dd[:, :, i + 1] = oldgam + (alpha - gamma)
                        p = np.where(c != 0, gamma * gamma * r / p, oldc * bb)
                    ee[:, :, l] = s * p
                    dd[:, :, l] = gamma
                d[A] = dd; e2[A] = ee
                it[A] += 1; sweeps[A] += 1; steps[A] += N - 1 - l
                capped = A[it[A] >= CAP]
                if len(capped):
                    bad[capped] |= ~(e2[capped][:, :, l] <= thr2[capped])
            lam[:, :, l] = d[:, :, l] + fshift
    bad |= ~np.isfinite(lam).all(axis=2)
    return lam, sweeps, steps, bad

def workload(name, N, C=100, K=1024, sigma=0.05, seed=0):
    rng = np.random.default_rng(N)
    ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
    h0 = np.zeros(N)
    g = sigma * np.random.default_rng(seed).standard_normal((C, K, N, 3))
    if name == "xxz":
        h0 = orc.xxz_delta(N)
    elif name == "flat":                      # no bias at all: d = noise only (exact zeros / symmetric structure)
        ctrl[:, :N] = 0.0
    elif name == "sigma0":                    # all samples identical, d integers
        ctrl[:, :N] = np.round(ctrl[:, :N]); g[:] = 0.0
    elif name == "resonant":                  # two sites 1e-6 .. 1e-2 apart
        for c in range(C):
            i, j = sorted(rng.choice(N, 2, replace=False))
            ctrl[c, j] = ctrl[c, i] + 10.0 ** rng.uniform(-6, -2)
    elif name == "big":
        ctrl[:, :N] *= 10.0
    d = (ctrl[:, None, :N] + h0 + g[..., 0]).reshape(-1, N)
    e = np.hypot(1.0 + g[..., 1:, 1], g[..., 1:, 2]).reshape(-1, N - 1)
    return d, e

def report(tag, lam32, sweeps, steps, bad, true, gaps, scale):
    lam = np.sort(lam32.reshape(-1, lam32.shape[-1]).astype(np.float64), axis=1)
    err = np.abs(lam - true).max(axis=1)
    err = np.where(np.isfinite(err), err, 1e30)
    unc = 3 * 1.19e-7 * scale
    okk = (err ** 3 <= 1e-14 * np.maximum(gaps - unc, 0) ** 2) & ~bad.reshape(-1)
    tiles = okk.reshape(-1, 64).all(axis=1)
    print(f"  {tag:9s} sweeps/tile {sweeps.mean():6.2f}  steps/tile {steps.mean():6.1f}  err/scale median {np.median(err / scale):.1e} "
          f"p99 {np.quantile(err / scale, 0.99):.1e} max {np.max(err / scale):.1e}  capped/NaN {bad.mean():.1e}  "
          f"samples one-step {okk.mean():.4f}  tiles one-step {tiles.mean():.3f}")

for name, N in (("plain", 7), ("xxz", 10), ("plain", 5), ("plain", 13), ("flat", 7), ("sigma0", 7), ("resonant", 7), ("big", 10)):
    d, e = workload(name, N)
    S = d.shape[0]
    H = np.zeros((S, N, N)); idx = np.arange(N)
    H[:, idx, idx] = d; H[:, idx[:-1], idx[1:]] = e; H[:, idx[1:], idx[:-1]] = e
    true = np.linalg.eigvalsh(H)
    gaps = np.diff(true, axis=1).min(axis=1)
    scale = np.maximum(np.abs(d).max(axis=1), np.abs(e).max(axis=1))
    d3 = d.astype(f32).reshape(-1, 64, N); e3 = e.astype(f32).reshape(-1, 64, N - 1)
    print(f"{name} N={N}: {S} samples")
    report("implicit", *ql_implicit(d3, e3), true, gaps, scale)
    report("rational", *ql_rational(d3, e3), true, gaps, scale)
    report("pwk", *ql_rational(d3, e3, pwk=True), true, gaps, scale)
