#!/usr/bin/env python3
"""Prototype (numpy fp32, lock-step tiles of 64): how much of the fp32 QL's lock-step waste does RUN-AHEAD recover?
A lane whose coupling e[l] has converged while the tile still sweeps stage l takes its Wilkinson shift from its own next
unconverged stage (up to `depth` levels ahead); the sweep itself stays wave-uniform (a rotation through a converged
coupling is the identity).  Reports inner steps per tile for depth 0 (the kernel today), 1, 2, unlimited, and what a
sample alone would need."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import robchar_oracle as orc
f32 = np.float32
TOL = f32(1e-4); CAP = 10

def ql(d, e, depth, W):
    S, N = d.shape
    T = S // W
    d = d.reshape(T, W, N).copy(); e = np.concatenate([e.reshape(T, W, N - 1), np.zeros((T, W, 1), f32)], axis=2)
    scale = np.maximum(np.abs(d).max(axis=2), np.abs(e).max(axis=2)); thr = TOL * scale
    steps = np.zeros(T, int); sweeps = np.zeros(T, int)
    for l in range(N - 2):
        it = np.zeros(T, int)
        while True:
            done = np.abs(e[:, :, l]) <= thr
            act = ~done.all(axis=1) & (it < CAP)
            if not act.any():
                break
            A = np.where(act)[0]
            dd = d[A]; ee = e[A]; th = thr[A]
            # per-lane stage: first j in [l, l+depth] (and <= N-3) with |e[j]| > thr; else the last candidate
            ll = np.full(dd.shape[:2], l)
            for j in range(l, min(l + depth, N - 3) + 1):
                cur = np.take_along_axis(ee, ll[:, :, None], axis=2)[:, :, 0]
                adv = (np.abs(cur) <= th) & (ll == j) & (j + 1 <= min(l + depth, N - 3))
                ll = np.where(adv, j + 1, ll)
            dl = np.take_along_axis(dd, ll[:, :, None], axis=2)[:, :, 0]
            dl1 = np.take_along_axis(dd, ll[:, :, None] + 1, axis=2)[:, :, 0]
            el = np.take_along_axis(ee, ll[:, :, None], axis=2)[:, :, 0]
            delta = f32(0.5) * (dl1 - dl)
            h0 = delta * delta + (el * el + f32(1e-30))
            g = dd[:, :, N - 1] - dl + np.copysign(np.sqrt(h0) - np.abs(delta), delta)
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
    l = N - 2
    el = e[:, :, l]; delta = f32(0.5) * (d[:, :, l + 1] - d[:, :, l])
    h = delta * delta + (el * el + f32(1e-30))
    t = np.copysign(np.sqrt(h) - np.abs(delta), delta)
    d[:, :, l] -= t; d[:, :, l + 1] += t
    return d.reshape(S, N), sweeps, steps

for name, N in (("plain", 7), ("xxz", 10), ("plain", 5), ("plain", 13)):
    rng = np.random.default_rng(N); C, K = 100, 1024
    ctrl = rng.uniform(-10, 10, (C, N)); g = 0.05 * np.random.default_rng(0).standard_normal((C, K, N, 3))
    h0 = orc.xxz_delta(N) if name == "xxz" else np.zeros(N)
    d = (ctrl[:, None, :] + h0 + g[..., 0]).reshape(-1, N); e = np.hypot(1.0 + g[..., 1:, 1], g[..., 1:, 2]).reshape(-1, N - 1)
    Hm = np.zeros((d.shape[0], N, N)); idx = np.arange(N)
    Hm[:, idx, idx] = d; Hm[:, idx[:-1], idx[1:]] = e; Hm[:, idx[1:], idx[:-1]] = e
    true = np.linalg.eigvalsh(Hm)
    d32, e32 = d.astype(f32), e.astype(f32)
    print(f"{name} N={N}")
    for depth, W in ((0, 1), (0, 64), (1, 64), (2, 64), (99, 64)):
        lam, sw, st = ql(d32, e32, depth, W)
        err = np.abs(np.sort(lam.astype(np.float64), axis=1) - true).max(axis=1) / np.abs(d).max(axis=1)
        print(f"  tile {W:2d} depth {depth:2d}: sweeps/tile {sw.mean():6.2f}  steps/tile {st.mean():6.1f}   err/scale median {np.median(err):.1e} max {err.max():.1e}")
