#!/usr/bin/env python3
"""Prototype (numpy): controller-shared starting values for the mixed-precision eigenvalue path.
Unperturbed H0 = tridiag(x + h0d, J) per controller -> (lam0, V0); per sample first-order perturbation theory in fp32,
then 0/1/2 fp32 Halley steps on chi; how many 64-sample tiles would pass the one-step fp64 acceptance?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import robchar_oracle as orc

def halley32(d, e2, mu, steps):
    # d: (S,N) f32, e2: (S,N-1) f32, mu: (S,N) f32
    N = d.shape[1]
    mu = mu.copy()
    for _ in range(steps):
        for k in range(N):
            m_ = mu[:, k]
            pm = np.ones_like(m_); p = m_ - d[:, 0]
            dm = np.zeros_like(m_); dp = np.ones_like(m_)
            qm = np.zeros_like(m_); q = np.zeros_like(m_)
            for m in range(1, N):
                t = m_ - d[:, m]; c = e2[:, m - 1]
                pn = t * p - c * pm
                dn = t * dp - c * dm + p
                qn = t * q - c * qm + dp
                pm, p, dm, dp, qm, q = p, pn, dp, dn, q, qn
            den = dp * dp - p * q
            mu[:, k] = m_ - (p * dp) / den
    return mu

def run(N, cfg_id, xxz, sigma=0.05, C=100, K=1024, seed=0):
    rng = np.random.default_rng(20220714 + cfg_id)
    ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
    h0 = orc.xxz_delta(N) if xxz else np.zeros(N)
    g = sigma * np.random.default_rng(seed).standard_normal((C, K, N, 3))
    res = {}
    tiles = {k: 0 for k in ("pt1", "pt1+h", "pt1+2h", "gapok")}
    ntile = 0
    worst = []
    for c in range(C):
        d0 = ctrl[c, :N] + h0
        H0 = np.diag(d0) + np.diag(np.ones(N - 1), 1) + np.diag(np.ones(N - 1), -1)
        lam0, V = np.linalg.eigh(H0)
        gap0 = np.diff(lam0).min()
        d = d0[None, :] + g[c, :, :, 0]
        e = np.hypot(1.0 + g[c, :, 1:, 1], g[c, :, 1:, 2])
        # truth
        Hs = np.zeros((K, N, N)); idx = np.arange(N)
        Hs[:, idx, idx] = d; Hs[:, idx[:-1], idx[1:]] = e; Hs[:, idx[1:], idx[:-1]] = e
        lam = np.linalg.eigvalsh(Hs)
        # PT1 in fp32
        A = (V * V).T.astype(np.float32)                       # [k][i]
        B = (2 * V[:-1] * V[1:]).T.astype(np.float32)          # [k][i]
        dd = g[c, :, :, 0].astype(np.float32); de = (e - 1.0).astype(np.float32)
        mu = lam0.astype(np.float32)[None, :] + dd @ A.T + de @ B.T
        d32 = d.astype(np.float32); e2 = (e * e).astype(np.float32)
        gaps = np.diff(lam, axis=1).min(axis=1)
        def passes(muv):
            muv = np.sort(muv.astype(np.float64), axis=1)
            err = np.abs(muv - lam).max(axis=1)
            return (err ** 3 <= 1e-14 * np.maximum(gaps - 4e-6, 0) ** 2), err
        for name, st in (("pt1", 0), ("pt1+h", 1), ("pt1+2h", 2)):
            ok, err = passes(halley32(d32, e2, mu, st))
            for t in range(K // 64):
                tiles[name] += ok[t * 64:(t + 1) * 64].all()
            res.setdefault(name, []).append(np.median(err))
            if name == "pt1+h": worst.append((gap0, ok.mean(), np.median(err)))
        ntile += K // 64
    print(f"N={N} xxz={xxz} sigma={sigma}: tiles passing one-step fp64 acceptance: " + ", ".join(f"{k} {tiles[k]/ntile:.3f}" for k in ("pt1", "pt1+h", "pt1+2h")))
    worst.sort()
    print("   controllers by unperturbed min gap (gap0, sample pass rate pt1+h, median err):")
    for w in worst[:12]: print("    %.3f  %.3f  %.2e" % w)
    w = np.array(worst)
    for thr in (0.1, 0.2, 0.3, 0.5, 1.0):
        sel = w[:, 0] >= thr
        print(f"   gap0 >= {thr}: {sel.mean():.2f} of controllers, their sample pass rate {w[sel,1].mean():.4f}")

if __name__ == "__main__":
    run(7, 3, False)
    run(10, 5, True)
