#!/usr/bin/env python3
"""Prototype of the eigenvalue-only route for the RING topology (numpy, checked against dense eigh):
  chi_ring(lam) = P_{0..N-1}(lam) - c^2 P_{1..N-2}(lam) - Phi,   Phi = 2 c Re(prod h_i) * sign
  w_k = u_k[out] conj(u_k[in]) = cofactor / chi'(lam_k): two paths around the ring."""
import numpy as np
rng = np.random.default_rng(0)

def P(d, e2, lo, hi, lam):
    """char poly of the open chain on sites lo..hi (inclusive), e2[i] = |coupling between i and i+1|^2; empty -> 1"""
    if hi < lo: return np.ones_like(lam)
    pm, p = np.ones_like(lam), lam - d[lo]
    for m in range(lo + 1, hi + 1):
        pm, p = p, (lam - d[m]) * p - e2[m - 1] * pm
    return p

for N in (3, 4, 5, 7, 10):
    for trial in range(3):
        d = rng.uniform(-10, 10, N)
        h = (1 + 0.3 * rng.standard_normal(N - 1)) + 1j * 0.3 * rng.standard_normal(N - 1)   # h[i] = H[i+1, i]
        c = 1.0 + 0.2 * rng.standard_normal()                                                 # H[N-1, 0] = H[0, N-1] = c (real)
        H = np.diag(d).astype(complex)
        for i in range(N - 1):
            H[i + 1, i] = h[i]; H[i, i + 1] = np.conj(h[i])
        H[N - 1, 0] += c; H[0, N - 1] += c
        lam, U = np.linalg.eigh(H)
        e2 = np.abs(h) ** 2
        if N == 3:
            pass
        # characteristic polynomial
        Phi = 2 * c * np.real(np.prod(np.conj(h)))            # candidate: orientation / sign pinned below
        for sgn in (+1, -1):
            for conj in (False, True):
                pr = np.prod(h if not conj else np.conj(h))
                Phi = sgn * 2 * c * np.real(pr)
                chi = P(d, e2, 0, N - 1, lam) - c * c * P(d, e2, 1, N - 2, lam) - Phi
                if np.abs(chi).max() < 1e-6 * np.abs(P(d, e2, 0, N - 1, lam)).max() + 1e-7:
                    found = (sgn, conj)
        # weights for every (a, b)
        dchi = np.array([np.prod([lam[k] - lam[m] for m in range(N) if m != k]) for k in range(N)])
        worst = 0
        for a in range(N):
            for b in range(N):
                ref = U[b, :] * np.conj(U[a, :])               # u_k[out=b] conj(u_k[in=a])
                if a == b:
                    # diagonal cofactor: the ring with site a removed = open chain of the other N-1 sites (wrapped)
                    # wrapped chain a+1..N-1,0..a-1 with the corner c as one of its bonds
                    dd = np.concatenate([d[a + 1:], d[:a]])
                    ee = np.concatenate([e2[a + 1:], [c * c] if (a != 0 and a != N - 1) else [], e2[:max(a - 1, 0)]])
                    w = P(dd, ee, 0, N - 2, lam) / dchi
                else:
                    lo, hi = min(a, b), max(a, b)
                    # path A: lo -> hi through the sites between (bonds h[lo..hi-1]); complement = wrapped chain hi+1..N-1,0..lo-1
                    dd = np.concatenate([d[hi + 1:], d[:lo]])
                    nA = (N - 1 - hi) + lo
                    eeA = np.concatenate([e2[hi + 1:], [c * c] if (hi != N - 1 and lo != 0) else [], e2[:max(lo - 1, 0)]])
                    PA = P(dd, eeA, 0, nA - 1, lam)
                    prodA = np.prod(h[lo:hi])                  # H[lo+1,lo] ... H[hi,hi-1]: amplitude lo -> hi
                    # path B: hi -> N-1 -> 0 -> lo around the corner; complement = sites lo+1..hi-1
                    PB = P(d, e2, lo + 1, hi - 1, lam)
                    prodB = np.prod(np.conj(h[:lo])) * c * np.prod(np.conj(h[hi:]))   # amplitude lo -> 0 -> N-1 -> hi (backwards along the bonds)
                    sA, sB = 1.0, (-1.0) ** (N)                 # sign of path B pinned below
                    best = None
                    for sB in (+1, -1):
                        amp = prodA * PA + sB * prodB * PB      # amplitude for (row hi, col lo)
                        wk = amp / dchi
                        if b == lo: wk = np.conj(wk)
                        err = np.abs(wk - ref).max()
                        if best is None or err < best[0]: best = (err, sB)
                    w = None; worst = max(worst, best[0]); signB = best[1]
                    continue
                worst = max(worst, np.abs(w - ref).max())
        print(N, trial, "chi sign/conj", found, "max weight err", f"{worst:.1e}", "path-B sign", signB, "(-1)^N", (-1) ** N)
