#!/usr/bin/env python3
"""Prototype (numpy + mpmath): eigenvalue-only ring weights next to a close eigenvalue pair, with and without the SUM-RULE
PROJECTION - the residual r0 = sum_k w_k - [in = out] of the computed cofactor weights is subtracted from one member of the pair.
The weights of a pair (a, b) carry errors eps_a, eps_b ~ nu / gap from the rounding noise nu of their numerators; the symmetric
part eps_a + eps_b = r0 is what breaks the fidelity, the antisymmetric part only enters as eps (e^{-iT lam_a} - e^{-iT lam_b}) ~
eps T gap = nu T.  Rings with ONE pair split by a tunable amount; eigenvalues from mpmath rounded to double (what the polished
roots are), weights from the kernel's formulas in double."""
import numpy as np, mpmath as mp
mp.mp.dps = 50
rng = np.random.default_rng(1)

def P(d, e2, idx, lam):
    """char poly of the open chain through the site list idx (consecutive in idx are coupled by e2 keyed on the lower site; the
    corner is passed in e2 under key -1)"""
    if len(idx) == 0: return np.ones_like(lam)
    pm, p = np.ones_like(lam), lam - d[idx[0]]
    for a, b in zip(idx[:-1], idx[1:]):
        key = min(a, b) if abs(a - b) == 1 else -1
        pm, p = p, (lam - d[b]) * p - e2[key] * pm
    return p

def case(N, amp, delta, T, a, b):
    """two resonant sites (d_j = d_i + delta) on a ring with |d| <= amp: ONE eigenvalue pair, split by delta and by the effective
    coupling through the rest of the ring"""
    d = rng.uniform(-amp, amp, N)
    i, j = sorted(rng.choice(N, 2, replace=False))
    d[j] = d[i] + delta
    dd = d
    h = (1 + 0.05 * rng.standard_normal(N - 1)) + 1j * 0.05 * rng.standard_normal(N - 1)
    c = 1.0
    def build(dd):
        H = mp.zeros(N, N)
        for i in range(N): H[i, i] = mp.mpf(float(dd[i]))
        for i in range(N - 1):
            H[i + 1, i] = mp.mpc(float(h[i].real), float(h[i].imag)); H[i, i + 1] = mp.mpc(float(h[i].real), -float(h[i].imag))
        H[N - 1, 0] += c; H[0, N - 1] += c
        return H
    H = build(dd)
    E, Q = mp.eighe(H)
    phi = sum(Q[b, k] * mp.conj(Q[a, k]) * mp.expj(-T * E[k]) for k in range(N))
    Fexact = float(abs(phi) ** 2)
    lam = np.array([float(e) for e in E])                       # polished roots: exact to an ulp
    e2 = {i: float(abs(h[i]) ** 2) for i in range(N - 1)}; e2[-1] = c * c
    dchi = np.array([np.prod([lam[k] - lam[m] for m in range(N) if m != k]) for k in range(N)])
    lo, hi = min(a, b), max(a, b)
    wrap = list(range(hi + 1, N)) + list(range(0, lo))
    between = list(range(lo + 1, hi))
    A = np.prod(h[lo:hi]) if hi > lo else 1.0
    B = 0.0 if lo == hi else c * np.conj(np.prod(h[:lo]) * np.prod(h[hi:]))
    W = (A * P(dd, e2, wrap, lam) + B * P(dd, e2, between, lam)) / dchi      # amplitude for (row hi, col lo)
    if b == lo and a != b: W = np.conj(W)
    m0 = 1.0 if a == b else 0.0
    r0 = W.sum() - m0
    F0 = abs((W * np.exp(-1j * T * lam)).sum()) ** 2
    ka = int(np.argmin(np.diff(lam)))                           # the pair (ka, ka + 1)
    Wp = W.copy(); Wp[ka] -= r0
    F1 = abs((Wp * np.exp(-1j * T * lam)).sum()) ** 2
    return np.diff(lam).min(), abs(r0), abs(F0 - Fexact), abs(F1 - Fexact)

import collections
bins = collections.defaultdict(list)
for N in (5, 7, 10):
    for amp in (10.0, 30.0, 100.0):
        for delta in (1e-2, 1e-4, 1e-6, 1e-9, 0.0):
            for rep in range(6):
                for (a, b) in ((0, N // 2), (1, 1), (N - 1, 0)):
                    g, r0, e0, e1 = case(N, amp, delta, 30.0, a, b)
                    if g <= 0: continue
                    bins[int(np.floor(np.log10(g / amp)))].append((r0, e0, e1, N, amp))
for k in sorted(bins, reverse=True):
    v = bins[k]
    print(f"gap/scale ~1e{k}: {len(v):4d} samples  |r0| up to {max(x[0] for x in v):.1e}   |dF| plain up to {max(x[1] for x in v):.1e}   projected up to {max(x[2] for x in v):.1e}")
