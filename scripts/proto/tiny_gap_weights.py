#!/usr/bin/env python3
"""Prototype: how small may an eigenvalue gap be before the product-formula weights
w_k = prod(e) * phi_i(lam_k) psi_j(lam_k) / prod_{m != k}(lam_k - lam_m) stop giving an accurate |sum_k w_k e^{-iT lam_k}|^2 ?
Reference: mpmath (50 digits) eigen-decomposition.  Eigenvalues fed to the formula: the exact ones + noise of eps*scale
(what a backward-stable fp64 QL delivers), INDEPENDENT per eigenvalue (worst case for a close pair)."""
import numpy as np, mpmath as mp
mp.mp.dps = 60
rng = np.random.default_rng(5)

def exact(d, e, T, a, b):
    N = len(d)
    A = mp.matrix(N, N)
    for i in range(N):
        A[i, i] = mp.mpf(float(d[i]))
        if i < N - 1:
            A[i, i + 1] = A[i + 1, i] = mp.mpf(float(e[i]))
    E, Q = mp.eigsy(A)
    lam = [E[k] for k in range(N)]
    phi = sum(Q[b, k] * Q[a, k] * mp.e ** (-1j * T * lam[k]) for k in range(N))
    return float(abs(phi) ** 2), np.array([float(x) for x in lam]), lam

def formula(d, e, lam, T, a, b):
    N = len(d)
    lo, hi = min(a, b), max(a, b)
    pe = np.prod(e[lo:hi])
    w = np.empty(N)
    for k in range(N):
        den = 1.0
        for m in range(N):
            if m != k:
                df = lam[k] - lam[m]
                if df == 0.0:
                    df = np.copysign(2.2e-16 * max(1.0, np.abs(lam).max()), k - m)
                den *= df
        # phi_lo(lam_k): leading lo x lo block; psi: trailing block below hi
        p0, p1 = 1.0, 1.0
        if lo > 0:
            p0, p1 = 1.0, lam[k] - d[0]
            for m in range(1, lo):
                p0, p1 = p1, (lam[k] - d[m]) * p1 - e[m - 1] ** 2 * p0
        q0, q1 = 1.0, 1.0
        if hi < N - 1:
            q0, q1 = 1.0, lam[k] - d[N - 1]
            for m in range(N - 2, hi, -1):
                q0, q1 = q1, (lam[k] - d[m]) * q1 - e[m] ** 2 * q0
        w[k] = pe * p1 * q1 / den
    phi = (w * np.exp(-1j * T * (lam - lam[0]))).sum()
    return abs(phi) ** 2, w

def lanczos_mp(lam, q):
    N = len(lam)
    L = [mp.mpf(float(x)) for x in lam]
    q = [mp.mpf(float(x)) for x in q]
    nq = mp.sqrt(sum(x * x for x in q)); q = [x / nq for x in q]
    Q, alpha, beta = [q], [], []
    for i in range(N):
        w = [L[t] * Q[i][t] for t in range(N)]
        a = sum(w[t] * Q[i][t] for t in range(N)); alpha.append(a)
        w = [w[t] - a * Q[i][t] - (beta[-1] * Q[i - 1][t] if i else 0) for t in range(N)]
        for qq in Q:
            c = sum(w[t] * qq[t] for t in range(N)); w = [w[t] - c * qq[t] for t in range(N)]
        if i < N - 1:
            b = mp.sqrt(sum(x * x for x in w)); beta.append(b); Q.append([x / b for x in w])
    return np.array([float(x) for x in alpha]), np.array([float(x) for x in beta])

worst = {}
for trial in range(240):
    N = int(rng.choice([5, 7, 10]))
    while True:
        lam0 = np.sort(rng.uniform(-10, 10, N))
        if np.diff(lam0).min() > 0.5: break
    j = int(rng.integers(0, N - 1))
    g = 10.0 ** rng.uniform(-15.5, -3)
    lam0[j + 1:] -= (lam0[j + 1] - lam0[j]) - g
    if trial % 4 == 3 and j + 2 < N:                     # a triple
        lam0[j + 2:] -= (lam0[j + 2] - lam0[j + 1]) - g * rng.uniform(0.3, 3)
    d, e = lanczos_mp(lam0, rng.uniform(0.3, 1, N))
    T = rng.uniform(2, 30)
    for (a, b) in ((0, N - 1), (0, N // 2), (1, N - 2), (0, 1)):
        f_ref, lam_f, lam_mp = exact(d, e, T, a, b)
        scale = max(1.0, np.abs(lam_f).max())
        for rep in range(4):
            lam = np.sort(lam_f + 3 * 2.2e-16 * scale * rng.uniform(-1, 1, N))        # independent rounding-level errors
            gap = np.diff(np.array([float(x) for x in sorted(lam_mp)])).min()
            f, w = formula(d, e, lam, T, a, b)
            key = int(np.floor(np.log10(max(gap, 1e-20))))
            err = abs(f - f_ref)
            rec = worst.setdefault(key, [0.0, 0.0, 0, 0.0])
            if err > rec[0]: rec[3] = np.abs(w).max()
            rec[0] = max(rec[0], err); rec[1] = max(rec[1], np.abs(w).max()); rec[2] += 1
for k in sorted(worst):
    print(f"min gap ~1e{k:+d}: cases {worst[k][2]:4d}  max |dF| {worst[k][0]:.2e} (max|w| there {worst[k][3]:.1e})  max |w| {worst[k][1]:.2e}")
