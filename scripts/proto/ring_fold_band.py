#!/usr/bin/env python3
"""Prototype (numpy) of the FOLDED BAND REDUCTION of the ring Hamiltonian (round 5, hermitian_core.h: ring_fold_*).

A periodic tridiagonal Hermitian matrix (chain bonds h_i = H[i+1, i], corner H[N-1, 0]) in the site order
0, N-1, 1, N-2, 2, ... is PENTADIAGONAL: bonds sit at distance 2, the corner at (1, 0) and the middle bond at (N-1, N-2).
Schwarz' band reduction takes it to tridiagonal form with ~N^2/4 unitary 2 x 2 rotations that each touch O(1) entries of
three diagonals (+ one bulge), O(N^2) work and 6N numbers of state - against O(N^3) work and N^2 numbers for the dense
Householder reduction the lane-per-sample ring kernels use up to N = 10.  Checks: eigenvalues of the resulting real
tridiagonal matrix = eigvalsh(H); U[out, in] from the accumulated rows = expm(-i T H)[out, in]."""
import numpy as np
from scipy.linalg import expm


def fold_positions(N):
    site = [0] * N
    for j in range((N + 1) // 2):
        site[2 * j] = j
        if 2 * j + 1 < N:
            site[2 * j + 1] = N - 1 - j
    return site                      # site[p] = chain site at folded position p


def ring_fold_tridiag(d, h, corner, rows=()):
    """d[N] real diagonal, h[N-1] complex bonds H[i+1, i], corner = H[N-1, 0].  Returns (a, b) of the real symmetric
    tridiagonal matrix and, for every site r in `rows`, the row r of Q (H = Q T Q^H)."""
    N = len(d)
    site = fold_positions(N)
    pos = {s: p for p, s in enumerate(site)}
    a = np.array([d[site[p]] for p in range(N)], dtype=float)
    b = np.zeros(N, dtype=complex)           # b[p] = B[p+1][p]
    c = np.zeros(N, dtype=complex)           # c[p] = B[p+2][p]

    def H(i, j):                             # lower or upper element of the ring matrix between sites i, j
        if i == j + 1:
            return h[j]
        if j == i + 1:
            return np.conj(h[i])
        if i == N - 1 and j == 0:
            return corner
        if i == 0 and j == N - 1:
            return np.conj(corner)
        return 0.0
    for p in range(N - 1):
        b[p] = H(site[p + 1], site[p])
    for p in range(N - 2):
        c[p] = H(site[p + 2], site[p])
    z = {r: np.eye(N, dtype=complex)[pos[r]] for r in rows}       # columns of U_total

    def rot(i, x, y):
        """unitary U on the plane (i, i+1) with U [x; y] = [rho; 0]; updates the 2x2 block, the column to the left is the
        caller's; returns (rho, p, q)"""
        rho = np.sqrt(abs(x) ** 2 + abs(y) ** 2)
        if abs(y) == 0:
            return (x, 1.0, 0.0)             # nothing to eliminate (x stays as it is - possibly complex)
        return (rho, np.conj(x) / rho, np.conj(y) / rho)

    nrot = 0
    for k in range(N - 2):
        # eliminate c[k] = B[k+2][k] with a rotation of rows / columns (k+1, k+2)
        i = k + 1
        x, y = b[k], c[k]
        rho, p, q = rot(i, x, y)
        b[k], c[k] = rho, 0.0
        bulge = 0.0
        while True:
            nrot += 1
            # 2 x 2 diagonal block (i, i+1)
            al, de, be = a[i], a[i + 1], b[i]
            t = 2 * (q * be * np.conj(p)).real
            n11 = abs(p) ** 2 * al + abs(q) ** 2 * de + t
            a[i], a[i + 1] = n11, al + de - n11
            b[i] = np.conj(p) * np.conj(q) * (de - al) + np.conj(p) ** 2 * be - np.conj(q) ** 2 * np.conj(be)
            # rows below: column op on columns (i, i+1): [B[t][i], B[t][i+1]] <- [..] U^H
            if i + 2 < N:
                u, v = c[i], b[i + 1]                         # row i+2: B[i+2][i], B[i+2][i+1]
                c[i] = u * np.conj(p) + v * np.conj(q)
                b[i + 1] = -u * q + v * p
            new_bulge = 0.0
            if i + 3 < N:
                u, v = 0.0, c[i + 1]                          # row i+3: B[i+3][i] (outside the band: 0), B[i+3][i+1]
                new_bulge = u * np.conj(p) + v * np.conj(q)
                c[i + 1] = -u * q + v * p
            for r in z:
                zi, zj = z[r][i], z[r][i + 1]
                z[r][i], z[r][i + 1] = p * zi + q * zj, -np.conj(q) * zi + np.conj(p) * zj
            # chase the bulge at (i+3, i): rotation of (i+2, i+3) on [c[i]; bulge]
            if i + 3 >= N or new_bulge == 0.0 and False:
                break
            x, y = c[i], new_bulge
            rho, p, q = rot(i + 2, x, y)
            c[i] = rho
            # the column i+1 entries of rows (i+2, i+3): row op
            u, v = b[i + 1], c[i + 1]                         # B[i+2][i+1], B[i+3][i+1]
            b[i + 1] = p * u + q * v
            c[i + 1] = -np.conj(q) * u + np.conj(p) * v
            i += 2
    # make the sub-diagonal real: D^H T D with a diagonal unitary D accumulated from the top
    e = np.zeros(N)
    ph = 1.0 + 0j
    out_rows = {}
    phases = np.ones(N, dtype=complex)
    for p_ in range(N - 1):
        # T[p+1][p] = b[p]; after scaling basis vector p+1 by phase f: b'[p] = conj(f_{p+1}) b[p] f_p -> choose f_{p+1} = f_p b[p]/|b[p]|
        bb = b[p_] * phases[p_]
        m = abs(bb)
        phases[p_ + 1] = bb / m if m > 0 else 1.0
        e[p_] = m
    for r in z:
        # Q = U_total^H D  ->  row r of Q: conj(z[r][k]) * phases[k]
        out_rows[r] = np.conj(z[r]) * phases
    return a, e, out_rows, nrot


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    worst = 0.0
    for N in range(3, 17):
        for trial in range(30):
            d = rng.uniform(-10, 10, N) if trial % 3 else np.zeros(N)
            h = 1.0 + 0.05 * (rng.standard_normal(N - 1) + 1j * rng.standard_normal(N - 1))
            if trial % 7 == 3:
                h[rng.integers(0, N - 1)] = 0.0            # a cut bond
            corner = 1.0
            Hm = np.diag(d).astype(complex)
            for i in range(N - 1):
                Hm[i + 1, i] = h[i]
                Hm[i, i + 1] = np.conj(h[i])
            if N > 2:
                Hm[N - 1, 0] += corner
                Hm[0, N - 1] += np.conj(corner)
            a_, b_ = rng.integers(0, N), rng.integers(0, N)
            a, e, rows, nrot = ring_fold_tridiag(d, h, corner if N > 2 else 0.0, rows=(a_, b_))
            T = np.diag(a) + np.diag(e[:N - 1], 1) + np.diag(e[:N - 1], -1)
            lam, S = np.linalg.eigh(T)
            err_l = np.abs(lam - np.linalg.eigvalsh(Hm)).max()
            Tt = rng.uniform(2, 30)
            qi, qo = rows[a_] @ S, rows[b_] @ S
            amp = (qo * np.conj(qi) * np.exp(-1j * Tt * lam)).sum()
            want = expm(-1j * Tt * Hm)[b_, a_]
            worst = max(worst, err_l, abs(abs(amp) ** 2 - abs(want) ** 2))
        print(N, "rotations", nrot, "worst so far %.2e" % worst)
