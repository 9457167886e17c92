// Per-sample arithmetic of the RING-topology fidelity kernel (one sample per lane).
//
// With `topo="ring"` (noise_model.py:83-85: HH[N-1,0] = HH[0,N-1] = 1) the perturbed Hamiltonian is Hermitian
// "periodic tridiagonal": the chain's diagonal and complex nearest-neighbour couplings plus one corner element.  The
// diagonal gauge that makes the chain real (tridiag_core.h) leaves the total phase of the loop on the corner, so the
// real-tridiagonal trick does not apply.  Here the complex Hermitian N x N matrix is held in registers (lower triangle)
// and reduced to REAL symmetric tridiagonal form by N - 2 Householder reflections (the unblocked LAPACK zhetd2 scheme,
// restated for fully unrolled compile-time indices), carrying rows `in` and `out` of the accumulated unitary Q
// (A = Q T Q^H); the last sub-diagonal entry is made real by a phase on the last column of Q.  Then the SAME implicit
// QL iteration as the chain kernel (tridiag_ql2_fast with four real row planes = two complex rows) gives
//     U[out,in] = sum_k (Q S)[out,k] conj((Q S)[in,k]) exp(-i T lambda_k),          T = S Lambda S^T.
// ~ (4/3) N^3 real flops for the reduction on top of the chain kernel's work: N <= 10 stays in registers.
//
// Plain C++ header shared by the HIP kernel and the host unit test (tests/host/host_core.cpp).
#pragma once
#include "tridiag_core.h"

namespace rc {

template <int N>
struct HermLower {          // A[i][j], i >= j; im[i][i] is identically zero and never touched
    double re[N][N];
    double im[N][N];
};

// Householder tridiagonalisation; on return s.d / s.e hold the real tridiagonal matrix and s.z[0..3] the rows `in`
// (re, im) and `out` (re, im) of Q.  Exact-zero columns (already tridiagonal, e.g. a chain passed in) take tau = 0.
template <int N>
RC_HD void hermitian_tridiag_rows(HermLower<N>& A, int in, int out, TriEig<N, 4>& s) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        s.z[0][i] = (i == in) ? 1.0 : 0.0;
        s.z[1][i] = 0.0;
        s.z[2][i] = (i == out) ? 1.0 : 0.0;
        s.z[3][i] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < N - 2; ++k) {
        constexpr int kDummy = 0;
        (void)kDummy;
        const int m = N - 1 - k;                      // length of the column below the diagonal (compile time after unrolling)
        const double alphr = A.re[k + 1][k], alphi = A.im[k + 1][k];
        double xn2 = 0.0;
#pragma unroll
        for (int j = 1; j < N; ++j)
            if (j < m) xn2 = fma(A.re[k + 1 + j][k], A.re[k + 1 + j][k], fma(A.im[k + 1 + j][k], A.im[k + 1 + j][k], xn2));
        const bool triv = (xn2 == 0.0) && (alphi == 0.0);
        double nrm, inrm;
        sqrt_rsqrt(fma(alphr, alphr, fma(alphi, alphi, xn2 + 1e-300)), nrm, inrm);
        const double beta = -copysign(nrm, alphr), ibeta = -copysign(inrm, alphr);
        double taur = (beta - alphr) * ibeta, taui = -alphi * ibeta;
        // 1 / (alpha - beta): |alpha_r - beta| >= |beta| > 0
        const double denr = alphr - beta;
        const double rden = rcp_full(fma(denr, denr, alphi * alphi));
        const double scr = denr * rden, sci = -alphi * rden;
        double vr[N], vi[N];                          // v[0] = 1, v[j] = x[j] / (alpha - beta)
        vr[0] = 1.0;
        vi[0] = 0.0;
#pragma unroll
        for (int j = 1; j < N; ++j)
            if (j < m) {
                const double xr = A.re[k + 1 + j][k], xi = A.im[k + 1 + j][k];
                vr[j] = fma(xr, scr, -xi * sci);
                vi[j] = fma(xr, sci, xi * scr);
            }
        if (triv) {
            taur = 0.0;
            taui = 0.0;
        }
        s.d[k] = A.re[k][k];
        s.e[k] = triv ? alphr : beta;
        // p = tau * A22 * v   (A22 = trailing Hermitian block, rows / columns k+1 .. N-1)
        double pr[N], pi[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < m) {
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j < m) {
                        // element (i, j) of A22: stored for i >= j, conj of (j, i) otherwise; real on the diagonal
                        const double ar = (i >= j) ? A.re[k + 1 + i][k + 1 + j] : A.re[k + 1 + j][k + 1 + i];
                        const double ai = (i == j) ? 0.0 : ((i > j) ? A.im[k + 1 + i][k + 1 + j] : -A.im[k + 1 + j][k + 1 + i]);
                        sr = fma(ar, vr[j], sr);
                        si = fma(ar, vi[j], si);
                        if (i != j) {
                            sr = fma(-ai, vi[j], sr);
                            si = fma(ai, vr[j], si);
                        }
                    }
                pr[i] = fma(taur, sr, -taui * si);
                pi[i] = fma(taur, si, taui * sr);
            }
        // alpha2 = -(1/2) tau (p^H v);  w = p + alpha2 v
        double dr = 0.0, di = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < m) {
                dr = fma(pr[i], vr[i], fma(pi[i], vi[i], dr));
                di = fma(pr[i], vi[i], fma(-pi[i], vr[i], di));
            }
        const double a2r = -0.5 * fma(taur, dr, -taui * di), a2i = -0.5 * fma(taur, di, taui * dr);
        double wr[N], wi[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < m) {
                wr[i] = fma(a2r, vr[i], fma(-a2i, vi[i], pr[i]));
                wi[i] = fma(a2r, vi[i], fma(a2i, vr[i], pi[i]));
            }
        // A22 -= v w^H + w v^H   (lower triangle; the diagonal stays real)
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (i < m && j <= i) {
                    // (v_i conj(w_j) + w_i conj(v_j))
                    const double ur = fma(vr[i], wr[j], fma(vi[i], wi[j], fma(wr[i], vr[j], wi[i] * vi[j])));
                    A.re[k + 1 + i][k + 1 + j] -= ur;
                    if (i != j) {
                        const double ui = fma(vi[i], wr[j], fma(-vr[i], wi[j], fma(wi[i], vr[j], -wr[i] * vi[j])));
                        A.im[k + 1 + i][k + 1 + j] -= ui;
                    }
                }
        // rows of Q:  r <- r (I - tau v v^H)  on the trailing entries
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (j < m) {
                    sr = fma(s.z[q][k + 1 + j], vr[j], fma(-s.z[q + 1][k + 1 + j], vi[j], sr));
                    si = fma(s.z[q][k + 1 + j], vi[j], fma(s.z[q + 1][k + 1 + j], vr[j], si));
                }
            const double tr = fma(taur, sr, -taui * si), ti = fma(taur, si, taui * sr);
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (j < m) {
                    // r_j -= t * conj(v_j)
                    s.z[q][k + 1 + j] -= fma(tr, vr[j], ti * vi[j]);
                    s.z[q + 1][k + 1 + j] -= fma(ti, vr[j], -tr * vi[j]);
                }
        }
    }
    s.d[N - 2] = A.re[N - 2][N - 2];
    s.d[N - 1] = A.re[N - 1][N - 1];
    // last sub-diagonal entry z -> |z| by a phase on the last column of Q (the 1e-150 nudge makes the phase 1 for z = 0)
    const double zr = A.re[N - 1][N - 2] + 1e-150, zi = A.im[N - 1][N - 2];
    double az, iaz;
    sqrt_rsqrt(fma(zr, zr, zi * zi), az, iaz);
    s.e[N - 2] = az;
    s.e[N - 1] = 0.0;
    const double phr = zr * iaz, phi = zi * iaz;
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
        const double rr = s.z[q][N - 1], ri = s.z[q + 1][N - 1];
        s.z[q][N - 1] = fma(rr, phr, -ri * phi);
        s.z[q + 1][N - 1] = fma(rr, phi, ri * phr);
    }
}

// |sum_k zo_k conj(zi_k) exp(-i T lambda_k)|^2 from the QL result (rows in s.z[0..3], eigenvalues in s.d)
template <int N>
RC_HD double complex_rows_fidelity(const TriEig<N, 4>& s, double T, const double* sctab) {
    const double Tk = T * kTurnsPerRadian;
    // weights w_k = zo_k conj(zi_k)
    double re = fma(s.z[2][0], s.z[0][0], s.z[3][0] * s.z[1][0]);
    double im = fma(s.z[3][0], s.z[0][0], -s.z[2][0] * s.z[1][0]);
#pragma unroll
    for (int k = 1; k < N; ++k) {
        const double wr = fma(s.z[2][k], s.z[0][k], s.z[3][k] * s.z[1][k]);
        const double wi = fma(s.z[3][k], s.z[0][k], -s.z[2][k] * s.z[1][k]);
        double sk, ck;
        if (kTableSinCos) sincos_table(Tk * (s.d[k] - s.d[0]), sctab, sk, ck);
        else sincos_reduced(T * (s.d[k] - s.d[0]), sk, ck);
        // (wr + i wi)(c - i s)
        re = fma(wr, ck, fma(wi, sk, re));
        im = fma(wi, ck, fma(-wr, sk, im));
    }
    return fma(re, re, im * im);
}

// Fidelity of one ring sample - fast path.  loadg(j): this sample's j-th draw (g0_i, g1_i, g2_i), i = 0..N-1; corner =
// static weight of the ring closure (1.0, noise_model.py:84-85).  Returns false - per sample - when the QL sweep cap
// was hit (the caller recomputes that sample with ring_fidelity_general).
template <int N, typename LoadG>
RC_HD bool ring_fidelity_fast(const double* x, const double* h0d, const double* h0o, double corner, LoadG loadg, int in,
                              int out, const double* sctab, double& fid) {
    static_assert(N >= 3, "a ring needs three sites (N = 2: the closure coincides with the chain bond)");
    HermLower<N> A;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            A.re[i][j] = 0.0;
            A.im[i][j] = 0.0;
        }
#pragma unroll
    for (int i = 0; i < N; ++i) A.re[i][i] = x[i] + h0d[i] + loadg(3 * i);
#pragma unroll
    for (int i = 1; i < N; ++i) {
        A.re[i][i - 1] = h0o[i - 1] + loadg(3 * i + 1);          // z[i][i-1] = g1 + i g2 (noise_model.py:141-143)
        A.im[i][i - 1] = loadg(3 * i + 2);
    }
    A.re[N - 1][0] += corner;
    TriEig<N, 4> s;
    hermitian_tridiag_rows<N>(A, in, out, s);
    const bool ok = tridiag_ql2_fast(s);
    fid = complex_rows_fidelity<N>(s, fabs(x[N]), sctab);
    return ok;
}

// GENERAL PATH of the QL iteration with R row vectors (rare): tridiag_ql2_general with any number of rows.
template <typename Vec, int R>
RC_HD void tridiag_ql_general_rows(int n, Vec d, Vec e, Vec (&z)[R]) {
    for (int l = 0; l < n - 1; ++l) {
        for (int iter = 0; iter < kMaxSweepsPerEig; ++iter) {
            int m = l;
            for (; m < n - 1; ++m) {
                const double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) <= kEps * dd) break;
            }
            if (m == l) break;
            const double delta = 0.5 * (d[l + 1] - d[l]);
            const double e2 = e[l] * e[l];
            const double rho = sqrt_fast(fma(delta, delta, e2) + 1e-300);
            double g = d[m] - d[l] + e2 * rcp_fast(delta + copysign(rho, delta));
            double sn = 1.0, cs = 1.0, p = 0.0;
            for (int i = m - 1; i >= l; --i) {
                const double f = sn * e[i];
                const double b = cs * e[i];
                const double gn = g + 1e-150;
                double r, rinv;
                sqrt_rsqrt(fma(f, f, gn * gn), r, rinv);
                e[i + 1] = r;
                sn = f * rinv;
                cs = gn * rinv;
                g = d[i + 1] - p;
                r = fma(d[i] - g, sn, 2.0 * cs * b);
                p = sn * r;
                d[i + 1] = g + p;
                g = fma(cs, r, -b);
                for (int q = 0; q < R; ++q) {
                    const double a1 = z[q][i + 1], a0 = z[q][i];
                    z[q][i + 1] = fma(sn, a0, cs * a1);
                    z[q][i] = fma(cs, a0, -sn * a1);
                }
            }
            d[l] = d[l] - p;
            e[l] = g;
            e[m] = 0.0;
        }
    }
}

// Fidelity of one ring sample - general path: the same Householder reduction in registers, then the textbook
// per-sample QL with its vectors (d, e, four rows: 6 N doubles) in caller-provided storage.
template <int N, typename LoadG, typename Vec>
RC_HD double ring_fidelity_general(const double* x, const double* h0d, const double* h0o, double corner, LoadG loadg,
                                   int in, int out, Vec d, Vec e, Vec (&z)[4]) {
    HermLower<N> A;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            A.re[i][j] = 0.0;
            A.im[i][j] = 0.0;
        }
#pragma unroll
    for (int i = 0; i < N; ++i) A.re[i][i] = x[i] + h0d[i] + loadg(3 * i);
#pragma unroll
    for (int i = 1; i < N; ++i) {
        A.re[i][i - 1] = h0o[i - 1] + loadg(3 * i + 1);
        A.im[i][i - 1] = loadg(3 * i + 2);
    }
    A.re[N - 1][0] += corner;
    TriEig<N, 4> s;
    hermitian_tridiag_rows<N>(A, in, out, s);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        d[i] = s.d[i];
        e[i] = s.e[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) z[q][i] = s.z[q][i];
    }
    tridiag_ql_general_rows<Vec, 4>(N, d, e, z);
    const double T = fabs(x[N]);
    double re = 0.0, im = 0.0;
    for (int k = 0; k < N; ++k) {
        const double wr = fma(z[2][k], z[0][k], z[3][k] * z[1][k]);
        const double wi = fma(z[3][k], z[0][k], -z[2][k] * z[1][k]);
        double sk, ck;
        sincos_reduced(T * d[k], sk, ck);
        re = fma(wr, ck, fma(wi, sk, re));
        im = fma(wi, ck, fma(-wr, sk, im));
    }
    return fma(re, re, im * im);
}

}  // namespace rc
