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
#include <type_traits>

namespace rc {

// Up to this size the lane-per-sample ring routes hold the dense lower triangle in registers (Householder reduction); above,
// the folded band reduction further down (ring_fold_*, round 5).
#ifndef RC_RING_DENSE_MAX_N
#define RC_RING_DENSE_MAX_N 10
#endif
constexpr int kRingDenseMaxN = RC_RING_DENSE_MAX_N;
template <int N>
RC_HD void ring_fold_tridiag_rows(const double (&d)[N], const double (&hr)[N], const double (&hi)[N], double corner, int in, int out,
                                  TriEig<N, 4>& s);
template <int N>
RC_HD void ring_fold_tridiag_f32(const float (&d)[N], const float (&hr)[N], const float (&hi)[N], float corner, float (&df)[N],
                                 float (&ef)[N]);

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
// FREEZE: tridiag_ql2_fast's - the result of a lane independent of its wave neighbours (the repair kernel).
template <int N, bool FREEZE = false, typename LoadG>
RC_HD bool ring_fidelity_fast(const double* x, const double* h0d, const double* h0o, double corner, LoadG loadg, int in,
                              int out, const double* sctab, double& fid) {
    static_assert(N >= 3, "a ring needs three sites (N = 2: the closure coincides with the chain bond)");
    TriEig<N, 4> s;
    if constexpr (N > kRingDenseMaxN) {            // the folded band reduction: 10 N numbers of state instead of N^2
        double d[N], hr[N], hi[N];
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = x[i] + h0d[i] + loadg(3 * i);
#pragma unroll
        for (int i = 1; i < N; ++i) {
            hr[i - 1] = h0o[i - 1] + loadg(3 * i + 1);
            hi[i - 1] = loadg(3 * i + 2);
        }
        hr[N - 1] = 0.0;
        hi[N - 1] = 0.0;
        ring_fold_tridiag_rows<N>(d, hr, hi, corner, in, out, s);
    } else {
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
        hermitian_tridiag_rows<N>(A, in, out, s);
    }
    const bool ok = tridiag_ql2_fast<N, 4, FREEZE>(s);
    fid = complex_rows_fidelity<N>(s, fabs(x[N]), sctab);
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------
// Mixed-precision eigenvalue route for the ring (round 3) - the chain kernel's scheme carried over:
//   * the diagonal gauge of tridiag_core.h leaves a real tridiagonal part (d, e_i = |h_i|) and puts the loop's phase on the
//     corner; the spectrum depends on that phase only through ONE constant:
//         chi_ring(lam) = P_{0..N-1}(lam) - c^2 P_{1..N-2}(lam) - Phi,     Phi = 2 c Re(prod_i h_i)
//     (P_{a..b} = characteristic polynomial of the open chain on sites a..b, c = corner weight, h_i = H[i+1,i]);
//   * starting values: the complex Hermitian matrix in fp32 -> Householder tridiagonalisation WITHOUT the rows of Q,
//     exploiting the sparsity of a periodic tridiagonal matrix (at step k the column below the diagonal has entries only
//     in its first and its last k+1 rows, the trailing block is band + a dense block of its last k+1 rows + the first row
//     against the last k) -> the fp32 QL of the chain kernels: every eigenvalue to ~1e-6 of the scale;
//   * one fp64 Halley step per eigenvalue on chi_ring (two coupled three-term recurrences, 13 operations per site),
//     acceptance / stepping exactly as for the chain (mixed_refine);
//   * weights w_k = u_k[out] conj(u_k[in]) from the adjugate of (lam_k - H): the two paths around the ring,
//         w_k = [ A P_wrap(lam_k) + B P_between(lam_k) ] / prod_{m != k}(lam_k - lam_m)
//     A = product of the couplings lo -> hi along the chain, P_wrap = characteristic polynomial of the sites outside
//     [lo, hi] (one open chain through the corner), B = c x conj(product of the other couplings) (the way round through
//     the corner), P_between = characteristic polynomial of the sites strictly between; conjugated when out < in.
// A sample that is not settled, or has a pair closer than kRingGapTol of the scale, reports false: the kernel puts it on a
// list and the all-fp64 route above recomputes the listed samples, lane per sample (mc_fid_ring_repair_kernel).
// scripts/proto/ring_formulas.py checks the formulas against dense eigh.
// ---------------------------------------------------------------------------------------------------------------
// Smallest eigenvalue gap (relative to the spectral scale) the two-path weights are trusted with.  Host scan over
// translation-invariant rings whose +-k pairs are split by noise of every size (N = 3 .. 10): |dF| = 1.3e-10 at a splitting
// of 2e-4, 7e-12 at 2e-3, ~1 / gap; the fuzz campaign of round 3 found 6e-10 on the GPU with the 4e-6 of the chain route.
constexpr double kRingGapTol = 1e-3;

template <int N>
struct RingChi {
    const double (&d0)[N];
    const double (&e0sq)[N];
    double c2, phi;
    RC_HD double trace() const {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) t += d0[i];
        return t;
    }
    RC_HD void eval(const double mu, double& p_out, double& dp_out, double& q_out) const {
        double pm = 1.0, p = mu - d0[0], dm = 0.0, dp = 1.0, qm = 0.0, q = 0.0;             // P_{0..m}
        double im = 1.0, ip = 1.0, idm = 0.0, idp = 0.0, iqm = 0.0, iq = 0.0;               // P_{1..m}
#pragma unroll
        for (int m = 1; m < N; ++m) {
            const double t = mu - d0[m];
            const double c = e0sq[m - 1];
            const double pn = fma(t, p, -c * pm);
            const double dn = fma(t, dp, fma(-c, dm, p));
            const double qn = fma(t, q, fma(-c, qm, dp));
            pm = p; p = pn;
            dm = dp; dp = dn;
            qm = q; q = qn;
            if (m == 1) {                          // P_{1..1} = mu - d_1
                ip = t;
                idp = 1.0;
            } else if (m <= N - 2) {
                const double in_ = fma(t, ip, -c * im);
                const double idn = fma(t, idp, fma(-c, idm, ip));
                const double iqn = fma(t, iq, fma(-c, iqm, idp));
                im = ip; ip = in_;
                idm = idp; idp = idn;
                iqm = iq; iq = iqn;
            }
        }
        p_out = fma(-c2, ip, p) - phi;
        dp_out = fma(-c2, idp, dp);
        q_out = fma(-c2, iq, q);
    }
    RC_HD void eval2(const double mu, double& p_out, double& dp_out) const {      // chi and chi' only
        double pm = 1.0, p = mu - d0[0], dm = 0.0, dp = 1.0;
        double im = 1.0, ip = 1.0, idm = 0.0, idp = 0.0;
#pragma unroll
        for (int m = 1; m < N; ++m) {
            const double t = mu - d0[m];
            const double c = e0sq[m - 1];
            const double pn = fma(t, p, -c * pm);
            const double dn = fma(t, dp, fma(-c, dm, p));
            pm = p; p = pn;
            dm = dp; dp = dn;
            if (m == 1) {
                ip = t;
                idp = 1.0;
            } else if (m <= N - 2) {
                const double in_ = fma(t, ip, -c * im);
                const double idn = fma(t, idp, fma(-c, idm, ip));
                im = ip; ip = in_;
                idm = idp; idp = idn;
            }
        }
        p_out = fma(-c2, ip, p) - phi;
        dp_out = fma(-c2, idp, dp);
    }
};

// sparsity of the Householder reduction of a periodic tridiagonal matrix (relative indices within the trailing block of
// size m = N - 1 - k at step k; all arguments are compile-time constants after unrolling)
constexpr bool ring_col_nz(int N, int k, int j) { return j == 0 || j >= (N - 1 - k) - 1 - k; }
constexpr bool ring_blk_nz(int N, int k, int i, int j) {
    const int m = N - 1 - k;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    if (hi - lo <= 1) return true;                                  // band
    if (lo >= m - 1 - k) return true;                               // dense block of the last k+1 rows
    return lo == 0 && hi >= m - k;                                  // first row against the last k
}
constexpr bool ring_p_nz(int N, int k, int i) {
    for (int j = 0; j < N - 1 - k; ++j)
        if (ring_blk_nz(N, k, i, j) && ring_col_nz(N, k, j)) return true;
    return false;
}
constexpr bool ring_w_nz(int N, int k, int i) { return ring_p_nz(N, k, i) || ring_col_nz(N, k, i); }

template <int N>
struct HermLowerF {         // fp32 twin of HermLower
    float re[N][N];
    float im[N][N];
};

// compile-time loop: f(std::integral_constant<int, B>{}), ..., f(std::integral_constant<int, E - 1>{}) - the sparsity
// predicates above must be evaluated AT COMPILE TIME (`if constexpr`): left to the optimiser, N >= 9 kept them as run-time
// tests and the work arrays went to scratch memory
template <int B, int E, typename F>
RC_HD void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// fp32 Householder tridiagonalisation of the periodic tridiagonal Hermitian matrix in A (lower triangle; entries outside
// the periodic pattern are never read) -> real tridiagonal (d, e), eigenvalues only.
template <int N>
RC_HD void ring_tridiag_f32(HermLowerF<N>& A, float (&d)[N], float (&e)[N]) {
    static_for<0, N - 2>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value;
        constexpr int m = N - 1 - k;
        const float alphr = A.re[k + 1][k], alphi = A.im[k + 1][k];
        float xn2 = 0.0f;
        static_for<1, m>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            if constexpr (ring_col_nz(N, k, j))
                xn2 = fmaf(A.re[k + 1 + j][k], A.re[k + 1 + j][k], fmaf(A.im[k + 1 + j][k], A.im[k + 1 + j][k], xn2));
        });
        // |x|^2 > 0 always (the nudge): tau = 0 can only come out of alpha_i = xn2 = 0 exactly, where v = (1, 0..) anyway
        const float h = fmaf(alphr, alphr, fmaf(alphi, alphi, xn2 + 1e-30f));
        const float inrm = seed_rsqf(h);
        const float nrm = h * inrm;
        const float beta = -copysignf(nrm, alphr), ibeta = -copysignf(inrm, alphr);
        const float taur = (beta - alphr) * ibeta, taui = -alphi * ibeta;
        const float denr = alphr - beta;
        const float rden = seed_rcpf(fmaf(denr, denr, alphi * alphi));
        const float scr = denr * rden, sci = -alphi * rden;
        float vr[m], vi[m];                          // v[0] = 1, v[j] = x[j] / (alpha - beta) on the column's pattern
        vr[0] = 1.0f;
        vi[0] = 0.0f;
        static_for<1, m>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            if constexpr (ring_col_nz(N, k, j)) {
                const float xr = A.re[k + 1 + j][k], xi = A.im[k + 1 + j][k];
                vr[j] = fmaf(xr, scr, -xi * sci);
                vi[j] = fmaf(xr, sci, xi * scr);
            }
        });
        d[k] = A.re[k][k];
        e[k] = beta;
        // p = tau * A22 * v
        float pr[m], pi[m];
        static_for<0, m>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            if constexpr (ring_p_nz(N, k, i)) {
                float sr = 0.0f, si = 0.0f;
                static_for<0, m>([&](auto Jc) {
                    constexpr int j = decltype(Jc)::value;
                    if constexpr (ring_col_nz(N, k, j) && ring_blk_nz(N, k, i, j)) {
                        const float ar = (i >= j) ? A.re[k + 1 + i][k + 1 + j] : A.re[k + 1 + j][k + 1 + i];
                        sr = fmaf(ar, vr[j], sr);
                        si = fmaf(ar, vi[j], si);
                        if constexpr (i != j) {
                            const float ai = (i > j) ? A.im[k + 1 + i][k + 1 + j] : -A.im[k + 1 + j][k + 1 + i];
                            sr = fmaf(-ai, vi[j], sr);
                            si = fmaf(ai, vr[j], si);
                        }
                    }
                });
                pr[i] = fmaf(taur, sr, -taui * si);
                pi[i] = fmaf(taur, si, taui * sr);
            }
        });
        // alpha2 = -(1/2) tau (p^H v);  w = p + alpha2 v
        float dr = 0.0f, di = 0.0f;
        static_for<0, m>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            if constexpr (ring_p_nz(N, k, i) && ring_col_nz(N, k, i)) {
                dr = fmaf(pr[i], vr[i], fmaf(pi[i], vi[i], dr));
                di = fmaf(pr[i], vi[i], fmaf(-pi[i], vr[i], di));
            }
        });
        const float a2r = -0.5f * fmaf(taur, dr, -taui * di), a2i = -0.5f * fmaf(taur, di, taui * dr);
        float wr[m], wi[m];
        static_for<0, m>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            if constexpr (ring_w_nz(N, k, i)) {
                float p_r = 0.0f, p_i = 0.0f;
                if constexpr (ring_p_nz(N, k, i)) {
                    p_r = pr[i];
                    p_i = pi[i];
                }
                if constexpr (ring_col_nz(N, k, i)) {
                    wr[i] = fmaf(a2r, vr[i], fmaf(-a2i, vi[i], p_r));
                    wi[i] = fmaf(a2r, vi[i], fmaf(a2i, vr[i], p_i));
                } else {
                    wr[i] = p_r;
                    wi[i] = p_i;
                }
            }
        });
        // A22 -= v w^H + w v^H   (lower triangle, only where a term is structurally non-zero; an entry outside the old
        // pattern starts from zero)
        static_for<0, m>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            static_for<0, i + 1>([&](auto Jc) {
                constexpr int j = decltype(Jc)::value;
                constexpr bool t1 = ring_col_nz(N, k, i) && ring_w_nz(N, k, j);      // v_i conj(w_j)
                constexpr bool t2 = ring_w_nz(N, k, i) && ring_col_nz(N, k, j);      // w_i conj(v_j)
                if constexpr (t1 || t2) {
                    float ur = 0.0f, ui = 0.0f;
                    if constexpr (t1) {
                        ur = fmaf(vr[i], wr[j], vi[i] * wi[j]);
                        ui = fmaf(vi[i], wr[j], -vr[i] * wi[j]);
                    }
                    if constexpr (t2) {
                        ur = fmaf(wr[i], vr[j], fmaf(wi[i], vi[j], ur));
                        ui = fmaf(wi[i], vr[j], fmaf(-wr[i], vi[j], ui));
                    }
                    constexpr bool had = ring_blk_nz(N, k, i, j);
                    A.re[k + 1 + i][k + 1 + j] = had ? A.re[k + 1 + i][k + 1 + j] - ur : -ur;
                    if constexpr (i != j) A.im[k + 1 + i][k + 1 + j] = had ? A.im[k + 1 + i][k + 1 + j] - ui : -ui;
                }
            });
        });
    });
    d[N - 2] = A.re[N - 2][N - 2];
    d[N - 1] = A.re[N - 1][N - 1];
    const float zr = A.re[N - 1][N - 2], zi = A.im[N - 1][N - 2];
    const float hz = fmaf(zr, zr, fmaf(zi, zi, 1e-30f));
    e[N - 2] = hz * seed_rsqf(hz);
    e[N - 1] = 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// FOLDED BAND REDUCTION of the ring (round 5): the route for N > kRingDenseMaxN, where the dense lower triangle of the
// Householder reduction (N^2 numbers per sample) no longer fits the register file.
//
// In the site order 0, N-1, 1, N-2, 2, ... (position 2j <-> site j, position 2j+1 <-> site N-1-j) a periodic tridiagonal
// Hermitian matrix is PENTADIAGONAL: the chain's bonds sit at distance 2, the corner element at (1, 0) and the middle bond at
// (N-1, N-2); every other entry at distance 1 starts as zero.  Schwarz' band reduction takes it to tridiagonal form: for
// k = 0 .. N-3 a unitary 2 x 2 rotation of the planes (k+1, k+2) annihilates B[k+2][k] and pushes a bulge to (k+4, k+1), which
// rotations of (k+3, k+4), (k+5, k+6), ... chase off the matrix - N^2/4 rotations in all (N = 11: 25, N = 16: 56), each touching
// O(1) entries of three diagonals: O(N^2) work and 6 N numbers of state (+ 4 N for two rows of the accumulated unitary),
// against O(N^3) and N^2.  The rotation U = [[p, q], [-conj q, conj p]], p = conj(x)/rho, q = conj(y)/rho maps (x, y) to
// (rho, 0) with rho REAL, so the sub-diagonal comes out real wherever something was eliminated; what is left complex (the last
// entry; an entry next to an exact zero) is made real by a diagonal unitary at the end.  scripts/proto/ring_fold_band.py is the
// numpy prototype (eigenvalues and U[out, in] against eigvalsh / expm, N = 3 .. 16, cut bonds, flat diagonals).
// ---------------------------------------------------------------------------------------------------------------
constexpr int ring_fold_site(int N, int p) { return (p % 2 == 0) ? p / 2 : N - 1 - p / 2; }
constexpr int ring_fold_pos(int N, int s) { return (s < (N + 1) / 2) ? 2 * s : 2 * (N - 1 - s) + 1; }

RC_HD void fold_rsqrt(double x, double& root, double& inv) { sqrt_rsqrt(x, root, inv); }
RC_HD void fold_rsqrt(float x, float& root, float& inv) {
    inv = seed_rsqf(x);
    root = x * inv;
}

// The folded band (lower part): a[p] real diagonal, (br, bi)[p] = B[p+1][p], (cr, ci)[p] = B[p+2][p].
template <int N, typename T>
struct RingBand {
    T a[N], br[N], bi[N], cr[N], ci[N];
};

// d[site], (hr, hi)[j] = H[j+1][j], corner = H[N-1][0] (real) -> the folded band
template <int N, typename T>
RC_HD void ring_fold_load(const T (&d)[N], const T (&hr)[N], const T (&hi)[N], T corner, RingBand<N, T>& B) {
    // element H[i][j] of the ring for compile-time sites i != j (0 when they are not neighbours)
    auto Hre = [&](int i, int j) -> T {
        return (i == j + 1) ? hr[j] : ((j == i + 1) ? hr[i] : (((i == N - 1 && j == 0) || (i == 0 && j == N - 1)) ? corner : (T)0));
    };
    auto Him = [&](int i, int j) -> T { return (i == j + 1) ? hi[j] : ((j == i + 1) ? -hi[i] : (T)0); };
#pragma unroll
    for (int p = 0; p < N; ++p) {
        B.a[p] = d[ring_fold_site(N, p)];
        B.br[p] = (p + 1 < N) ? Hre(ring_fold_site(N, p + 1), ring_fold_site(N, p)) : (T)0;
        B.bi[p] = (p + 1 < N) ? Him(ring_fold_site(N, p + 1), ring_fold_site(N, p)) : (T)0;
        B.cr[p] = (p + 2 < N) ? Hre(ring_fold_site(N, p + 2), ring_fold_site(N, p)) : (T)0;
        B.ci[p] = (p + 2 < N) ? Him(ring_fold_site(N, p + 2), ring_fold_site(N, p)) : (T)0;
    }
}

// The reduction.  ROWS = true: (zr, zi)[0 / 1][p] hold columns pos(in) / pos(out) of the accumulated transformation (start:
// unit vectors) - row r of Q is their conjugate.  On return a[] and (br, bi)[] hold the tridiagonal matrix (sub-diagonal
// possibly complex where nothing was eliminated: ring_fold_finish).
// coefficients of the rotation U = [[p, q], [-conj q, conj p]] that maps (x, y) to (rho, 0): p = conj(x) / rho, q = conj(y) / rho;
// y = 0 exactly (a cut bond): the identity, and x stays what it is (possibly complex: ring_fold_finish)
template <typename T>
struct FoldRot {
    T pr, pi, qr, qi;          // p, q
    T nr, ni;                  // what x becomes: rho (real) - or x itself for the identity
};
template <typename T>
RC_HD FoldRot<T> ring_fold_rot(T xr, T xi, T yr, T yi) {
    const T tiny = std::is_same<T, float>::value ? (T)1e-30f : (T)1e-300;
    const bool triv = (yr == (T)0) && (yi == (T)0);
    T rho, inv;
    fold_rsqrt(fma(xr, xr, fma(xi, xi, fma(yr, yr, fma(yi, yi, tiny)))), rho, inv);
    FoldRot<T> r;
    r.pr = triv ? (T)1 : xr * inv;
    r.pi = triv ? (T)0 : -xi * inv;
    r.qr = triv ? (T)0 : yr * inv;
    r.qi = triv ? (T)0 : -yi * inv;
    r.nr = triv ? xr : rho;
    r.ni = triv ? xi : (T)0;
    return r;
}

// The reduction.  ROWS = true: (zr, zi)[0 / 1][p] hold columns pos(in) / pos(out) of the accumulated transformation (start:
// unit vectors) - row r of Q is their conjugate.  On return a[] and (br, bi)[] hold the tridiagonal matrix (sub-diagonal
// possibly complex where nothing was eliminated: ring_fold_finish).
template <int N, typename T, bool ROWS>
RC_HD void ring_fold_reduce(RingBand<N, T>& B, T (&zr)[2][N], T (&zi)[2][N]) {
#pragma unroll
    for (int k = 0; k < N - 2; ++k) {
        // annihilate B[k+2][k] with a rotation of the plane (k+1, k+2)
        FoldRot<T> u = ring_fold_rot<T>(B.br[k], B.bi[k], B.cr[k], B.ci[k]);
        B.br[k] = u.nr;
        B.bi[k] = u.ni;
        B.cr[k] = (T)0;
        B.ci[k] = (T)0;
#pragma unroll
        for (int i = k + 1; i < N - 1; i += 2) {   // the rotation `u` acts on the plane (i, i+1)
            const T pr = u.pr, pi = u.pi, qr = u.qr, qi = u.qi;
            // 2 x 2 diagonal block [[al, conj be], [be, de]] -> U (.) U^H
            const T al = B.a[i], de = B.a[i + 1], ber = B.br[i], bei = B.bi[i];
            const T qbr = fma(qr, ber, -qi * bei), qbi = fma(qr, bei, qi * ber);          // q beta
            const T t = (T)2 * fma(qbr, pr, qbi * pi);                                     // 2 Re(q beta conj p)
            const T n11 = fma(fma(pr, pr, pi * pi), al, fma(fma(qr, qr, qi * qi), de, t));
            B.a[i] = n11;
            B.a[i + 1] = al + de - n11;
            // beta' = conj(p) conj(q) (de - al) + conj(p)^2 beta - conj(q)^2 conj(beta)
            const T pqr = fma(pr, qr, -pi * qi), pqi = -fma(pr, qi, pi * qr);              // conj(p q)
            const T ppr = fma(pr, pr, -pi * pi), ppi = (T)-2 * pr * pi;                    // conj(p)^2
            const T qqr = fma(qr, qr, -qi * qi), qqi = (T)-2 * qr * qi;                    // conj(q)^2
            const T dd = de - al;
            B.br[i] = fma(pqr, dd, fma(ppr, ber, fma(-ppi, bei, -fma(qqr, ber, qqi * bei))));
            B.bi[i] = fma(pqi, dd, fma(ppr, bei, fma(ppi, ber, -fma(qqi, ber, -qqr * bei))));
            // column op on the rows below the block: [B[t][i], B[t][i+1]] <- [s, v] U^H = [s conj p + v conj q, -s q + v p]
            if (i + 2 < N) {                       // row i+2: (s, v) = (c[i], b[i+1])
                const T sr = B.cr[i], si = B.ci[i], vr = B.br[i + 1], vi = B.bi[i + 1];
                B.cr[i] = fma(sr, pr, fma(si, pi, fma(vr, qr, vi * qi)));
                B.ci[i] = fma(si, pr, fma(-sr, pi, fma(vi, qr, -vr * qi)));
                B.br[i + 1] = fma(-sr, qr, fma(si, qi, fma(vr, pr, -vi * pi)));
                B.bi[i + 1] = fma(-sr, qi, fma(-si, qr, fma(vr, pi, vi * pr)));
            }
            T bur = (T)0, bui = (T)0;              // the bulge this rotation makes at (i+3, i)
            if (i + 3 < N) {                       // row i+3: (s, v) = (0, c[i+1])
                const T vr = B.cr[i + 1], vi = B.ci[i + 1];
                bur = fma(vr, qr, vi * qi);        // v conj q
                bui = fma(vi, qr, -vr * qi);
                B.cr[i + 1] = fma(vr, pr, -vi * pi);
                B.ci[i + 1] = fma(vr, pi, vi * pr);
            }
            if (ROWS) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {      // [z_i; z_{i+1}] <- U [z_i; z_{i+1}]
                    const T ar = zr[r][i], ai = zi[r][i], cr_ = zr[r][i + 1], ci_ = zi[r][i + 1];
                    zr[r][i] = fma(pr, ar, fma(-pi, ai, fma(qr, cr_, -qi * ci_)));
                    zi[r][i] = fma(pr, ai, fma(pi, ar, fma(qr, ci_, qi * cr_)));
                    zr[r][i + 1] = fma(-qr, ar, fma(-qi, ai, fma(pr, cr_, pi * ci_)));     // -conj(q) z_i + conj(p) z_{i+1}
                    zi[r][i + 1] = fma(-qr, ai, fma(qi, ar, fma(pr, ci_, -pi * cr_)));
                }
            }
            if (i + 3 >= N) break;                 // (compile time) the bulge fell off the matrix
            // chase: the plane (i+2, i+3) takes (B[i+2][i], bulge) = (c[i], bulge) to (rho, 0); its ROW op mixes the two entries of
            // column i+1 in those rows, (b[i+1], c[i+1]) <- U (b[i+1], c[i+1])
            u = ring_fold_rot<T>(B.cr[i], B.ci[i], bur, bui);
            B.cr[i] = u.nr;
            B.ci[i] = u.ni;
            const T sr = B.br[i + 1], si = B.bi[i + 1], vr = B.cr[i + 1], vi = B.ci[i + 1];
            B.br[i + 1] = fma(u.pr, sr, fma(-u.pi, si, fma(u.qr, vr, -u.qi * vi)));
            B.bi[i + 1] = fma(u.pr, si, fma(u.pi, sr, fma(u.qr, vi, u.qi * vr)));
            B.cr[i + 1] = fma(-u.qr, sr, fma(-u.qi, si, fma(u.pr, vr, u.pi * vi)));
            B.ci[i + 1] = fma(-u.qr, si, fma(u.qi, sr, fma(u.pr, vi, -u.pi * vr)));
        }
    }
}

// The tridiagonal matrix of ring_fold_reduce with a REAL sub-diagonal: a diagonal unitary D (phases accumulated from the top:
// f_{p+1} = f_p b_p / |b_p|) takes b_p to |b_p|; e[] receives the moduli, and - ROWS - the rows of Q = U^H D land in s.z
// (row `in`: planes 0 / 1, row `out`: planes 2 / 3) the way complex_rows_fidelity reads them.
template <int N, typename T, bool ROWS>
RC_HD void ring_fold_finish(const RingBand<N, T>& B, const T (&zr)[2][N], const T (&zi)[2][N], T (&d)[N], T (&e)[N],
                            T (&rows)[ROWS ? 4 : 1][N]) {
    const T tiny = std::is_same<T, float>::value ? (T)1e-30f : (T)1e-300;
    T fr = (T)1, fi = (T)0;                        // phase of basis vector p
#pragma unroll
    for (int p = 0; p < N; ++p) {
        d[p] = B.a[p];
        if (ROWS) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {          // conj(z) * f
                rows[2 * r][p] = fma(zr[r][p], fr, zi[r][p] * fi);
                rows[2 * r + 1][p] = fma(zr[r][p], fi, -zi[r][p] * fr);
            }
        }
        if (p + 1 < N) {
            const T wr = fma(B.br[p], fr, -B.bi[p] * fi), wi = fma(B.br[p], fi, B.bi[p] * fr);     // b_p f_p
            T m, im;
            fold_rsqrt(fma(wr, wr, fma(wi, wi, tiny)), m, im);
            e[p] = m;
            const bool zero = (wr == (T)0) && (wi == (T)0);
            if (ROWS) {
                fr = zero ? (T)1 : wr * im;
                fi = zero ? (T)0 : wi * im;
            }
        } else {
            e[p] = (T)0;
        }
    }
}

// fp32 starting values of the ring's eigenvalues through the folded band (the mixed route for N > kRingDenseMaxN):
// d[site] (fp32), (hr, hi)[j] = H[j+1][j] -> real tridiagonal (df, ef) for tridiag_ql_f32
template <int N>
RC_HD void ring_fold_tridiag_f32(const float (&d)[N], const float (&hr)[N], const float (&hi)[N], float corner, float (&df)[N],
                                 float (&ef)[N]) {
    RingBand<N, float> B;
    ring_fold_load<N, float>(d, hr, hi, corner, B);
    float zr[2][N], zi[2][N], rows[1][N];
    ring_fold_reduce<N, float, false>(B, zr, zi);
    ring_fold_finish<N, float, false>(B, zr, zi, df, ef, rows);
}

// all-fp64, with rows `in` / `out` of Q: the counterpart of hermitian_tridiag_rows for N > kRingDenseMaxN
template <int N>
RC_HD void ring_fold_tridiag_rows(const double (&d)[N], const double (&hr)[N], const double (&hi)[N], double corner, int in, int out,
                                  TriEig<N, 4>& s) {
    RingBand<N, double> B;
    ring_fold_load<N, double>(d, hr, hi, corner, B);
    double zr[2][N], zi[2][N];
#pragma unroll
    for (int p = 0; p < N; ++p) {                  // unit vectors at the folded positions of `in` / `out` (wave-uniform)
        const int site = ring_fold_site(N, p);
        zr[0][p] = (site == in) ? 1.0 : 0.0;
        zr[1][p] = (site == out) ? 1.0 : 0.0;
        zi[0][p] = 0.0;
        zi[1][p] = 0.0;
    }
    ring_fold_reduce<N, double, true>(B, zr, zi);
    ring_fold_finish<N, double, true>(B, zr, zi, s.d, s.e, s.z);
}

// Fidelity of one ring sample - mixed-precision fast path.  Same arguments as ring_fidelity_fast.  Returns false - per
// sample - when the sample must be recomputed by the all-fp64 route.
template <int N, typename LoadG>
RC_HD bool ring_fidelity_mixed(const double* x, const double* h0d, const double* h0o, double corner, LoadG loadg, int in,
                               int out, const double* sctab, double& fid, int* extra_steps = nullptr) {
    static_assert(N >= 3, "a ring needs three sites");
    const int lo = in < out ? in : out, hi = in < out ? out : in;
    double d0[N], e0sq[N];
    constexpr bool FOLD = N > kRingDenseMaxN;      // fp32 starts through the folded band instead of the dense Householder
    HermLowerF<FOLD ? 1 : N> A;
    float fd[N], fhr[N], fhi[N];                   // (FOLD) the fp32 copy the band is loaded from
    // products of the complex couplings along the two ways from lo to hi: ar + i ai = prod_{lo <= i < hi} h_i, and
    // rr + i ri = the product of all the others
    double ar = 1.0, ai = 0.0, rr = 1.0, ri = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        d0[i] = x[i] + h0d[i] + loadg(3 * i);
        if constexpr (FOLD) fd[i] = (float)d0[i];
        else A.re[i][i] = (float)d0[i];
    }
#pragma unroll
    for (int i = 1; i < N; ++i) {
        const double hr = h0o[i - 1] + loadg(3 * i + 1);          // h_{i-1} = H[i][i-1] = g1 + i g2 (noise_model.py:141-143)
        const double hi_ = loadg(3 * i + 2);
        e0sq[i - 1] = fma(hr, hr, fma(hi_, hi_, 1e-300));
        if constexpr (FOLD) {
            fhr[i - 1] = (float)hr;
            fhi[i - 1] = (float)hi_;
        } else {
            A.re[i][i - 1] = (float)hr;
            A.im[i][i - 1] = (float)hi_;
        }
        if (i - 1 >= lo && i - 1 < hi) {                          // wave-uniform
            const double t = fma(ar, hr, -ai * hi_);
            ai = fma(ar, hi_, ai * hr);
            ar = t;
        } else {
            const double t = fma(rr, hr, -ri * hi_);
            ri = fma(rr, hi_, ri * hr);
            rr = t;
        }
    }
    e0sq[N - 1] = 0.0;
    // the guard's view of the original matrix, picked here so that d0 / e0sq die where they used to (tridiag_core.h)
    double g_slo = 0.0, g_shi = 0.0, g_e2 = 0.0;
    if (kSumRuleGuard && kSumRuleMoments >= 3 && (hi - lo <= 1 || hi - lo == N - 1)) {     // wave-uniform: same site, chain / corner neighbours
        g_slo = pick_site<N>(d0, lo);
        g_shi = pick_site<N>(d0, hi);
        if (hi == lo)
            g_e2 = ((lo == 0) ? corner * corner : pick_site<N>(e0sq, lo - 1)) + ((lo == N - 1) ? corner * corner : pick_site<N>(e0sq, lo));
    }
    if constexpr (!FOLD) {
        A.re[N - 1][0] = (float)corner;                           // (N >= 3: not a chain bond)
        A.im[N - 1][0] = 0.0f;
    }
    // the way round: B = c * conj(prod of the others); none for in == out (diagonal cofactor)
    const double br = (lo == hi) ? 0.0 : corner * rr, bi = (lo == hi) ? 0.0 : -corner * ri;
    const RingChi<N> chi{d0, e0sq, corner * corner, 2.0 * corner * fma(ar, rr, -ai * ri)};
    float df[N], ef[N], scale32;
    if constexpr (FOLD) {
        fhr[N - 1] = 0.0f;
        fhi[N - 1] = 0.0f;
        ring_fold_tridiag_f32<N>(fd, fhr, fhi, (float)corner, df, ef);
    } else {
        ring_tridiag_f32<N>(A, df, ef);
    }
    const bool ok32 = tridiag_ql_f32<N>(df, ef, scale32);
    double lam[N];
    bool ok = mixed_refine<N>(chi, df, scale32, ok32, lam, extra_steps);
    // (a lane that is not settled keeps computing - on garbage - and reports false: the decision is per SAMPLE)
    // 1 / chi'(lam_k) = 1 / prod_{m != k}(lam_k - lam_m); a sample with a pair closer than kRingGapTol of the scale is
    // handed to the eigenvector route as well: the numerators below are recurrences evaluated beside their own roots AND the
    // two path terms interfere destructively next to a degeneracy (translation-invariant ring: |dF| ~ 3e-14 / gap)
    double w[N];
    ok = ends_weights<N, true>(1.0, lam, w, kRingGapTol) && ok;
    // P_wrap: open chain hi+1 .. N-1, 0 .. lo-1 (through the corner); P_between: sites lo+1 .. hi-1
    double pa[N], pam[N], pb[N], pbm[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        pa[k] = 1.0; pam[k] = 0.0;
        pb[k] = 1.0; pbm[k] = 0.0;
    }
#pragma unroll
    for (int m = 1; m < N; ++m) {
        if (m > hi) {                                             // wave-uniform
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double t = fma(lam[k] - d0[m], pa[k], -e0sq[m - 1] * pam[k]);
                pam[k] = pa[k];
                pa[k] = t;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < N - 1; ++m) {
        if (m < lo) {
            const double cpl = (m == 0) ? corner * corner : e0sq[m - 1];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double t = fma(lam[k] - d0[m], pa[k], -cpl * pam[k]);
                pam[k] = pa[k];
                pa[k] = t;
            }
        }
    }
#pragma unroll
    for (int m = 1; m < N - 1; ++m) {
        if (m > lo && m < hi) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double t = fma(lam[k] - d0[m], pb[k], -e0sq[m - 1] * pbm[k]);
                pbm[k] = pb[k];
                pb[k] = t;
            }
        }
    }
    // amplitude for (row hi, column lo); the conjugate when out < in
    const double sgn = (out < in) ? -1.0 : 1.0;
    const double T = fabs(x[N]);
    const double Tk = T * kTurnsPerRadian;
    double re = 0.0, im = 0.0;
    // a-posteriori guard (tridiag_core.h: kSumRuleGuard): complex moments of the weights about lam_0, for (row hi, column lo)
    double s0r = 0.0, s0i = 0.0, s1r = 0.0, s1i = 0.0, s2r = 0.0, s2i = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double wr = w[k] * fma(ar, pa[k], br * pb[k]);
        const double wq = w[k] * fma(ai, pa[k], bi * pb[k]);
        const double wi = sgn * wq;
        if (kSumRuleGuard) {
            s0r += wr;
            s0i += wq;
        }
        if (k == 0) {
            re = wr;
            im = wi;
        } else {
            const double dl = lam[k] - lam[0];
            double sk, ck;
            if (kTableSinCos) sincos_table(Tk * dl, sctab, sk, ck);
            else sincos_reduced(T * dl, sk, ck);
            re = fma(wr, ck, fma(wi, sk, re));                    // (wr + i wi)(c - i s)
            im = fma(wi, ck, fma(-wr, sk, im));
            if (kSumRuleGuard && kSumRuleMoments >= 3) {
                const double tr = wr * dl, ti = wq * dl;
                s1r += tr;
                s1i += ti;
                s2r = fma(tr, dl, s2r);
                s2i = fma(ti, dl, s2i);
            }
        }
    }
    fid = fma(re, re, im * im);
    if (kSumRuleGuard) {
        // right-hand sides ((H - c)^m)[hi, lo] of the periodic tridiagonal matrix: D hops along the chain (product A of the
        // couplings), N - D the way round through the corner (product B); see the block comment at kSumRuleGuard
        const int D = hi - lo, ND = N - D;                        // wave-uniform
        const double c = lam[0];
        double m0 = 0.0, m1r = 0.0, m1i = 0.0, m2r = 0.0, m2i = 0.0;
        if (D == 0) {
            const double xlo = g_slo - c;
            m0 = 1.0;
            m1r = xlo;
            m2r = fma(xlo, xlo, g_e2);
        } else {
            double nr = 0.0, ni = 0.0;                            // the one-hop amplitude(s)
            if (D == 1) { nr += ar; ni += ai; }
            if (ND == 1) { nr += br; ni += bi; }
            m1r = nr;
            m1i = ni;
            if (D == 1 || ND == 1) {
                const double sd = (g_slo - c) + (g_shi - c);
                m2r = nr * sd;
                m2i = ni * sd;
            }
            if (D == 2) { m2r += ar; m2i += ai; }
            if (ND == 2) { m2r += br; m2i += bi; }
        }
        const double sc = (double)scale32;
        ok = ok && moments_ok(s0r - m0, s1r - m1r, s2r - m2r, sc) && moments_ok(s0i, s1i - m1i, s2i - m2i, sc);
    }
    return ok && (fid <= 2.0);
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
    TriEig<N, 4> s;
    if constexpr (N > kRingDenseMaxN) {
        double dd[N], hr[N], hi[N];
#pragma unroll
        for (int i = 0; i < N; ++i) dd[i] = x[i] + h0d[i] + loadg(3 * i);
#pragma unroll
        for (int i = 1; i < N; ++i) {
            hr[i - 1] = h0o[i - 1] + loadg(3 * i + 1);
            hi[i - 1] = loadg(3 * i + 2);
        }
        hr[N - 1] = 0.0;
        hi[N - 1] = 0.0;
        ring_fold_tridiag_rows<N>(dd, hr, hi, corner, in, out, s);
    } else {
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
        hermitian_tridiag_rows<N>(A, in, out, s);
    }
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
