// Per-sample arithmetic of the chain-topology fidelity kernel (one sample per lane).
//
// What it computes (reference: noise_model.py:98-109 with :122-147): for the Hermitian tridiagonal
//   H = diag(d) + offdiag(h0_off_i + g1_i + i g2_i),   d_i = x_i + h0_diag_i + g0_i,
// the transfer amplitude phi = [exp(-i T H)]_{out,in} and the fidelity |phi|^2.
//
// How (not how the reference does it - the reference calls scipy.linalg.expm on the dense matrix):
//   * a diagonal unitary gauge makes H real symmetric tridiagonal with couplings e_i = |h0_off_i+g1_i+i g2_i|;
//     |phi| is invariant under that gauge, so only (d, e) are needed;
//   * implicit-shift QL iteration (Wilkinson shift) on (d, e) held in registers, accumulating only the two
//     rows `in` and `out` of the eigenvector matrix;
//   * phi = sum_k Q[out,k] Q[in,k] exp(-i T lambda_k).
// Every loop over matrix indices is fully unrolled (N is a template parameter) so that d/e/z stay in VGPRs;
// the active QL window [l, m] is handled by predication, never by runtime indexing.
//
// The kernel is bound by fp64 VALU issue, so the arithmetic is written to minimise instruction count:
//   * one v_rsq_f64 + a Goldschmidt step pair yields BOTH sqrt(h) and 1/sqrt(h) of a rotation (no division,
//     no IEEE sqrt expansion with its range scaling - operands here are O(1e-300 .. 1e8));
//   * the Wilkinson shift needs only a low-accuracy sqrt and reciprocal (a shift changes the convergence
//     speed, never the result: every step is an exact orthogonal similarity whatever the shift);
//   * sin/cos use a two-constant Cody-Waite reduction (|T lambda| < 1e5 here) and the fdlibm kernels.
//
// The header is plain C++ so that the exact same algorithm can be compiled for the host by the CPU unit
// tests (tests/test_host_core.py builds it with g++); the product only ever runs it inside the HIP kernels.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define RC_HD __host__ __device__ __forceinline__
#else
#define RC_HD inline
#endif

namespace rc {

constexpr double kEps = 2.220446049250313e-16;   // DBL_EPSILON: split tolerance of the QL iteration
constexpr int kMaxSweepsPerEig = 40;             // hard cap on QL iterations per eigenvalue (never reached)

// ---- hardware seeds ---------------------------------------------------------------------------------------
RC_HD double seed_rsq(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);              // v_rsq_f64: ~2^-26 relative accuracy
#else
    return 1.0 / sqrt(x);
#endif
}
RC_HD double seed_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);              // v_rcp_f64
#else
    return 1.0 / x;
#endif
}

// sqrt(x) and 1/sqrt(x) for x > 0: v_rsq_f64 seed (measured 5e-8 relative) + two coupled Goldschmidt steps
// (5e-8 -> 4e-15 -> rounding level).  9 VALU ops, no division, no range scaling.
RC_HD void sqrt_rsqrt(double x, double& root, double& inv) {
    const double y = seed_rsq(x);
    double g = x * y;                 // ~ sqrt(x)
    double h = 0.5 * y;               // ~ 1 / (2 sqrt(x))
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    root = g;
    inv = h + h;
}

// cheap sqrt (~2^-50): good enough for the shift
RC_HD double sqrt_fast(double x) {
    const double y = seed_rsq(x);
    const double g = x * y;
    const double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    return fma(g, r, g);
}

// cheap reciprocal (~2^-50)
RC_HD double rcp_fast(double x) {
    const double y = seed_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

// sin and cos for |x| < ~1e5 (here |x| = T |lambda| < 1e3): n = rint(x 2/pi), r = x - n pi/2 in two fma steps
// (pi/2 split hi + lo, error n * 1e-33), then the classic degree-13 / degree-14 minimax kernels on |r| <= pi/4.
RC_HD void sincos_reduced(double x, double& s, double& c) {
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)n;
    const double sa = (q & 1) ? cr : sr;
    const double ca = (q & 1) ? sr : cr;
    s = (q & 2) ? -sa : sa;
    c = ((q + 1) & 2) ? -ca : ca;
}

template <int N>
struct TriEig {
    double d[N];    // diagonal -> eigenvalues
    double e[N];    // e[i] couples sites i and i+1; e[N-1] is padding (0)
    double zi[N];   // row `in`  of the accumulated eigenvector matrix
    double zo[N];   // row `out`
};

// Implicit QL with Wilkinson shift on a real symmetric tridiagonal matrix, two eigenvector rows.
template <int N>
RC_HD void tridiag_ql2(TriEig<N>& s) {
#pragma unroll
    for (int l = 0; l < N - 1; ++l) {
        for (int iter = 0; iter < kMaxSweepsPerEig; ++iter) {
            // smallest m >= l with negligible e[m]  (m = N-1 if none)
            int m = N - 1;
#pragma unroll
            for (int mm = N - 2; mm >= l; --mm) {
                const double dd = fabs(s.d[mm]) + fabs(s.d[mm + 1]);
                if (fabs(s.e[mm]) <= kEps * dd) m = mm;
            }
            if (m == l) break;
            // d[m] without runtime indexing
            double dm = s.d[N - 1];
#pragma unroll
            for (int mm = N - 2; mm > l; --mm) dm = (m == mm) ? s.d[mm] : dm;

            // Wilkinson shift from the leading 2x2 of the window: mu = d_l - e_l^2 / (delta + sign(delta) rho),
            // delta = (d_{l+1} - d_l)/2, rho = sqrt(delta^2 + e_l^2);  g = d_m - mu
            const double el = s.e[l];
            const double delta = 0.5 * (s.d[l + 1] - s.d[l]);
            const double e2 = el * el;
            const double rho = sqrt_fast(fma(delta, delta, e2));
            double g = dm - s.d[l] + e2 * rcp_fast(delta + copysign(rho, delta));
            double sn = 1.0, cs = 1.0, p = 0.0;
#pragma unroll
            for (int i = N - 2; i >= l; --i) {
                if (i < m) {
                    double f = sn * s.e[i];
                    const double b = cs * s.e[i];
                    // Rotation annihilating the bulge: r = hypot(f, g), s = f/r, c = g/r.  g is nudged by
                    // 1e-150 (a no-op unless |g| < 1e-134) so that f = g = 0 gives the identity rotation
                    // without any compare/select; the neglected bulge is then < 1e-150.
                    const double gn = g + 1e-150;
                    double r, rinv;
                    sqrt_rsqrt(fma(f, f, gn * gn), r, rinv);
                    s.e[i + 1] = (i + 1 == m) ? 0.0 : r;
                    sn = f * rinv;
                    cs = gn * rinv;
                    g = s.d[i + 1] - p;
                    r = fma(s.d[i] - g, sn, 2.0 * cs * b);
                    p = sn * r;
                    s.d[i + 1] = g + p;
                    g = fma(cs, r, -b);
                    f = s.zi[i + 1];
                    s.zi[i + 1] = fma(sn, s.zi[i], cs * f);
                    s.zi[i] = fma(cs, s.zi[i], -sn * f);
                    f = s.zo[i + 1];
                    s.zo[i + 1] = fma(sn, s.zo[i], cs * f);
                    s.zo[i] = fma(cs, s.zo[i], -sn * f);
                }
            }
            s.d[l] -= p;
            s.e[l] = g;
        }
    }
}

// Fidelity of one sample.  loadg(j) returns this sample's j-th draw, laid out (g0_i, g1_i, g2_i), i = 0..N-1.
// x: controller (N biases, then T).
template <int N, typename LoadG>
RC_HD double chain_fidelity(const double* x, const double* h0d, const double* h0o, LoadG loadg,
                            int in, int out) {
    TriEig<N> s;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        s.d[i] = x[i] + h0d[i] + loadg(3 * i);
        s.zi[i] = (i == in) ? 1.0 : 0.0;
        s.zo[i] = (i == out) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int i = 1; i < N; ++i) {
        const double re = h0o[i - 1] + loadg(3 * i + 1);
        const double im = loadg(3 * i + 2);
        const double h = fma(re, re, im * im);
        double r, rinv;
        sqrt_rsqrt(h, r, rinv);
        s.e[i - 1] = (h > 0.0) ? r : 0.0;
    }
    s.e[N - 1] = 0.0;
    tridiag_ql2<N>(s);
    const double T = fabs(x[N]);
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double sk, ck;
        sincos_reduced(T * s.d[k], sk, ck);
        const double w = s.zo[k] * s.zi[k];
        re = fma(w, ck, re);
        im = fma(-w, sk, im);
    }
    return fma(re, re, im * im);
}

}  // namespace rc
