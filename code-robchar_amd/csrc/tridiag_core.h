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
// The header is plain C++ so that the exact same arithmetic can be compiled for the host by the CPU unit
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

// sqrt / reciprocal used by the rotations.  Plain IEEE ops; kept in one place so that the device build can
// swap in a refined v_rsq_f64 sequence without touching the algorithm.
RC_HD double rc_sqrt(double x) { return sqrt(x); }

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

            double g = (s.d[l + 1] - s.d[l]) / (2.0 * s.e[l]);
            double r = rc_sqrt(g * g + 1.0);
            g = dm - s.d[l] + s.e[l] / (g + copysign(r, g));
            double sn = 1.0, cs = 1.0, p = 0.0;
#pragma unroll
            for (int i = N - 2; i >= l; --i) {
                if (i < m) {
                    double f = sn * s.e[i];
                    const double b = cs * s.e[i];
                    r = rc_sqrt(f * f + g * g);
                    s.e[i + 1] = (i + 1 == m) ? 0.0 : r;
                    const double rinv = (r > 0.0) ? 1.0 / r : 0.0;
                    sn = f * rinv;
                    cs = (r > 0.0) ? g * rinv : 1.0;
                    g = s.d[i + 1] - p;
                    r = (s.d[i] - g) * sn + 2.0 * cs * b;
                    p = sn * r;
                    s.d[i + 1] = g + p;
                    g = cs * r - b;
                    f = s.zi[i + 1];
                    s.zi[i + 1] = sn * s.zi[i] + cs * f;
                    s.zi[i] = cs * s.zi[i] - sn * f;
                    f = s.zo[i + 1];
                    s.zo[i + 1] = sn * s.zo[i] + cs * f;
                    s.zo[i] = cs * s.zo[i] - sn * f;
                }
            }
            s.d[l] -= p;
            s.e[l] = g;
        }
    }
}

// Fidelity of one sample.  g points at this sample's 3N draws laid out (g0_i, g1_i, g2_i), i = 0..N-1,
// with element stride `gs` (1 for a private copy).  x: controller (N biases, then T).
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
        s.e[i - 1] = rc_sqrt(re * re + im * im);
    }
    s.e[N - 1] = 0.0;
    tridiag_ql2<N>(s);
    const double T = fabs(x[N]);
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double sk, ck;
        sincos(T * s.d[k], &sk, &ck);
        const double w = s.zo[k] * s.zi[k];
        re += w * ck;
        im -= w * sk;
    }
    return re * re + im * im;
}

}  // namespace rc
