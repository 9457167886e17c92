// Per-sample arithmetic for a NON-HERMITIAN chain Hamiltonian with Hermitian couplings and a COMPLEX diagonal - what
// `directional_perturbation` produces for its diagonal "directions" (noise_model.py:196-199: z[p,p] = a + ib is
// overwritten by a - ib, so H[p,p] gets an imaginary part) and what rc_mc_fidelity_nh_f64_async accepts in general
// (H = HH + Z(draws) + diag(x) + i diag(diag_imag)).
//
// The diagonal gauge of tridiag_core.h still applies - the couplings come in Hermitian pairs (h, conj h), a diagonal
// unitary similarity makes them |h| and leaves the (complex) diagonal alone, and |U[out,in]| is invariant.  What is left
// is a COMPLEX SYMMETRIC tridiagonal matrix T = T^T (real couplings, complex diagonal).  The implicit QL iteration works on
// it verbatim in complex arithmetic with complex-ORTHOGONAL plane rotations (c^2 + s^2 = 1, no conjugation; Cullum &
// Willoughby 1996): T = Q diag(lam) Q^T, Q^T Q = I, hence
//     U[out,in] = sum_k Q[out,k] Q[in,k] exp(-i T lam_k),      lam_k complex,
// with the two rows of Q accumulated through the sweeps as in the rows mode of the Hermitian kernel.  Such rotations are
// not norm-preserving: a rotation whose f^2 + g^2 (nearly) vanishes breaks down.  With an imaginary part of the size of
// the noise on a spectrum of the size of the biases that does not happen; a sample where it does (non-finite result,
// sweep cap) reports false and is recomputed by the dense Pade-expm kernel - the reference's own algorithm shape.
// One sample per lane, everything unrolled in registers (12 N doubles of state), wave-uniform control flow.
#pragma once
#include "tridiag_core.h"

namespace rc {

template <int N>
struct CTriEig {
    double dr[N], di[N];            // diagonal -> eigenvalues
    double er[N], ei[N];            // e[i] couples sites i and i+1 (real on entry); e[N-1] is padding
    double zr[2][N], zi[2][N];      // rows `in`, `out` of the accumulated complex-orthogonal Q
};

// principal square root of x + i y and its reciprocal
RC_HD void csqrt_rsqrt(double x, double y, double& sr, double& si, double& ir, double& ii) {
    double m, minv;
    sqrt_rsqrt(fma(x, x, fma(y, y, 1e-300)), m, minv);                 // |z|
    // t = sqrt((|z| + |x|) / 2) is well conditioned; the other component is y / (2 t)
    double t, tinv;
    sqrt_rsqrt(0.5 * (m + fabs(x)) + 1e-300, t, tinv);
    const double u = 0.5 * y * tinv;                                    // |u| = sqrt((|z| - |x|) / 2)
    sr = (x >= 0.0) ? t : fabs(u);
    si = (x >= 0.0) ? u : copysign(t, y);
    // 1 / sqrt(z) = conj(sqrt(z)) / |z|
    ir = sr * minv;
    ii = -si * minv;
}

constexpr int kCsymSweepCap = 14;
constexpr double kCsymOrthTol = 1e-11;        // |sum_k Q[out,k] Q[in,k] - delta| above this: recompute by expm
constexpr double kCsymGrowthMax = 1e3;        // sum_k |w_k| above this (eps * growth ~ 1e-13 of rounding in the phase sum)

// Implicit QL with Wilkinson shift on the complex symmetric tridiagonal matrix in s, two rows of Q.  Same wave-uniform
// structure as tridiag_ql2_fast.  Returns false - per lane - on the sweep cap; a breakdown shows up as a non-finite result.
template <int N>
RC_HD bool csym_ql_rows(CTriEig<N>& s) {
    lanemask_t badm = 0ull;
    const lanemask_t full = lane_ballot(true);
#pragma unroll
    for (int l = 0; l < N - 1; ++l) {
        auto small = [&]() {
            return fabs(s.er[l]) + fabs(s.ei[l]) <= kEps * (fabs(s.dr[l]) + fabs(s.di[l]) + fabs(s.dr[l + 1]) + fabs(s.di[l + 1]));
        };
        lanemask_t donem = lane_ballot(small());
        if ((donem | badm) == full) continue;
        int iter = 0;
#pragma unroll 1
        do {
            // Wilkinson shift: mu = d_l - e_l^2 / (delta + sgn rho), delta = (d_{l+1} - d_l)/2, rho = sqrt(delta^2 + e_l^2),
            // the sign that makes |delta + sgn rho| the larger one
            const double elr = s.er[l], eli = s.ei[l];
            const double der = 0.5 * (s.dr[l + 1] - s.dr[l]), dei = 0.5 * (s.di[l + 1] - s.di[l]);
            const double e2r = fma(elr, elr, -eli * eli), e2i = 2.0 * elr * eli;
            double rr, ri, qr, qi;
            csqrt_rsqrt(fma(der, der, -dei * dei) + e2r, 2.0 * der * dei + e2i, rr, ri, qr, qi);
            const double sg = (fma(der, rr, dei * ri) >= 0.0) ? 1.0 : -1.0;
            const double tr = fma(sg, rr, der), ti = fma(sg, ri, dei);                 // delta + sgn rho
            const double tn = rcp_full(fma(tr, tr, fma(ti, ti, 1e-300)));
            const double ir = tr * tn, ii = -ti * tn;                                    // 1 / (delta + sgn rho)
            double gr = s.dr[N - 1] - s.dr[l] + fma(e2r, ir, -e2i * ii);
            double gi = s.di[N - 1] - s.di[l] + fma(e2r, ii, e2i * ir);
            double snr = 1.0, sni = 0.0, csr = 1.0, csi = 0.0, pr = 0.0, pi = 0.0;
#pragma unroll
            for (int i = N - 2; i >= l; --i) {
                const double fr = fma(snr, s.er[i], -sni * s.ei[i]), fi = fma(snr, s.ei[i], sni * s.er[i]);
                const double br = fma(csr, s.er[i], -csi * s.ei[i]), bi = fma(csr, s.ei[i], csi * s.er[i]);
                // r = sqrt(f^2 + g^2) (complex, no conjugation), s = f / r, c = g / r
                const double h2r = fma(fr, fr, -fi * fi) + fma(gr, gr, -gi * gi);
                const double h2i = 2.0 * (fr * fi + gr * gi);
                double hr, hi, vr, vi;
                csqrt_rsqrt(h2r, h2i, hr, hi, vr, vi);
                if (i + 1 <= N - 2) {
                    s.er[i + 1] = hr;
                    s.ei[i + 1] = hi;
                }
                snr = fma(fr, vr, -fi * vi);
                sni = fma(fr, vi, fi * vr);
                csr = fma(gr, vr, -gi * vi);
                csi = fma(gr, vi, gi * vr);
                gr = s.dr[i + 1] - pr;
                gi = s.di[i + 1] - pi;
                // r = (d_i - g) s + 2 c b
                const double ur = s.dr[i] - gr, ui = s.di[i] - gi;
                const double cbr = fma(csr, br, -csi * bi), cbi = fma(csr, bi, csi * br);
                const double rr2 = fma(ur, snr, fma(-ui, sni, 2.0 * cbr));
                const double ri2 = fma(ur, sni, fma(ui, snr, 2.0 * cbi));
                pr = fma(snr, rr2, -sni * ri2);
                pi = fma(snr, ri2, sni * rr2);
                s.dr[i + 1] = gr + pr;
                s.di[i + 1] = gi + pi;
                gr = fma(csr, rr2, -csi * ri2) - br;
                gi = fma(csr, ri2, csi * rr2) - bi;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const double f1r = s.zr[q][i + 1], f1i = s.zi[q][i + 1];
                    const double z0r = s.zr[q][i], z0i = s.zi[q][i];
                    s.zr[q][i + 1] = fma(snr, z0r, fma(-sni, z0i, fma(csr, f1r, -csi * f1i)));
                    s.zi[q][i + 1] = fma(snr, z0i, fma(sni, z0r, fma(csr, f1i, csi * f1r)));
                    s.zr[q][i] = fma(csr, z0r, fma(-csi, z0i, fma(-snr, f1r, sni * f1i)));
                    s.zi[q][i] = fma(csr, z0i, fma(csi, z0r, fma(-snr, f1i, -sni * f1r)));
                }
            }
            s.dr[l] -= pr;
            s.di[l] -= pi;
            s.er[l] = gr;
            s.ei[l] = gi;
            ++iter;
            donem = lane_ballot(small());
            if (iter >= kCsymSweepCap) badm |= full & ~donem;
        } while ((donem | badm) != full);
    }
    return !lane_bit(badm);
}

// exp(x) for |x| <~ 700: 2^k * e^r, r = x - k ln2 in two steps, degree-11 Taylor on |r| <= ln2 / 2 (truncation 2e-17 relative)
RC_HD double exp_small(double x) {
    const double kf = rint(x * 1.44269504088896340736e+00);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.0 / 39916800.0;
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}

// Fidelity of one sample with a complex diagonal.  loadg(j): the sample's j-th draw (g0_i, g1_i, g2_i); loadi(i): the
// imaginary part added to H[i][i].  Returns false - per sample - when the sample must be recomputed by the expm route.
template <int N, typename LoadG, typename LoadI>
RC_HD bool csym_fidelity(const double* x, const double* h0d, const double* h0o, LoadG loadg, LoadI loadi, int in, int out,
                         const double* sctab, double& fid) {
    CTriEig<N> s;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        s.dr[i] = x[i] + h0d[i] + loadg(3 * i);
        s.di[i] = loadi(i);
        s.zr[0][i] = (i == in) ? 1.0 : 0.0;
        s.zr[1][i] = (i == out) ? 1.0 : 0.0;
        s.zi[0][i] = 0.0;
        s.zi[1][i] = 0.0;
    }
#pragma unroll
    for (int i = 1; i < N; ++i) {
        const double re = h0o[i - 1] + loadg(3 * i + 1);
        const double im = loadg(3 * i + 2);
        // an exactly cancelled coupling enters as 1e-60, not as the 1e-150 of the real kernels: below a cut the bulge chase
        // works on f, g of the size of that coupling, and a complex modulus squares them TWICE (|f^2 + g^2|^2: 1e-600
        // underflows, the rotation comes out as garbage - found by the near-breakdown test of round 4); what the nudge costs
        // is e^2 / gap = 1e-120 in an eigenvalue
        double r, rinv;
        sqrt_rsqrt(fma(re, re, fma(im, im, 1e-120)), r, rinv);
        s.er[i - 1] = r;
        s.ei[i - 1] = 0.0;
    }
    s.er[N - 1] = 0.0;
    s.ei[N - 1] = 0.0;
    bool ok = csym_ql_rows<N>(s);
    const double T = fabs(x[N]);
    const double Tk = T * kTurnsPerRadian;
    double re = 0.0, im = 0.0;
    // CONDITIONING GUARD (round 4).  Complex-orthogonal rotations are not norm-preserving: next to a breakdown (f^2 + g^2 ~ 0,
    // a near-defective H) the rows of Q grow and the result is finite but inaccurate - nothing the sweep cap or a
    // non-finite test would notice.  Cheap invariants of the accumulated rows tell: Q Q^T = I restricted to the two rows
    // carried - sum_k Q[out,k] Q[in,k] = delta(in, out), sum_k Q[in,k]^2 = sum_k Q[out,k]^2 = 1 (complex squares, no
    // conjugate; their violation IS the accumulated rotation error, and rows that underflowed to nothing fail the last two)
    // - and the growth sum_k |w_k| (1 for a unitary Q; the rounding error of the phase sum is eps times it).  A sample that
    // fails any of them is marked and recomputed by the Pade-expm pass - the reference's own algorithm shape, which has no
    // such failure mode.
    double s0r = 0.0, s0i = 0.0, grow = 0.0, n0r = 0.0, n0i = 0.0, n1r = 0.0, n1i = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        // w_k = Q[out,k] Q[in,k] (no conjugate);  exp(-i T (lr + i li)) = exp(T li) (cos(T lr) - i sin(T lr))
        const double wr = fma(s.zr[1][k], s.zr[0][k], -s.zi[1][k] * s.zi[0][k]);
        const double wi = fma(s.zr[1][k], s.zi[0][k], s.zi[1][k] * s.zr[0][k]);
        s0r += wr;
        s0i += wi;
        grow += fabs(wr) + fabs(wi);
        n0r += fma(s.zr[0][k], s.zr[0][k], -s.zi[0][k] * s.zi[0][k]);
        n0i = fma(s.zr[0][k], s.zi[0][k], n0i);                  // (half the imaginary part: compared with 0)
        n1r += fma(s.zr[1][k], s.zr[1][k], -s.zi[1][k] * s.zi[1][k]);
        n1i = fma(s.zr[1][k], s.zi[1][k], n1i);
        double sk, ck;
        if (kTableSinCos) sincos_table(Tk * s.dr[k], sctab, sk, ck);
        else sincos_reduced(T * s.dr[k], sk, ck);
        const double amp = exp_small(T * s.di[k]);
        const double cr = amp * ck, ci = -amp * sk;
        re = fma(wr, cr, fma(-wi, ci, re));
        im = fma(wr, ci, fma(wi, cr, im));
    }
    fid = fma(re, re, im * im);
    ok = ok && (fid == fid) && (fid < 1e300);
    ok = ok && (fabs(s0r - ((in == out) ? 1.0 : 0.0)) <= kCsymOrthTol) && (fabs(s0i) <= kCsymOrthTol) && (grow <= kCsymGrowthMax);
    ok = ok && (fabs(n0r - 1.0) <= kCsymOrthTol) && (fabs(n0i) <= kCsymOrthTol) && (fabs(n1r - 1.0) <= kCsymOrthTol) &&
         (fabs(n1i) <= kCsymOrthTol);
    return ok;
}

}  // namespace rc
