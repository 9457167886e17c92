// Per-sample arithmetic of the chain-topology fidelity kernel (one sample per lane).
//
// What it computes (reference: noise_model.py:98-109 with :122-147): for the Hermitian tridiagonal
//   H = diag(d) + offdiag(h0_off_i + g1_i + i g2_i),   d_i = x_i + h0_diag_i + g0_i,
// the transfer amplitude phi = [exp(-i T H)]_{out,in} and the fidelity |phi|^2.
//
// How (not how the reference does it - the reference calls scipy.linalg.expm on the dense matrix):
//   * a diagonal unitary gauge makes H real symmetric tridiagonal with couplings e_i = |h0_off_i+g1_i+i g2_i|;
//     |phi| is invariant under that gauge, so only (d, e) are needed;
//   * implicit-shift QL iteration (Wilkinson shift) on (d, e) held in registers;
//   * weights w_k = Q[out,k] Q[in,k] either from the two rows `in` / `out` of the eigenvector matrix accumulated
//     through the sweeps (kWeightsRows) or - default - from the eigenvalues alone through the adjugate of
//     (lambda I - H) (kWeightsAdjugate / kWeightsEnds), which also allows a 1e-10 split tolerance;
//   * phi = sum_k w_k exp(-i T lambda_k).
// Every loop over matrix indices is fully unrolled (N is a template parameter) so that the state stays in VGPRs;
// the QL control flow is wave-uniform and the common sweep has no predication at all (see tridiag_ql2_fast).
//
// The kernel runs the socket at its power cap on fp64 VALU work, so the arithmetic is written to minimise the
// instruction count, transcendental seeds (3.4 FMAs each) first:
//   * one v_rsq_f64 + one third-order correction yields BOTH sqrt(h) and 1/sqrt(h) of a rotation (7 ops, no
//     division, no IEEE sqrt expansion with its range scaling - operands here are O(1e-300 .. 1e8));
//   * the Wilkinson shift uses one raw hardware seed and no reciprocal (a shift changes the convergence speed,
//     never the result: every step is an exact orthogonal similarity whatever the shift);
//   * the N weight denominators share one reciprocal; the last 2x2 block is solved in closed form;
//   * sin/cos use a two-constant Cody-Waite reduction (|T lambda| < 1e5 here) and the fdlibm kernels.
//
// The header is plain C++ so that the exact same algorithm can be compiled for the host by the CPU unit
// tests (tests/test_host_core.py builds it with g++); the product only ever runs it inside the HIP kernels.
#pragma once
#include <math.h>
#if defined(RC_GUARD_DEBUG)
#include <stdio.h>
#endif

#if defined(__HIPCC__)
#define RC_HD __host__ __device__ __forceinline__
#else
#define RC_HD inline
#endif

// Branch-weight hints for the rare paths (stepping path, tile-wide fp64 QL, repair): with -DRC_EXPECT_HINTS=1 the block
// placement moves the cold code behind the kernel's hot path instead of between its stages (the hot path then falls through
// where it otherwise takes three branches over ~5 KB of cold instructions).  Measured (round 4, same-box A/B, two rounds of
// 300 launches, profiles/r04_ab_expect_hints.txt): N = 7 -0.7 %, N = 10 XXZ +1.4 %, shipped controllers -0.5 %, ring +0.5 % -
// noise; the ~50 VALU instructions per wave by which a build WITHOUT the stepping path is shorter (PMC: 1 327 against 1 377)
// are the acceptance test itself (the step / critical-point maxima and the bound), not the code's presence.  Off.
#ifndef RC_EXPECT_HINTS
#define RC_EXPECT_HINTS 0
#endif
#if RC_EXPECT_HINTS
#define RC_LIKELY(x) __builtin_expect(!!(x), 1)
#define RC_UNLIKELY(x) __builtin_expect(!!(x), 0)
#else
#define RC_LIKELY(x) (x)
#define RC_UNLIKELY(x) (x)
#endif

#if defined(RC_FLAG_STATS) && !defined(__HIP_DEVICE_COMPILE__)
void rc_flag_stats_hook(int n, unsigned roots, double maxd, const double* lam);
#endif
#if defined(RC_HOST_WAVE) && !defined(__HIP_DEVICE_COMPILE__)
namespace rc_host_wave {            // tests/host/host_wave.cpp: lock-step emulation of a wave by host threads
unsigned long long ballot(bool v);  // mask of the ACTIVE lanes whose predicate holds (every active lane calls it: a barrier)
int lane();                         // this thread's lane index
}
#endif

namespace rc {

constexpr double kEps = 2.220446049250313e-16;   // DBL_EPSILON: split tolerance of the QL iteration
// Split tolerance of the fast path when only EIGENVALUES are needed (adjugate weight modes): dropping a coupling
// e_l <= tol * (|d_l| + |d_l+1|) moves the eigenvalues by O(e_l^2 / gap) only, so 1e-10 costs < 1e-13 in fidelity
// (host study over 6.4e6 random / near-degenerate samples, N = 4..10, T <= 70: no change of the max error up to
// tol = 1e-10, 2e-12 at 1e-9, 2e-10 at 1e-8 - quadratic as expected) and saves the last, already-converged sweep of
// most eigenvalues: 12 % fewer rotations per 64-sample tile.  Eigenvector rows are first order in e_l, so the rows
// mode keeps DBL_EPSILON.
#ifndef RC_FAST_EPS
#define RC_FAST_EPS 1e-10
#endif
constexpr double kFastEpsValues = RC_FAST_EPS;
constexpr int kMaxSweepsPerEig = 40;             // hard cap on QL iterations per eigenvalue (general path)
#ifndef RC_BATCH_INVERSE
#define RC_BATCH_INVERSE 1
#endif
constexpr bool kBatchInverse = RC_BATCH_INVERSE;   // adjugate weights: one reciprocal per batch of weights (ends_weights)
#ifndef RC_TABLE_SINCOS
#define RC_TABLE_SINCOS 1
#endif
constexpr bool kTableSinCos = RC_TABLE_SINCOS;     // fast path: table-driven sin/cos (sincos_table)
#ifndef RC_SHIFT_NODIV
#define RC_SHIFT_NODIV 1
#endif
#ifndef RC_CLOSED_2X2
#define RC_CLOSED_2X2 1
#endif
constexpr bool kClosedForm2x2 = RC_CLOSED_2X2;   // fast path: solve the last 2x2 block directly instead of sweeping
// Mixed path, after the all-fp64 QL: pairs closer than this * scale need eigenvectors (repair path).  END-TO-END weights
// (numerator = the constant prod e): the two weights of a pair are +-A / gap with the SAME computed gap and a smooth A, their
// joint contribution is a divided difference of a smooth function - accurate down to gaps at the eigenvalues' own rounding
// level (mpmath study scripts/proto/tiny_gap_weights.py; fuzz: 15 000 adversarial configurations).  GENERAL adjugate
// weights: the numerators phi_i(lam_k) psi_j(lam_k) are three-term recurrences evaluated next to their own roots (a level
// of the block behind a weak or cut bond sits right beside lam_k); their rounding noise nu ~ eps * scale * cond is NOT a
// smooth function of lam_k, and the pair's sum rule is violated by (nu_A - nu_B) / gap - the fuzz campaign of round 3
// found |dF| up to 2.7e-8 at gaps of 1e-12 .. 5e-9 of the scale (cut chains with mirror-symmetric halves), and still
// 4e-11 .. 9e-11 at 1.6e-7 .. 1.3e-6 (N = 13, |d| ~ 1: error ~ 1e-17 / relative gap).  So the general adjugate mode sends
// every pair closer than 4e-6 of the scale (the resolution of the mixed path's own distinct-roots check) to the
// eigenvector route - round 2 had 1e-7 there, which the same campaign shows to be marginal.
constexpr double kDegenerateGapEnds = 1e-12;
constexpr double kDegenerateGapNoMix = 1e-7;       // end-to-end weights behind the 1e-10-tolerance fp64 QL (N = 2, N >= 14): e_l^2 / gap must stay negligible
constexpr double kDegenerateGapAdjugate = 4e-6;
#ifndef RC_KEEP_SETTLED
#define RC_KEEP_SETTLED 1
#endif
// One Halley step on the all-fp64 QL's eigenvalues at N >= 14 (general adjugate mode; see chain_fidelity_fast).  Measured
// (round 4, profiles/r04_ab_polish_large_n.txt): the fuzz block's N = 16 worst case 1.03e-11 -> 6.4e-12 (lock-step host
// emulation of that configuration: 9.2e-12 -> 4.1e-13), kernel +16 % at N = 14 and N = 16 - while the campaign's overall worst
// (1.1e-11: adjugate numerators beside a pair 2e-5 of the scale apart at |T| = 63, N = 12, and the ring) is set elsewhere.  Off:
// 16 % of two non-benchmark sizes for no change of the bound the kernels can claim.
#ifndef RC_POLISH_LARGE_N
#define RC_POLISH_LARGE_N 0
#endif
constexpr bool kPolishLargeN = RC_POLISH_LARGE_N;
constexpr int kFastSweepCap = 10;                // fast path: more sweeps than this for one eigenvalue -> general path

// ---- hardware seeds ---------------------------------------------------------------------------------------
// v_rsq_f64 / v_rcp_f64 deliver ~5e-8 relative accuracy (measured, scripts/ubench/rsq_acc.hip).  The host build
// imitates that error (sign taken from a mantissa bit) so that the CPU unit tests exercise the refinement too.
RC_HD double seed_rsq(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);
#else
    union { double d; unsigned long long u; } b = {x};
    return (1.0 / sqrt(x)) * (1.0 + ((b.u & 8) ? 5e-8 : -5e-8));
#endif
}
RC_HD double seed_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    union { double d; unsigned long long u; } b = {x};
    return (1.0 / x) * (1.0 + ((b.u & 8) ? 5e-8 : -5e-8));
#endif
}

// sqrt(x) and 1/sqrt(x) for x > 0 to rounding level: seed y (5e-8), then ONE third-order (Householder) step
//   e = 1 - x y^2,  y <- y + y e (1/2 + 3/8 e)        (error ~ 5/16 e^3 ~ 3e-22 before rounding)
// and root = x y.  7 VALU ops in all, no division, no range scaling (operands here are O(1e-300 .. 1e8)).
RC_HD void sqrt_rsqrt(double x, double& root, double& inv) {
    const double y = seed_rsq(x);
    const double t = x * y;
    const double e = fma(-t, y, 1.0);
    const double p = fma(0.375, e, 0.5);
    const double q = y * e;
    inv = fma(q, p, y);
    root = x * inv;
}

// seed-accuracy sqrt and reciprocal (~5e-8): all the Wilkinson shift needs - its error scales with e_l^2, so the
// cubic convergence of the sweeps is untouched, and a shift never changes the result of an exact similarity
RC_HD double sqrt_fast(double x) { return x * seed_rsq(x); }
RC_HD double rcp_fast(double x) { return seed_rcp(x); }

// reciprocal to rounding level (weights of the adjugate modes): seed + two Newton steps
RC_HD double rcp_full(double x) {
    double y = seed_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    return fma(y, fma(-x, y, 1.0), y);
}

// ---- mixed-precision eigenvalues (fast path of the eigenvalue-only weight modes) ------------------------------
// The kernel is bound by the ENERGY of its fp64 VALU work (the socket sits at its power cap), and a 32-bit VALU
// instruction costs ~0.4 of a 64-bit one (scripts/ubench/energy_mix).  So the O(N^2) rotations of the QL iteration run
// in fp32 - they deliver every eigenvalue to ~1e-6 - and ONE third-order (Halley) step on the characteristic polynomial
// of the ORIGINAL fp64 matrix (three-term recurrences for p, p', p''/2: ~7N operations per eigenvalue, no rotation, no
// transcendental except one reciprocal) takes each of them to rounding level: error_new ~ error^3 / gap^2.  A tile in
// which some sample's step is too large for that bound (close pair, or a start the loose fp32 tolerance left a few 1e-6 off:
// ~9 % of the tiles of the N = 7 benchmark workload) keeps stepping.
#ifndef RC_MIXED_EIG
#define RC_MIXED_EIG 1
#endif
constexpr bool kMixedEig = RC_MIXED_EIG;
// Measured on MI355X (scripts/kbench.py, 1e6 evaluations, end-to-end mode, all-fp64 QL -> mixed): N = 3 / 4 / 5 / 6 / 7 / 8:
// -6 / -12 / -13 / -16 / -25 / -27 % kernel time; N = 9 / 10 / 11 / 12 / 13: -19 / -14 (XXZ: -18) / -11 / -8 / -2 (XXZ: -8) %
// (adjugate mode: -20 / -22 / -14 / -10 / -5 %).  Above, the fp64 (d, e^2) kept through the fp32 phase cost more registers
// than the kernels have without spilling.  (With the SLP vectoriser on - csrc/Makefile - the gain ended at N = 8.)
#ifndef RC_MIXED_MAX_N
#define RC_MIXED_MAX_N 13
#endif
// Split tolerance of the fp32 QL.  Looser than fp32 rounding on purpose: what a dropped e_l costs (e_l^2 / gap) is
// taken out again by the Halley step, and the step's own size is the acceptance test.  Measured (N = 7, VALU
// instructions per wave / kernel time): 2e-6: 1750 / 60.5 us, 1e-5: 1703, 3e-5: 1678 / 58.7, 1e-4: 1627 / 58.0,
// 3e-4: first tiles on the general path, 1e-3: 71 us.
#ifndef RC_F32_EPS
#define RC_F32_EPS 1e-4f
#endif
constexpr float kF32SplitTol = RC_F32_EPS;       // fp32 QL: e_l negligible below this * (|d_l| + |d_l+1|)
#ifndef RC_HALLEY_ACCEPT
#define RC_HALLEY_ACCEPT 1e-14
#endif
constexpr double kHalleyAccept = RC_HALLEY_ACCEPT;   // accept when max|step|^3 <= this * mingap^2 (error bound of the step)

RC_HD float seed_rsqf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);             // v_rsq_f32: 1 ulp
#else
    return 1.0f / sqrtf(x);
#endif
}
RC_HD float seed_rcpf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);               // v_rcp_f32: 1 ulp
#else
    return 1.0f / x;
#endif
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

// Table-driven sin/cos for the fast path: n = rint(x 32/pi), r = x - n pi/32 (|r| <= pi/64: degree-7 / degree-8
// Taylor kernels, truncation 5e-18 / 2e-20), (cos, sin)(n pi/32) from a 64-entry table covering the whole circle -
// no quadrant logic; the angle-addition formulas combine the two.  19 VALU operations + one 16-byte table read
// instead of 38.  `tab` = kSinCosTable values: 64 x (cos, sin)(2 pi k / 64), in LDS on the device.
#define RC_SINCOS_TABLE_VALUES \
    1.0, 0.0, 0.9951847266721969, 0.0980171403295606, \
    0.9807852804032304, 0.19509032201612828, 0.9569403357322088, 0.2902846772544624, \
    0.9238795325112867, 0.3826834323650898, 0.881921264348355, 0.47139673682599764, \
    0.8314696123025452, 0.5555702330196022, 0.773010453362737, 0.6343932841636455, \
    0.7071067811865476, 0.7071067811865476, 0.6343932841636455, 0.773010453362737, \
    0.5555702330196022, 0.8314696123025452, 0.47139673682599764, 0.881921264348355, \
    0.3826834323650898, 0.9238795325112867, 0.2902846772544624, 0.9569403357322088, \
    0.19509032201612828, 0.9807852804032304, 0.0980171403295606, 0.9951847266721969, \
    0.0, 1.0, -0.0980171403295606, 0.9951847266721969, \
    -0.19509032201612828, 0.9807852804032304, -0.2902846772544624, 0.9569403357322088, \
    -0.3826834323650898, 0.9238795325112867, -0.47139673682599764, 0.881921264348355, \
    -0.5555702330196022, 0.8314696123025452, -0.6343932841636455, 0.773010453362737, \
    -0.7071067811865476, 0.7071067811865476, -0.773010453362737, 0.6343932841636455, \
    -0.8314696123025452, 0.5555702330196022, -0.881921264348355, 0.47139673682599764, \
    -0.9238795325112867, 0.3826834323650898, -0.9569403357322088, 0.2902846772544624, \
    -0.9807852804032304, 0.19509032201612828, -0.9951847266721969, 0.0980171403295606, \
    -1.0, 0.0, -0.9951847266721969, -0.0980171403295606, \
    -0.9807852804032304, -0.19509032201612828, -0.9569403357322088, -0.2902846772544624, \
    -0.9238795325112867, -0.3826834323650898, -0.881921264348355, -0.47139673682599764, \
    -0.8314696123025452, -0.5555702330196022, -0.773010453362737, -0.6343932841636455, \
    -0.7071067811865476, -0.7071067811865476, -0.6343932841636455, -0.773010453362737, \
    -0.5555702330196022, -0.8314696123025452, -0.47139673682599764, -0.881921264348355, \
    -0.3826834323650898, -0.9238795325112867, -0.2902846772544624, -0.9569403357322088, \
    -0.19509032201612828, -0.9807852804032304, -0.0980171403295606, -0.9951847266721969, \
    0.0, -1.0, 0.0980171403295606, -0.9951847266721969, \
    0.19509032201612828, -0.9807852804032304, 0.2902846772544624, -0.9569403357322088, \
    0.3826834323650898, -0.9238795325112867, 0.47139673682599764, -0.881921264348355, \
    0.5555702330196022, -0.8314696123025452, 0.6343932841636455, -0.773010453362737, \
    0.7071067811865476, -0.7071067811865476, 0.773010453362737, -0.6343932841636455, \
    0.8314696123025452, -0.5555702330196022, 0.881921264348355, -0.47139673682599764, \
    0.9238795325112867, -0.3826834323650898, 0.9569403357322088, -0.2902846772544624, \
    0.9807852804032304, -0.19509032201612828, 0.9951847266721969, -0.0980171403295606

RC_HD void sincos_table(double u, const double* tab, double& s, double& c) {
    // u = angle / (pi/32), formed by the caller as (lambda_k - lambda_0) * (T * 32/pi): its rounding is the same
    // ulp(angle) the plain product T * dlambda would carry, and r = u - n is then EXACT, so the two reduction fmas
    // are not needed; pi/32 is folded into the Taylor coefficients.
    const double n = rint(u);
    const double r = u - n;
    const double z = r * r;
    double ps = fma(z, -1.7440893260086657e-11, 7.600081085793214e-08);
    ps = fma(z, ps, -1.577060784927359e-04);
    ps = fma(z, ps, 9.817477042468103e-02);
    const double sl = r * ps;
    double pc = fma(z, 2.140319614762968e-13, -1.2435603596778489e-09);
    pc = fma(z, pc, 3.870689512650269e-06);
    pc = fma(z, pc, -4.819142773969413e-03);
    const double cl = fma(z, pc, 1.0);
    const int k = ((int)n) & 63;
    const double ch = tab[2 * k], sh = tab[2 * k + 1];
    s = fma(sh, cl, ch * sl);
    c = fma(ch, cl, -sh * sl);
}
constexpr double kTurnsPerRadian = 1.0185916357881302e+01;     // 32 / pi

// R = number of REAL row vectors carried through the rotations: 0 (eigenvalues only), 2 (rows `in`, `out` of the
// eigenvector matrix: z[0], z[1]) or 4 (the same two rows of a COMPLEX accumulated transformation, re / im planes - the
// ring kernel, hermitian_core.h).
template <int N, int R = 2>
struct TriEig {
    double d[N];                    // diagonal -> eigenvalues
    double e[N];                    // e[i] couples sites i and i+1; e[N-1] is padding (0)
    double z[R > 0 ? R : 1][N];     // rows of the accumulated eigenvector matrix
};

// Wave-level votes.  On the device the QL control flow is WAVE-UNIFORM (one sample per lane, the 64 samples
// of a wave share a controller and converge almost in lock-step); on the host a "wave" is one sample - unless the
// translation unit defines RC_HOST_WAVE: then the votes go through rc_host_wave::ballot, which tests/host/host_wave.cpp
// implements over up to 64 host threads running one lane each in lock-step (round 4: what a wave-uniform decision does to
// the OTHER lanes of a tile - e.g. healthy samples sent through the tile-wide fp64 QL by a neighbour - is invisible to a
// one-sample "wave"; with the emulator the CPU suite sees it).
// (the ballot builtin keeps the predicate in a scalar mask register: __all() / __any() go through an i32 per lane -
// one v_cndmask + one v_cmp per vote, ~40 VALU instructions per tile)
RC_HD bool vote_all(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(v) == __builtin_amdgcn_ballot_w64(true);
#elif defined(RC_HOST_WAVE)
    return rc_host_wave::ballot(!v) == 0ull;
#else
    return v;
#endif
}
RC_HD bool vote_any(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(v) != 0ull;
#elif defined(RC_HOST_WAVE)
    return rc_host_wave::ballot(v) != 0ull;
#else
    return v;
#endif
}

// Lane masks (wave-uniform, scalar registers on the device; one bit on the host where a "wave" is one sample): the sweep
// loops keep their per-lane "converged" / "hit the cap" flags in them - a per-lane bool carried around a loop costs a
// v_cndmask + v_cmp per vote.
typedef unsigned long long lanemask_t;
RC_HD lanemask_t lane_ballot(bool v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(v);
#elif defined(RC_HOST_WAVE)
    return rc_host_wave::ballot(v);
#else
    return v ? 1ull : 0ull;
#endif
}
RC_HD bool lane_bit(lanemask_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (m >> (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))) & 1ull;
#elif defined(RC_HOST_WAVE)
    return (m >> rc_host_wave::lane()) & 1ull;
#else
    return m & 1ull;
#endif
}

// Implicit QL with Wilkinson shift on a real symmetric tridiagonal matrix, two eigenvector rows - FAST PATH.
//
// Control flow is wave-uniform.  For the eigenvalue index l the wave sweeps until EVERY lane has a negligible
// e[l]; a lane that converged early keeps sweeping, which for it is a valid QL step on the remaining block
// [l+1, N-1] (its rotation at i = l is the identity up to sign because f = s e[l] ~ 0) - harmless, and it
// advances the later eigenvalues.  Every sweep covers the static window [l, N-1] with no predication at all
// (every index a compile-time constant, one basic block per sweep, which lets the scheduler overlap the
// eigenvector-row updates of rotation i with the serial chase of rotation i-1, and cuts live registers from
// 106 to 72 at N = 7).  There is no scan for interior splits (l < m < N-1, e.g. a cut chain): an exactly cancelled
// coupling enters as 1e-150 (chain_fidelity_fast), and since the chase depends only on the RATIO f : g the sweeps
// rotate the block above it normally.  Returns false - per lane - when some eigenvalue of this lane does not converge
// within kFastSweepCap sweeps (never observed, cut chains included); the caller then recomputes that sample with
// tridiag_ql2_general.
// `tol_values`: split tolerance of the eigenvalue-only use (R = 0); the caller that must resolve close pairs passes kEps.
// FREEZE (round 5; the ring route's repair kernel): a lane whose e[l] is already negligible sits the wave's further sweeps for
// this l out (the sweep runs under the lanes' execution mask) - every lane then performs exactly the sweeps it would perform
// alone, so its result does not depend on which other samples share its wave.  The repair kernel packs the listed samples of
// ALL tiles into waves in arrival order; without this its results were reproducible to rounding only (1.6e-15 observed under
// concurrent streams).  Elsewhere a wave IS a tile of one controller, fixed by the input: the default stays unpredicated.
template <int N, int R, bool FREEZE = false>
RC_HD bool tridiag_ql2_fast(TriEig<N, R>& s, const double tol_values = kFastEpsValues) {
    constexpr bool VEC = R > 0;
    lanemask_t badm = 0ull;                        // lanes that ran into the sweep cap at some l
    const lanemask_t full = lane_ballot(true);
#pragma unroll
    for (int l = 0; l < N - 1; ++l) {
        if (kClosedForm2x2 && l == N - 2) {
            // last 2x2 block [[a, e], [e, b]] in closed form (one Jacobi rotation): eigenvalues a - t, b + t with
            // t = e tau, tau = e / (delta + sign(delta) rho), delta = (b - a)/2, rho = sqrt(delta^2 + e^2); the 1e-300
            // keeps the denominator nonzero when e = delta = 0 (then t = tau = 0).
            const double el = s.e[l];
            const double delta = 0.5 * (s.d[l + 1] - s.d[l]);
            double rho, rinv;
            sqrt_rsqrt(fma(delta, delta, fma(el, el, 1e-300)), rho, rinv);
            const double tau = el * rcp_full(delta + copysign(rho, delta));
            const double t = el * tau;
            s.d[l] -= t;
            s.d[l + 1] += t;
            s.e[l] = 0.0;
            if (VEC) {
                double tn, cs;
                sqrt_rsqrt(fma(tau, tau, 1.0), tn, cs);
                const double sn = tau * cs;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const double f = s.z[q][l + 1];
                    s.z[q][l + 1] = fma(sn, s.z[q][l], cs * f);
                    s.z[q][l] = fma(cs, s.z[q][l], -sn * f);
                }
            }
            break;
        }
        // Converged for this l when e[l] is negligible on EVERY lane.  Healthy spectra need <= 6 sweeps per
        // eigenvalue (measured max over 1.8e6 samples); a lane still not converged after kFastSweepCap sweeps is
        // marked bad (its result is discarded and recomputed by the general path), stops voting and keeps executing
        // the wave's sweeps - arithmetic on garbage.  Written as a bottom-tested loop with the cap folded into the
        // vote: ONE exit whose live-out values are the back-edge values (any other shape makes the compiler copy
        // the whole state once per sweep).
        const double tol = VEC ? kEps : tol_values;
        lanemask_t donem = lane_ballot(fabs(s.e[l]) <= tol * (fabs(s.d[l]) + fabs(s.d[l + 1])));
        if ((donem | badm) == full) continue;
        int iter = 0;
#pragma unroll 1
        do {
            bool act = true;                       // (FREEZE: see above; a compile-time constant otherwise)
            if (FREEZE) act = !(fabs(s.e[l]) <= tol * (fabs(s.d[l]) + fabs(s.d[l + 1])));
            if (act) {
            // Wilkinson shift from the leading 2x2 of the window: mu = d_l - e_l^2 / (delta + sign(delta) rho),
            // delta = (d_{l+1} - d_l)/2, rho = sqrt(delta^2 + e_l^2);  g = d_{N-1} - mu.  The 1e-300 keeps rho > 0
            // for a converged lane whose e_l and delta are both exactly zero.
            const double el = s.e[l];
            const double delta = 0.5 * (s.d[l + 1] - s.d[l]);
#if RC_SHIFT_NODIV
            const double rho = sqrt_fast(fma(delta, delta, fma(el, el, 1e-300)));     // the nudge rides in the fma
#else
            const double e2 = el * el;
            const double rho = sqrt_fast(fma(delta, delta, e2) + 1e-300);
#endif
#if RC_SHIFT_NODIV
            // e^2 / (delta + sign(delta) rho) = sign(delta) (rho - |delta|): no reciprocal.  The cancellation costs
            // nothing that matters: the shift then carries the seed's 5e-8 relative to rho instead of to the
            // correction, which still contracts e_l by ~1e-7 per sweep once it is small (host emulation: +1 % rotations).
            double g = s.d[N - 1] - s.d[l] + copysign(rho - fabs(delta), delta);
#else
            double g = s.d[N - 1] - s.d[l] + e2 * rcp_fast(delta + copysign(rho, delta));
#endif
            double sn = 1.0, cs = 1.0, p = 0.0;
#pragma unroll
            for (int i = N - 2; i >= l; --i) {
                double f = sn * s.e[i];
                const double b = cs * s.e[i];
                // Rotation annihilating the bulge: r = hypot(f, g), s = f/r, c = g/r.  g is nudged by 1e-150
                // (a no-op unless |g| < 1e-134) so that f = g = 0 gives the identity rotation without any
                // compare/select; the neglected bulge is then < 1e-150.
                const double gn = g + 1e-150;
                double r, rinv;
                sqrt_rsqrt(fma(f, f, gn * gn), r, rinv);
                if (i + 1 <= N - 2) s.e[i + 1] = r;        // compile-time condition
                sn = f * rinv;
                cs = gn * rinv;
                g = s.d[i + 1] - p;
                r = fma(s.d[i] - g, sn, 2.0 * cs * b);
                p = sn * r;
                s.d[i + 1] = g + p;
                g = fma(cs, r, -b);
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    f = s.z[q][i + 1];
                    s.z[q][i + 1] = fma(sn, s.z[q][i], cs * f);
                    s.z[q][i] = fma(cs, s.z[q][i], -sn * f);
                }
            }
            s.d[l] -= p;
            s.e[l] = g;
            }
            ++iter;
            donem = lane_ballot(fabs(s.e[l]) <= tol * (fabs(s.d[l]) + fabs(s.d[l + 1])));
            if (iter >= kFastSweepCap) badm |= full & ~donem;
        } while ((donem | badm) != full);
    }
    return !lane_bit(badm);
}

// fp32 twin of tridiag_ql2_fast<N, 0>: eigenvalues only, same wave-uniform control flow, same nudges scaled to the
// fp32 range; v_rsq_f32 is accurate to 1 ulp, so a rotation needs no refinement (rsq + 17 operations).  d -> the
// eigenvalues to ~1e-6 (absolute, |d| ~ 10).  Returns false - per lane - on the sweep cap.
template <int N>
RC_HD bool tridiag_ql_f32(float (&d)[N], float (&e)[N], float& scale_out) {
    lanemask_t badm = 0ull;                        // lanes that ran into the sweep cap at some l
    const lanemask_t full = lane_ballot(true);
    // ABSOLUTE split threshold, kF32SplitTol x the size of the matrix: what dropping e_l costs is e_l^2 / gap whatever
    // the neighbouring diagonal entries are, and the usual relative test (|d_l| + |d_l+1|) would make the one lane of the
    // tile whose d_l happens to sit near zero hold all 64 in extra sweeps
    float scale = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) scale = fmaxf(scale, fabsf(d[i]));
#pragma unroll
    for (int i = 0; i < N - 1; ++i) scale = fmaxf(scale, fabsf(e[i]));
    const float thr = kF32SplitTol * scale;
    scale_out = scale;
#pragma unroll
    for (int l = 0; l < N - 1; ++l) {
        if (l == N - 2) {                          // last 2x2 block in closed form
            const float el = e[l];
            const float delta = 0.5f * (d[l + 1] - d[l]);
            const float h = fmaf(delta, delta, fmaf(el, el, 1e-30f));
            const float t = copysignf(h * seed_rsqf(h) - fabsf(delta), delta);
            d[l] -= t;
            d[l + 1] += t;
            break;
        }
        lanemask_t donem = lane_ballot(fabsf(e[l]) <= thr);
        if ((donem | badm) == full) continue;
        int iter = 0;
#pragma unroll 1
        do {
            const float el = e[l];
            const float delta = 0.5f * (d[l + 1] - d[l]);
            const float h0 = fmaf(delta, delta, fmaf(el, el, 1e-30f));
            float g = d[N - 1] - d[l] + copysignf(h0 * seed_rsqf(h0) - fabsf(delta), delta);
            float sn = 1.0f, cs = 1.0f, p = 0.0f;
#pragma unroll
            for (int i = N - 2; i >= l; --i) {
                const float f = sn * e[i];
                const float b = cs * e[i];
                const float gn = g + 1e-15f;       // f = g = 0 -> identity rotation without compare / select
                const float h = fmaf(f, f, gn * gn);
                const float rinv = seed_rsqf(h);
                if (i + 1 <= N - 2) e[i + 1] = h * rinv;
                sn = f * rinv;
                cs = gn * rinv;
                g = d[i + 1] - p;
                const float r = fmaf(d[i] - g, sn, 2.0f * cs * b);
                p = sn * r;
                d[i + 1] = g + p;
                g = fmaf(cs, r, -b);
            }
            d[l] -= p;
            e[l] = g;
            ++iter;
            donem = lane_ballot(fabsf(e[l]) <= thr);
            if (iter >= kFastSweepCap) badm |= full & ~donem;
        } while ((donem | badm) != full);
    }
    return !lane_bit(badm);
}

// (Round 3 tried Reinsch's rational QL - EISPACK tqlrat: squared couplings, two reciprocals, no square root - for these
// starting values: in isolation a step costs 0.67 of the rotation (scripts/ubench/ql32_body), but inside this function the
// compiler already has the rotation at 16 instructions against 14 + two quarter-rate reciprocals + the explicit shift, the
// starts come out ~12 % less accurate, and the kernels ran 0 ... +4 % SLOWER (vii).  Not kept;
// prototype: scripts/proto/qlrat_start.py.)

// One Halley step per eigenvalue on chi(mu) = det(mu I - T) of the fp64 tridiagonal (diag d0, SQUARED couplings e0sq):
//   p_{m+1} = (mu - d_m) p_m - e_{m-1}^2 p_{m-1},  p' and q = p''/2 by the differentiated recurrences,
//   mu <- mu - p p' / (p'^2 - p q).
// The recurrence is the Sturm sequence: its computed value is the exact chi of a matrix perturbed by a few ulp in d and
// e^2, so the converged root carries the same ~N eps |T| error as a QL eigenvalue.  Returns max_k |step_k|.
// CRITICAL-POINT GUARD (every path, round 3): between two close eigenvalues chi' vanishes and the step, ~ -2 (mu - c) near
// that critical point c, is tiny WITHOUT mu being a root (c repels: the iterates leave it by a factor 3 per step).  There
// |chi chi''/2| is not small against chi'^2, i.e. |p q| / (p'^2 - p q) is O(1) where a start inside the basin of a root
// has ~ step * sum_j 1/(mu - lam_j) << 1.  `crit` returns max_k |p q / den|; callers treat crit > kHalleyCritical as "not
// converged, keep stepping" whatever the step sizes say (1 multiplication + 1 max per eigenvalue).
// `roots` (stepping path only, SELECT = true): bit k set = this lane wants eigenvalue k stepped; an eigenvalue no lane of
// the tile wants is skipped wave-uniformly (typically ONE lane of a flagged tile has ONE close pair: 2 of the N chains run).
#ifndef RC_HALLEY_CRITICAL
#define RC_HALLEY_CRITICAL 0.2
#endif
constexpr double kHalleyCritical = RC_HALLEY_CRITICAL;
// The polynomial whose roots are wanted, as a functor: eval(mu) -> chi, chi', chi''/2; trace() = sum of the roots.
// ChainChi: det(mu I - T) of the real symmetric tridiagonal (d0, SQUARED couplings e0sq), three-term recurrences.
template <int N>
struct ChainChi {
    const double (&d0)[N];
    const double (&e0sq)[N];
    RC_HD double trace() const {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) t += d0[i];
        return t;
    }
    RC_HD void eval(const double mu, double& p_out, double& dp_out, double& q_out) const {
        double pm = 1.0, p = mu - d0[0];           // p_0, p_1
        double dm = 0.0, dp = 1.0;                 // p'_0, p'_1
        double qm = 0.0, q = 0.0;                  // q_0, q_1
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
        }
        p_out = p;
        dp_out = dp;
        q_out = q;
    }
    RC_HD void eval2(const double mu, double& p_out, double& dp_out) const {      // chi and chi' only (5 operations per site)
        double pm = 1.0, p = mu - d0[0];
        double dm = 0.0, dp = 1.0;
#pragma unroll
        for (int m = 1; m < N; ++m) {
            const double t = mu - d0[m];
            const double c = e0sq[m - 1];
            const double pn = fma(t, p, -c * pm);
            const double dn = fma(t, dp, fma(-c, dm, p));
            pm = p; p = pn;
            dm = dp; dp = dn;
        }
        p_out = p;
        dp_out = dp;
    }
};

template <int N, bool SELECT = false, typename Chi>
RC_HD double halley_polish(const Chi& chi, double (&lam)[N], double& crit, unsigned roots = ~0u) {
    double maxd = 0.0, maxc = 0.0;
    double rest = chi.trace();                     // trace - sum of the polished eigenvalues
    // N - 1 eigenvalues are polished; the last one is what the trace leaves (2N additions instead of 7N operations;
    // it inherits the summed error of the others, ~N 1e-14)
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        if (SELECT && !vote_any((roots >> k) & 1u)) {
            rest -= lam[k];
            continue;
        }
        const double mu = lam[k];
        double p, dp, q;
        chi.eval(mu, p, dp, q);
        const double pq = p * q;
        const double den = fma(dp, dp, -pq);
        double y = seed_rcp(den);
        y = fma(y, fma(-den, y, 1.0), y);
        const double step = (p * dp) * y;
        lam[k] = mu - step;
        rest -= lam[k];
        maxd = fmax(maxd, fabs(step));
        maxc = fmax(maxc, fabs(pq * y));
    }
    lam[N - 1] = rest;
    crit = maxc;
    return maxd;
}

// The FIRST step of the mixed path (round 3): an Ehrlich-Aberth step instead of the Halley step,
//   mu <- mu - p / (p' - p S_k),   S_k = sum_{j != k} 1 / (start_k - start_j)   (fp32, from the starting values themselves).
// chi''/(2 chi') = sum_{j != k} 1/(mu - lam_j) + O(mu - lam_k): the Halley step's second derivative is replaced by what the
// OTHER starting values already say about it - with exact neighbours the step lands on lam_k exactly (deflation), with
// neighbours off by delta_j the error after the step is (mu - lam_k)^2 sum_j delta_j / (mu - lam_j)^2 <= step^3 (N-1) / gap^2:
// the same cubic bound, so the acceptance rule is unchanged.  What it saves: the q recurrence (2 of the 7 fp64 operations
// per (eigenvalue, site)) for N (N-1) / 2 fp32 reciprocals (shared by both eigenvalues of a pair: 1/(a-b) = -1/(b-a)).
// The guard: a small step here is p / p' / (1 - rho), rho = p S_k / (p' - p S_k); with |rho| <= kHalleyCritical the Newton
// step |chi/chi'| is small too, and |chi'/chi| = |sum_j 1/(mu - lam_j)| <= N / min_j |mu - lam_j| puts a root within N steps
// of mu RIGOROUSLY (at a critical point chi' = 0 makes rho = -1).  `crit` returns max_k |rho_k| as for halley_polish.
template <int N, typename Chi>
RC_HD double aberth_polish(const Chi& chi, const float (&S)[N], double (&lam)[N], double& crit) {
    double maxd = 0.0, maxc = 0.0;
    double rest = chi.trace();
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        const double mu = lam[k];
        double p, dp;
        chi.eval2(mu, p, dp);
        const double ps = p * (double)S[k];
        const double den = dp - ps;
        double y = seed_rcp(den);
        y = fma(y, fma(-den, y, 1.0), y);
        const double step = p * y;
        lam[k] = mu - step;
        rest -= lam[k];
        maxd = fmax(maxd, fabs(step));
        maxc = fmax(maxc, fabs(ps * y));
    }
    // the last eigenvalue takes no step of its own, but its starting value sits in every S_k: what the trace moves it by
    // is its step as far as the acceptance rule is concerned (a poor last start would otherwise go unnoticed)
    maxd = fmax(maxd, fabs(rest - lam[N - 1]));
    lam[N - 1] = rest;
    crit = maxc;
    return maxd;
}
#ifndef RC_ABERTH_FIRST
#define RC_ABERTH_FIRST 1
#endif
constexpr bool kAberthFirst = RC_ABERTH_FIRST;

// A NEWTON step for every eigenvalue (chi, chi' only: 5 fp64 operations per (eigenvalue, site)), all N - 1 recurrence chains
// independent of each other - the scheduler interleaves them, where the stepping loop's SELECTed Halley chains are separated by
// wave-uniform branches and each runs at the dependent-issue rate.  The SECOND step of a tile that failed the one-step
// acceptance (round 5, RC_STEP2_NEWTON_ALL): after the Aberth step a flagged eigenvalue is ~step^3 (N-1)/gap^2 from its root,
// Newton's quadratic convergence finishes it: a step s from inside the basin leaves ~ s^2 sum_j 1/|lam_k - lam_j| <=
// s^2 (N-1)/gap.  A step that is not small (or not finite) is NOT taken - the iterate stays what it was for the stepping loop
// behind - and reported as 1e300.  Returns max_k |step_k| (the last eigenvalue: what the trace moves it by).
template <int N, typename Chi>
RC_HD double newton_polish_all(const Chi& chi, double (&lam)[N], double take_below) {
    double maxd = 0.0;
    double rest = chi.trace();
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        const double mu = lam[k];
        double p, dp;
        chi.eval2(mu, p, dp);
        double y = seed_rcp(dp);
        y = fma(y, fma(-dp, y, 1.0), y);
        const double step = p * y;
        const bool take = fabs(step) <= take_below;            // (false for NaN / inf)
        lam[k] = take ? mu - step : mu;
        rest -= lam[k];
        maxd = fmax(maxd, take ? fabs(step) : 1e300);
    }
    maxd = fmax(maxd, fabs(rest - lam[N - 1]));
    lam[N - 1] = rest;
    return maxd;
}
#ifndef RC_STEP2_NEWTON_ALL
#define RC_STEP2_NEWTON_ALL 0
#endif
#ifndef RC_STEP_ALL_FIRST
#define RC_STEP_ALL_FIRST 0
#endif

// Mixed-precision eigenvalues, the fp64 half: from fp32 starting values `start` (the fp32 QL's eigenvalues, ~1e-6 of the
// spectral scale; `ok32` = false when that QL hit its sweep cap: the starts are then arbitrary) to the eigenvalues of the
// polynomial `chi` (ChainChi: the fp64 tridiagonal (d0, e0sq); RingChi, hermitian_core.h: the ring) at rounding level in `lam`.  Returns true when `lam` is settled; false - per lane - when the
// caller must escalate (all-fp64 QL for the tile).  `scale32` = max(|d|, |e|) of the matrix (fp32 QL's by-product): the
// fp32 uncertainty of a computed gap is kGapUlps32 * FLT_EPSILON * scale32 (= 4.3e-6 at the benchmark's scale of ~12).
//   1. ONE Halley step per eigenvalue; accepted when  max|step|^3 <= kHalleyAccept * (g32 - uncertainty)^2  (the error
//      bound of a Halley step, ~ step^3 / gap^2, g32 = smallest gap of the fp32 spectrum) AND no start sat next to a
//      critical point of chi (`crit`, see halley_polish).
//   2. otherwise (wave-uniform: the whole tile) the flagged eigenvalues keep stepping until the step itself is <= 1e-9
//      and off the critical points - the iterate before a tiny step was converged (error after a step of 1e-9:
//      1e-27 / gap^2) -, up to 12 steps; then the fp64 gaps are checked so that no two starts fell into the same
//      eigenvalue (closer than 4e-6 of the scale).
// `extra_steps` (diagnostic, optional) is set when the tile left the one-step path.
#ifndef RC_ALWAYS_DISTINCT_CHECK
#define RC_ALWAYS_DISTINCT_CHECK 0
#endif
constexpr float kGapUlps32 = 3.0f;
template <int N, typename Chi>
RC_HD bool mixed_refine(const Chi& chi, const float (&start)[N], float scale32, bool ok32,
                        double (&lam)[N], int* extra_steps = nullptr) {
    const float unc = kGapUlps32 * 1.1920929e-7f * scale32;
    // smallest gap of the spectrum, from the fp32 eigenvalues (all the step bound below needs)
    float g32 = 1e30f;
    float S[N];                                           // Aberth sums of the starting values (aberth_polish)
#pragma unroll
    for (int k = 0; k < N; ++k) S[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int m = k + 1; m < N; ++m) {
            const float diff = start[k] - start[m];
            g32 = fminf(g32, fabsf(diff));
            if (kAberthFirst) {
                const float r = seed_rcpf(diff);
                S[k] += r;
                S[m] -= r;
            }
        }
    }
    const float g32c = fmaxf(g32 - unc, 0.0f);            // less the fp32 uncertainty of a difference
    const double gap2 = kHalleyAccept * ((double)g32c * (double)g32c);
#pragma unroll
    for (int k = 0; k < N; ++k) lam[k] = (double)start[k];
    double crit;
    double maxd;
    if (kAberthFirst) {
        // two starts closer than the fp32 uncertainty (possibly equal: 1/0): the sums mean nothing and the sample is not
        // accepted whatever the step says (gap2 = 0) - a plain Newton step keeps its iterate finite for the stepping path
        const bool sep = g32c > 0.0f;
#pragma unroll
        for (int k = 0; k < N; ++k) S[k] = sep ? S[k] : 0.0f;
        maxd = aberth_polish<N>(chi, S, lam, crit);
    } else {
        maxd = halley_polish<N>(chi, lam, crit);
    }
    bool need = !(maxd * maxd * maxd <= gap2) || !(crit <= kHalleyCritical) || !ok32;
#ifdef RC_EXPERIMENT_NO_STEPPING
    // TIMING EXPERIMENT ONLY (scripts/build_variant.sh nostep; results of flagged samples are wrong): the kernel without its
    // stepping path = the most that deferring flagged samples to a second launch could save
    return true;
#endif
    if (RC_LIKELY(!vote_any(need))) return true;
#if defined(RC_EXPERIMENT_STEP_NOT_RUN) && defined(__HIP_DEVICE_COMPILE__)
    {   // TIMING EXPERIMENT ONLY: the stepping path (and everything behind it) compiled in but never executed - an opaque
        // always-true flag the optimiser cannot see through; separates what the code's PRESENCE costs from what running it costs
        int never = 1;
        asm volatile("" : "+s"(never));
        if (never) return true;
    }
#endif
    if (extra_steps) *extra_steps = 1;
#if RC_STEP2_NEWTON_ALL
    {
        // (round 5) Before any bookkeeping: ONE Newton step for every eigenvalue of every lane, chains interleaved.  Accepted -
        // wave-wide - when (N-1) s^2 <= kHalleyAccept (g32 - uncertainty) for the largest step s of the sample (error after a
        // Newton step from inside the basin: s^2 sum_j 1/|lam_k - lam_j| <= s^2 (N-1)/gap; s <= 1e-7 sqrt(gap/(N-1)) also puts
        // the iterate N s << gap from its root - |chi'/chi| <= N / min_j|mu - lam_j| - so the root it converged to is isolated),
        // the first step was off the critical points and the fp32 QL had converged.  Otherwise the iterates - improved where the
        // step was small, untouched where it was not - go on into the stepping loop below as before.
        const double maxd2 = newton_polish_all<N>(chi, lam, 1e-3 * (double)fmaxf(scale32, 1.0f));
        const bool ok2 = ok32 && (crit <= kHalleyCritical) && ((double)(N - 1) * maxd2 * maxd2 <= kHalleyAccept * (double)g32c);
        if (!vote_any(!ok2)) {
            if (extra_steps) *extra_steps = 100;           // (diagnostic: settled by the all-eigenvalue Newton step)
            float moved2 = 0.0f;
#pragma unroll
            for (int k = 0; k < N; ++k) moved2 = fmaxf(moved2, (float)fabs((double)start[k] - lam[k]));
            const float res2 = 4e-6f * fmaxf(scale32, 1.0f);
            bool dup = false;
            if (RC_ALWAYS_DISTINCT_CHECK || vote_any(!(g32 > 2.0f * moved2 + res2))) {
                float lf2[N], mingap2 = 1e30f;
#pragma unroll
                for (int k = 0; k < N; ++k) lf2[k] = (float)lam[k];
#pragma unroll
                for (int k = 0; k < N; ++k) {
#pragma unroll
                    for (int m = k + 1; m < N; ++m) mingap2 = fminf(mingap2, fabsf(lf2[k] - lf2[m]));
                }
                dup = !(mingap2 > res2);
            }
            return !dup;
        }
        maxd = fmin(maxd, 1e10);       // (the bookkeeping below reads the FIRST step's size: unchanged)
    }
#endif
    // Rare per sample, not per tile (close pair or a poor fp32 start somewhere among the 64): which eigenvalues - the step
    // bound again, per eigenvalue, with ITS gap to the nearest other one and (round 4) ITS OWN step: what the first step
    // leaves of eigenvalue k's error is own_k^2 * (largest error among the other starts) * (N-1) / gap_k^2 (Aberth: the
    // deflation sums carry the others' errors; Halley: own_k^3), so `own_k^2 maxd` stands in where round 3 had maxd^3 - a
    // sample with one poor start no longer drags every eigenvalue with a modest gap into the stepping loop (N = 10 XXZ:
    // the union of wanted chains per flagged tile drops from ~4 to ~2.5).  own_k = |start_k - lam_k|: both are live anyway.
    // The bookkeeping runs in fp32 on the current iterate (a gap only has to be known to ~1e-6 of the scale); a lane whose
    // fp32 QL failed, or with a start at a critical point, wants all of them.
    unsigned roots = 0u;
    float moved = 0.0f;                                   // largest |start_k - lam_k| of this sample so far
    bool settled = false;
    const double maxd1 = maxd;                            // the FIRST step's size: what the bookkeeping's error estimate is written in
#if RC_STEP_ALL_FIRST
    {
        // (round 5 experiment) the first stepping iteration for EVERY eigenvalue - N - 1 independent Halley chains the scheduler can
        // interleave, where the SELECTed chains below run one after the other behind wave-uniform branches - and no bookkeeping
        // unless a second iteration is needed (83 % of the flagged tiles need one)
#pragma unroll
        for (int k = 0; k < N; ++k) moved = fmaxf(moved, (float)fabs((double)start[k] - lam[k]));
        maxd = halley_polish<N>(chi, lam, crit);
        moved += (float)fmin(maxd, 1e10);
        need = !(maxd <= 1e-9) || !(crit <= kHalleyCritical);
        if (extra_steps) *extra_steps = 2;
        settled = !vote_any(need);
    }
#endif
    if (!settled) {
        float lf[N];
#pragma unroll
        for (int k = 0; k < N; ++k) lf[k] = (float)lam[k];
        float gk[N];
#pragma unroll
        for (int k = 0; k < N; ++k) gk[k] = 1e30f;
#pragma unroll
        for (int k = 0; k < N; ++k) {
#pragma unroll
            for (int m = k + 1; m < N; ++m) {
                const float df = fabsf(lf[k] - lf[m]);
                gk[k] = fminf(gk[k], df);
                gk[m] = fminf(gk[m], df);
            }
        }
        const float mx = (float)fmin(maxd1, 1e10);
        const bool all = !ok32 || !(crit <= kHalleyCritical);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float own = (float)fabs((double)start[k] - lam[k]);      // (in fp64: a step of 1e-8 is below fp32 resolution)
            moved = fmaxf(moved, own);
            const float g = fmaxf(gk[k] - unc, 0.0f);
            // (the factor N - 1 of the bound is kept here: with every eigenvalue judged on its own, several can sit AT their
            // bound at once, and the last eigenvalue - what the trace leaves - collects all their errors)
            roots |= (!((float)(N - 1) * own * own * mx <= (float)kHalleyAccept * (g * g)) || all) ? (1u << k) : 0u;
        }
    }
#if defined(RC_FLAG_STATS) && !defined(__HIP_DEVICE_COMPILE__)
    rc_flag_stats_hook(N, roots, maxd, lam);              // (scripts/proto/flag_stats.cpp: host-side statistics of the stepping path)
#endif
#pragma unroll 1
    for (int it = 0; it < 12 && !settled; ++it) {         // (settled: wave-uniform)
        maxd = halley_polish<N, true>(chi, lam, crit, roots);
        moved += (float)fmin(maxd, 1e10);
        need = !(maxd <= 1e-9) || !(crit <= kHalleyCritical);
        if (extra_steps) *extra_steps = 2 + it;           // (diagnostic: 1 + stepping iterations executed)
        if (!vote_any(need)) break;
    }
    // distinct roots: two starts that fell into the SAME eigenvalue agree to ~1e-9; genuinely distinct eigenvalues closer
    // than ~4e-6 of the scale go to the all-fp64 QL.  Skipped - wave-uniformly - when it cannot fail: every iterate ended
    // within `moved` of its start, so with the smallest START gap above 2 moved + the resolution no two of them coincide.
    const float res = 4e-6f * fmaxf(scale32, 1.0f);
    if (RC_ALWAYS_DISTINCT_CHECK || vote_any(!(g32 > 2.0f * moved + res))) {
        float lf[N], mingap = 1e30f;
#pragma unroll
        for (int k = 0; k < N; ++k) lf[k] = (float)lam[k];
#pragma unroll
        for (int k = 0; k < N; ++k) {
#pragma unroll
            for (int m = k + 1; m < N; ++m) mingap = fminf(mingap, fabsf(lf[k] - lf[m]));
        }
        need = need || !(mingap > res);
    }
    return !need;
}

// Eigenvector weights w_k = Q[in,k] Q[out,k] WITHOUT eigenvectors, from the adjugate of (lambda I - T) of an
// unreduced symmetric tridiagonal T (diag d0, couplings e0), i = min(in,out), j = max(in,out):
//     w_k = (prod_{m=i}^{j-1} e0_m) * phi_i(lam_k) * psi_{j+1}(lam_k) / prod_{m != k} (lam_k - lam_m)
// phi_i = characteristic polynomial of the leading i x i block, psi_{j+1} of the trailing block below j (three-term
// recurrences).  The relative error of a close pair's difference enters both of its weights identically and
// multiplies only their (tiny, ~T*gap) joint contribution, so the result is as accurate as with accumulated
// eigenvectors (numpy prototype: <= 3e-13 on random, near-degenerate, resonant and graded spectra).  Returns false
// (per sample) when two computed eigenvalues are closer than 1e-7 of the spectral scale: such samples go to the
// general path.
template <int N, bool GAPS = true>
RC_HD bool ends_weights(double pe, const double (&lam)[N], double (&w)[N], double gap_tol = kDegenerateGapNoMix);

// d0: original diagonal; e0sq: SQUARED original couplings; i <= j: the two sites; pe: prod_{m=i}^{j-1} e0_m.
// Written as three strictly sequential phases over one pair of work arrays (w = pe / chi', w *= phi_i, w *= psi_j+1)
// so that at most 6N doubles are live at any point (the first version kept 9N alive and spilled from N = 7).
template <int N, bool GAPS = true>
RC_HD bool adjugate_weights(const double (&d0)[N], const double (&e0sq)[N], const double (&lam)[N], int i, int j,
                            double pe, double (&w)[N]) {
    const bool ok = ends_weights<N, GAPS>(pe, lam, w, kDegenerateGapAdjugate);
    double a[N], b[N];
    if (i > 0) {                                      // wave-uniform: in / out are kernel arguments
#pragma unroll
        for (int k = 0; k < N; ++k) {
            a[k] = lam[k] - d0[0];                    // phi_1
            b[k] = 1.0;                               // phi_0
        }
#pragma unroll
        for (int m = 1; m < N - 1; ++m) {
            if (m < i) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double t = fma(lam[k] - d0[m], a[k], -e0sq[m - 1] * b[k]);
                    b[k] = a[k];
                    a[k] = t;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] *= a[k];
    }
    if (j < N - 1) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            a[k] = lam[k] - d0[N - 1];                // psi_{N-1}
            b[k] = 1.0;                               // psi_N
        }
#pragma unroll
        for (int m = N - 2; m >= 1; --m) {
            if (m > j) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double t = fma(lam[k] - d0[m], a[k], -e0sq[m] * b[k]);
                    b[k] = a[k];
                    a[k] = t;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] *= a[k];
    }
    return ok;
}

// GENERAL PATH (rare): the textbook per-sample implicit QL with a per-sample window [l, m], runtime N, plain
// loops over runtime-indexed arrays.  Same arithmetic primitives as the fast path.  `Vec` is anything with
// operator[] returning double& (plain arrays on the host; lane-strided LDS views on the device, so that the
// kernel needs no scratch memory).  e has n entries (e[n-1] = 0 padding).
template <typename Vec>
RC_HD void tridiag_ql2_general(int n, Vec d, Vec e, Vec za, Vec zb) {
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
                double f = sn * e[i];
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
                const double a1 = za[i + 1], a0 = za[i];
                za[i + 1] = fma(sn, a0, cs * a1);
                za[i] = fma(cs, a0, -sn * a1);
                const double b1 = zb[i + 1], b0 = zb[i];
                zb[i + 1] = fma(sn, b0, cs * b1);
                zb[i] = fma(cs, b0, -sn * b1);
            }
            d[l] = d[l] - p;
            e[l] = g;
            e[m] = 0.0;
        }
    }
}

// End-to-end transfer ({in,out} = {0,N-1}): phi = psi = 1, so w_k = prod(e0) / prod_{m != k}(lam_k - lam_m).
// GAPS = false: the caller has already ruled out close pairs (mixed-precision path: fp32 gaps + Halley step size).
template <int N, bool GAPS>
RC_HD bool ends_weights(double pe, const double (&lam)[N], double (&w)[N], double gap_tol) {
    double mingap = 1e300, scale = 1.0;
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = 1.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (GAPS) scale = fmax(scale, fabs(lam[k]));
#pragma unroll
        for (int m = k + 1; m < N; ++m) {
            const double df = lam[k] - lam[m];
            if (GAPS) mingap = fmin(mingap, fabs(df));
            w[k] *= df;
            w[m] *= -df;
        }
    }
    bool ok = !GAPS || mingap > gap_tol * scale;
    if (kBatchInverse && N >= 3 && N <= 13) {         // (above: the prefix arrays cost registers the N >= 14 kernels lack)
        // One reciprocal per BATCH of weights (prefix products, invert the total, peel off): 3(B-1) multiplications + 1
        // reciprocal instead of B reciprocals (a v_rcp_f64 costs 3.4 FMAs, its refinement 5 more).  A batch total is a
        // product of B (N-1) eigenvalue differences, at most (2 scale)^(B (N-1)); B is chosen so that B (N-1) <= 60
        // (all N weights in one batch up to N = 8, two batches at N = 9..11, ...): in range up to scale ~ 1e4; a lane
        // where it overflows or underflows anyway is sent to the general path.
        constexpr int B = (60 / (N - 1)) < 1 ? 1 : (60 / (N - 1));
#pragma unroll
        for (int k0 = 0; k0 < N; k0 += B) {
            const int k1 = (k0 + B < N) ? (k0 + B) : N;         // compile-time after unrolling
            double pre[B];
            pre[0] = w[k0];
#pragma unroll
            for (int k = k0 + 1; k < k1; ++k) pre[k - k0] = pre[k - k0 - 1] * w[k];
            const double tot = pre[k1 - k0 - 1];
            ok = ok && (fabs(tot) > 1e-280) && (fabs(tot) < 1e280);
            double r = rcp_full(tot);
#pragma unroll
            for (int k = k1 - 1; k > k0; --k) {
                const double wk = w[k];
                w[k] = pe * (r * pre[k - k0 - 1]);
                r *= wk;
            }
            w[k0] = pe * r;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] = pe * rcp_full(w[k]);
    }
    return ok;
}

// A-POSTERIORI GUARD of the eigenvalue-only weight modes (round 4).  Whatever route produced them, the weights w_k =
// Q[out,k] Q[in,k] and the eigenvalues must reproduce the first moments of the matrix itself,
//     sum_k w_k (lam_k - c)^m = ((H - c I)^m)[out,in],   m = 0, 1, 2,
// and for a tridiagonal (or periodic tridiagonal) H the right-hand sides cost O(1) from what is in registers anyway: with
// D = |out - in| hops along the chain (and N - D the other way round a ring), A / B the products of the couplings along
// the two ways,
//     m = 0:  [D = 0]
//     m = 1:  [D = 0] (d_lo - c)  +  [D = 1] A  +  [N - D = 1] B
//     m = 2:  [D = 0] ((d_lo - c)^2 + |h_left|^2 + |h_right|^2)  +  ([D = 1] A + [N - D = 1] B) (d_lo + d_hi - 2 c)
//             +  [D = 2] A  +  [N - D = 2] B.
// What it catches: the adjugate numerators are three-term recurrences evaluated beside their own roots; their rounding noise
// nu is not a smooth function of lam_k, and a close pair's weights come out wrong by (nu_a - nu_b) / gap - exactly the
// amount by which sum_k w_k misses its value (with EXACT numerators the moments below D hold identically for any set of
// distinct lam_k: Lagrange).  The thresholds kDegenerateGap* above were tuned on fuzz campaigns (and moved by 40x in
// round 3); this check makes the decision self-validating: a sample whose moments are off by more than kSumRuleTol
// reports false and takes the eigenvector route like any other bad sample.  c = lam_0 (the differences lam_k - lam_0 are
// formed for the phases anyway); cost: 4 (N - 1) operations in the phase loop + ~10.
#ifndef RC_SUM_RULE_GUARD
#define RC_SUM_RULE_GUARD 1
#endif
constexpr bool kSumRuleGuard = RC_SUM_RULE_GUARD;
#ifndef RC_SUM_RULE_TOL
#define RC_SUM_RULE_TOL 2e-12
#endif
constexpr double kSumRuleTol = RC_SUM_RULE_TOL;     // |moment residual| <= tol * (2 scale)^m
// How many of the three rules are evaluated (1 = m = 0 only).  m = 0 alone already sees the failure the guard exists for:
// independent noise e_a, e_b on the two weights of a close pair shows up as e_a + e_b; the component it cannot see
// (e_a = -e_b) changes the amplitude by e_a (exp(-i T lam_a) - exp(-i T lam_b)) ~ e_a T gap - harmless exactly where the
// noise is large (small gap).  m = 1, 2 close that gap formally (they weigh the errors with lam_k - lam_0 and its square).
// Default: m = 0 only.  Same-box A/B (profiles/r04_ab_guard.txt; two sessions, two boxes), no guard -> m = 0 -> all three:
// general adjugate N = 7 +0.2 ... +0.9 % -> +3.7 ... +3.9 %, ring N = 5 / 7 / 10 +0.9 ... +1.7 / +1.0 ... +1.7 / +0.2 ... +0.4 % ->
// +6.5 ... +7.9 / +3.1 ... +4.2 / +1.9 ... +2.9 % (the three rules also cost the odd ring sizes a wave of residency); on the
// adversarial configurations that found the round-3 errors (55 seeds x 150, host build) both variants end at the same worst
// case, 6.6e-12 (no guard: 2.7e-11), the three rules flagging 0.14 % more samples.
#ifndef RC_SUM_RULE_MOMENTS
#define RC_SUM_RULE_MOMENTS 1
#endif
constexpr int kSumRuleMoments = RC_SUM_RULE_MOMENTS;
// END-TO-END weights (kWeightsEnds) carry no guard: their numerator is the CONSTANT prod e, so every moment below N - 1 is
// Lagrange's identity sum_k lam_k^m / chi'(lam_k) = 0 - true for ANY set of distinct lam_k up to the rounding of the
// products themselves (1e-16 relative): there is no recurrence noise to catch.  (What such a check would cost the headline
// kernel read +3.3 % in the first A/B and -0.1 % in the second, profiles/r04_ab_guard.txt - inside the box-to-box noise; it
// stays off because it cannot fail, not because of its price.)
#ifndef RC_SUM_RULE_ENDS
#define RC_SUM_RULE_ENDS 0
#endif
constexpr bool kSumRuleEnds = RC_SUM_RULE_ENDS;

// d[idx] for a wave-uniform runtime index WITHOUT dynamic register indexing: a chain of selects on array elements is turned
// into a load from a selected address by the optimiser - the array then lives in scratch memory (measured: 16 N bytes of
// private segment in every adjugate-mode kernel) - so the element is masked out arithmetically: N multiply-adds with a
// scalar 0 / 1, on a wave-uniform branch that only the neighbouring / same-site (in, out) pairs take.  (A non-finite entry
// poisons the sum: the guard then rejects the sample, which is the right answer for it anyway.)
template <int N>
RC_HD double pick_site(const double (&d)[N], int idx) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) v = fma(d[i], (i == idx) ? 1.0 : 0.0, v);
    return v;
}

// Right-hand sides of the three moment rules for a real tridiagonal chain (B = 0): see the block comment above.
// `e2l`, `e2r`: squared couplings left / right of site lo (0 at the chain ends); only read for D = 0.
RC_HD void chain_moment_rhs(int D, double dlo, double dhi, double e2l, double e2r, double A, double c,
                            double& m0, double& m1, double& m2) {
    m0 = (D == 0) ? 1.0 : 0.0;
    const double x = dlo - c;
    m1 = (D == 0) ? x : ((D == 1) ? A : 0.0);
    m2 = (D == 0) ? fma(x, x, e2l + e2r) : ((D == 1) ? A * (x + (dhi - c)) : ((D == 2) ? A : 0.0));
}

RC_HD bool moments_ok(double r0, double r1, double r2, double scale) {
    const double s2 = 2.0 * scale;
    if (kSumRuleMoments < 3) return fabs(r0) <= kSumRuleTol;       // (r1, r2 are dead code then)
    return (fabs(r0) <= kSumRuleTol) && (fabs(r1) <= kSumRuleTol * s2) && (fabs(r2) <= kSumRuleTol * s2 * s2);
}

// What the right-hand sides need of the matrix, picked EARLY (right after the matrix is formed) so that the arrays
// themselves can die where they used to: d_lo, d_hi and the squared couplings left / right of site lo - only for D <= 1.
struct GuardSites {
    double dlo = 0.0, dhi = 0.0, e2l = 0.0, e2r = 0.0;
};
template <int N>
RC_HD GuardSites chain_guard_sites(const double (&d)[N], const double (&e0sq)[N], int lo, int hi) {
    GuardSites g;
    if (hi - lo <= 1) {                                 // wave-uniform
        g.dlo = pick_site<N>(d, lo);
        g.dhi = pick_site<N>(d, hi);
        if (hi == lo) {
            g.e2l = lo > 0 ? pick_site<N>(e0sq, lo - 1) : 0.0;
            g.e2r = lo < N - 1 ? pick_site<N>(e0sq, lo) : 0.0;
        }
    }
    return g;
}

// The guard for a real tridiagonal chain: weights w and eigenvalues lam against the matrix (D = hi - lo hops, pe = product of
// the couplings between the two sites, `gs` = chain_guard_sites of the ORIGINAL matrix).  ENDS = true: {lo, hi} = {0, N-1} at
// compile time and gs is not read (N = 2: the original diagonal is gone on the all-fp64 route, d_0 + d_1 = lam_0 + lam_1
// stands in).  scale <= 0: taken from the spread of the eigenvalues.  (The differences lam_k - lam_0 are the ones the phase
// loop forms.)
template <int N, bool ENDS>
RC_HD bool chain_sum_rules_ok(const double (&w)[N], const double (&lam)[N], int D, const GuardSites& gs, double pe, double scale) {
    double s0 = w[0], s1 = 0.0, s2 = 0.0, spread = 0.0;
#pragma unroll
    for (int k = 1; k < N; ++k) {
        const double dl = lam[k] - lam[0];
        const double t = w[k] * dl;
        s0 += w[k];
        s1 += t;
        s2 = fma(t, dl, s2);
        spread = fmax(spread, fabs(dl));               // (dead code when the caller passes a scale)
    }
    double m0, m1, m2;
    const double c = lam[0];
    if (ENDS) chain_moment_rhs(N - 1, N == 2 ? lam[1] : 0.0, N == 2 ? lam[0] : 0.0, 0.0, 0.0, pe, c, m0, m1, m2);
    else chain_moment_rhs(D, gs.dlo, gs.dhi, gs.e2l, gs.e2r, pe, c, m0, m1, m2);
#if defined(RC_GUARD_DEBUG) && !defined(__HIP_DEVICE_COMPILE__)
    printf("guard: r0 %.3e r1 %.3e r2 %.3e scale %.3e\n", s0 - m0, s1 - m1, s2 - m2, scale > 0.0 ? scale : fmax(1.0, 0.5 * spread));
#endif
    return moments_ok(s0 - m0, s1 - m1, s2 - m2, scale > 0.0 ? scale : fmax(1.0, 0.5 * spread));
}

// How the eigenvector weights w_k = Q[in,k] Q[out,k] are obtained on the fast path.
enum WeightMode {
    kWeightsRows = 0,     // rows `in`, `out` of Q accumulated through the QL sweeps (most registers: 4N doubles of state)
    kWeightsAdjugate = 1, // QL on eigenvalues only, weights from the adjugate formula (any in/out; keeps d0, e0^2)
    kWeightsEnds = 2      // same, specialised to end-to-end transfer {in,out} = {0, N-1}: w_k = prod(e0) / chi'(lam_k),
                          // nothing of the original matrix survives the QL phase except one scalar (2N doubles of state)
};

// Fidelity of one sample - fast path.  loadg(j) returns this sample's j-th draw, laid out (g0_i, g1_i, g2_i),
// i = 0..N-1.  x: controller (N biases, then T); sctab: the sin/cos table (RC_SINCOS_TABLE_VALUES).  Returns false -
// per sample - when this sample needs the general path.
// Eigenvalues: the eigenvalue-only modes at 3 <= N <= RC_MIXED_MAX_N take the mixed-precision route (fp32 QL -> fp64 Halley
// step -> stepping path -> all-fp64 QL for the tile, in that order of escalation; see the block comment at kMixedEig);
// the rows mode and larger N run the all-fp64 QL (tridiag_ql2_fast).
// `stamp` is used by diagnostic builds only (-DRC_STAMPS); `extra_steps` (optional) is set to 1 when the tile left the
// one-step path of the mixed-precision route (rc_stats_polish_tiles).
template <int N, int MODE, typename LoadG>
RC_HD bool chain_fidelity_fast(const double* x, const double* h0d, const double* h0o, LoadG loadg,
                               int in, int out, const double* sctab, double& fid, long long* stamp = nullptr,
                               int* extra_steps = nullptr) {
    constexpr bool VEC = (MODE == kWeightsRows);
    constexpr bool MIXED = kMixedEig && !VEC && N >= 3 && N <= RC_MIXED_MAX_N;
    TriEig<N, VEC ? 2 : 0> s;
    double d0[N], e0sq[N], w[N];               // kWeightsAdjugate / mixed path: original diagonal / squared couplings
    float df[N], ef[N];                        // mixed path: the fp32 copy the QL rotations work on
    double pe_all = 1.0;                       // product of the couplings between the two sites (all of them: kWeightsEnds)
    const int lo = in < out ? in : out, hi = in < out ? out : in;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        s.d[i] = x[i] + h0d[i] + loadg(3 * i);
        if (VEC) {
            s.z[0][i] = (i == in) ? 1.0 : 0.0;
            s.z[VEC ? 1 : 0][i] = (i == out) ? 1.0 : 0.0;
        }
        if (MIXED) df[i] = (float)s.d[i];
    }
#pragma unroll
    for (int i = 1; i < N; ++i) {
        const double re = h0o[i - 1] + loadg(3 * i + 1);
        const double im = loadg(3 * i + 2);
        // the 1e-300 rides in the inner fma for free and keeps the seed finite: a coupling that cancels exactly
        // becomes 1e-150 instead of 0 (what the first sweep would leave there anyway), no compare / select
        const double h = fma(re, re, fma(im, im, 1e-300));
        if (MIXED) {
            // only the SQUARED couplings are needed in fp64 (Halley recurrences, adjugate recurrences); the product of
            // the couplings between the two sites is one square root of the product of the squares
            e0sq[i - 1] = h;
            const float hf = (float)h + 1e-30f;             // an exactly cancelled coupling enters the fp32 QL as 1e-15
            ef[i - 1] = hf * seed_rsqf(hf);
            if (MODE == kWeightsEnds || (i - 1 >= lo && i - 1 < hi)) pe_all *= h;
            continue;
        }
        double r, rinv;
        sqrt_rsqrt(h, r, rinv);
        s.e[i - 1] = r;
        if (MODE == kWeightsEnds) pe_all *= s.e[i - 1];
        if (MODE == kWeightsAdjugate) {
            e0sq[i - 1] = h;
            if (i - 1 >= lo && i - 1 < hi) pe_all *= s.e[i - 1];     // wave-uniform condition
        }
    }
    s.e[N - 1] = 0.0;
    GuardSites gsites;                             // the guard's view of the original matrix (general adjugate mode, D <= 1 only)
    if (kSumRuleGuard && kSumRuleMoments >= 3 && MODE == kWeightsAdjugate) {
        e0sq[N - 1] = 0.0;
        gsites = chain_guard_sites<N>(s.d, e0sq, lo, hi);
    }
#if defined(RC_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    if (stamp) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[0]) : "v"(s.e[0]), "v"(s.d[0]) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    bool ok;
    float scale32 = 0.0f;                          // max(|d|, |e|) of the matrix (mixed path; the guard's scale)
    if (MIXED) {
        ef[N - 1] = 0.0f;
        e0sq[N - 1] = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) d0[i] = s.d[i];
        {   // pe = sqrt(prod e^2); the nudge keeps the seed finite when the product underflows (several cut bonds)
            double r, rinv;
            sqrt_rsqrt(pe_all + 1e-300, r, rinv);
            pe_all = r;
        }
        // a lane that hit the fp32 sweep cap starts the refinement from garbage: it is simply one more lane that needs the
        // stepping path (converged + distinct roots are the eigenvalues whatever the start was)
        const bool ok32 = tridiag_ql_f32<N>(df, ef, scale32);
        ok = true;
        const ChainChi<N> chi{d0, e0sq};
        bool need = !mixed_refine<N>(chi, df, scale32, ok32, s.d, extra_steps);
#if defined(RC_EXPERIMENT_FALLBACK_NOT_RUN) && defined(__HIP_DEVICE_COMPILE__)
        {   // TIMING EXPERIMENT ONLY: the tile-wide all-fp64 QL compiled in but never executed (results of such tiles are wrong)
            int never = 1;
            asm volatile("" : "+s"(never));
            if (never) need = false;
        }
#endif
        if (RC_UNLIKELY(vote_any(need))) {
            // still not settled somewhere in the tile (a pair closer than ~5e-5: beyond what a polynomial iteration
            // from an fp32 start separates): the whole tile takes the all-fp64 QL from the original matrix, wave-wide
            // (~2x the cost of this tile), with the TIGHT split tolerance (the 1e-10 of the eigenvalue-only fast path
            // assumes e_l^2 / gap is negligible: not for a close pair).  Its eigenvalues carry ~N eps scale of error each,
            // and the product-formula weights stay accurate down to gaps of that size (the two weights of a pair are
            // +-A / gap with the SAME computed gap: their joint contribution is a divided difference of a smooth function)
            // - for the end-to-end weights; see kDegenerateGapEnds / kDegenerateGapAdjugate for the general adjugate mode
            // (round 4) A lane that WAS settled keeps its polished eigenvalues: a QL eigenvalue carries ~N eps scale of
            // error where a polished root of chi carries a few ulp, and at |d| ~ 100, T ~ 100 that difference is a phase error
            // of 1e-11 - the fuzz campaign's worst chain cases (1.4e-11, seeds 4020 / 4092 / 4117) were healthy samples whose
            // TILE took this route because of a neighbour.  Only the lanes that needed the route use its eigenvalues.
            double keep[N];
#pragma unroll
            for (int i = 0; i < N; ++i) keep[i] = s.d[i];
#pragma unroll
            for (int i = 0; i < N; ++i) s.d[i] = d0[i];
#pragma unroll
            for (int i = 0; i < N - 1; ++i) {
                double r, rinv;
                sqrt_rsqrt(e0sq[i], r, rinv);
                s.e[i] = r;
            }
            s.e[N - 1] = 0.0;
            const bool okql = tridiag_ql2_fast(s, kEps);
            double mingap = 1e300, scale = 1.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                scale = fmax(scale, fabs(s.d[k]));
#pragma unroll
                for (int m = k + 1; m < N; ++m) mingap = fmin(mingap, fabs(s.d[k] - s.d[m]));
            }
            const bool need2 = !(mingap > (MODE == kWeightsEnds ? kDegenerateGapEnds : kDegenerateGapAdjugate) * scale);
            // (-DRC_KEEP_SETTLED=0: the behaviour before the fix - every lane takes the QL's eigenvalues; the lock-step host test
            // builds both to show that it sees the difference)
#pragma unroll
            for (int i = 0; i < N; ++i) s.d[i] = (need || !RC_KEEP_SETTLED) ? s.d[i] : keep[i];
            ok = (!need && RC_KEEP_SETTLED) || okql;
            need = (need || !RC_KEEP_SETTLED) && need2;
        }
        const bool wok = (MODE == kWeightsAdjugate) ? adjugate_weights<N, false>(d0, e0sq, s.d, lo, hi, pe_all, w)
                                                    : ends_weights<N, false>(pe_all, s.d, w);
        ok = ok && wok && !need;
    } else {
        if (MODE == kWeightsAdjugate) {
#pragma unroll
            for (int i = 0; i < N; ++i) d0[i] = s.d[i];
            e0sq[N - 1] = 0.0;
        }
        ok = tridiag_ql2_fast(s);                   // per lane; a bad lane just keeps computing garbage
        // (round 4) N >= 14, general adjugate mode: ONE Halley step on chi of the original matrix takes the QL's eigenvalues
        // (error ~N eps scale: at |d| ~ 100, |T| ~ 100, N = 16 a phase error of 3e-11 - the fuzz campaign's worst chain
        // cases once the mixed path's were gone) to a few ulp.  d0 / e0sq are alive in this mode anyway (the weights'
        // recurrences); the end-to-end mode at these sizes has no register for them and keeps the QL's values.  A lane
        // whose step is not tiny (a pair the gap test below rejects anyway, or a non-converged QL) keeps what it had.
        if (kPolishLargeN && MODE == kWeightsAdjugate && N > RC_MIXED_MAX_N) {
            double lam[N];
#pragma unroll
            for (int i = 0; i < N; ++i) lam[i] = s.d[i];
            double crit;
            const ChainChi<N> chi{d0, e0sq};
            const double maxd = halley_polish<N>(chi, lam, crit);
            double scale = 1.0;
#pragma unroll
            for (int i = 0; i < N; ++i) scale = fmax(scale, fabs(s.d[i]));
            const bool take = (maxd <= 1e-9 * scale) && (crit <= kHalleyCritical);
#pragma unroll
            for (int i = 0; i < N; ++i) s.d[i] = take ? lam[i] : s.d[i];
        }
    }
#if defined(RC_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    if (stamp) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[1]) : "v"(s.e[0]), "v"(s.d[0]) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (MIXED) {
        // weights done above
    } else if (MODE == kWeightsRows) {
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] = s.z[VEC ? 1 : 0][k] * s.z[0][k];
    } else if (MODE == kWeightsAdjugate) {
        ok = adjugate_weights<N>(d0, e0sq, s.d, lo, hi, pe_all, w) && ok;
    } else {
        ok = ends_weights<N>(pe_all, s.d, w) && ok;
    }
    // |sum_k w_k exp(-i T lam_k)|^2 is unchanged by the global phase exp(i T lam_0): N-1 sincos instead of N
    const double T = fabs(x[N]);
    const double Tk = T * kTurnsPerRadian;
    double re = w[0], im = 0.0;
#pragma unroll
    for (int k = 1; k < N; ++k) {
        double sk, ck;
        if (kTableSinCos) sincos_table(Tk * (s.d[k] - s.d[0]), sctab, sk, ck);
        else sincos_reduced(T * (s.d[k] - s.d[0]), sk, ck);
        re = fma(w[k], ck, re);
        im = fma(-w[k], sk, im);
    }
    fid = fma(re, re, im * im);
    if (MIXED) ok = ok && (fid <= 2.0);             // a NaN (zero Halley denominator) goes to the general path
    // (the rows mode carries genuine eigenvector rows: it IS the repair route and needs no guard)
    if (kSumRuleGuard && !VEC && (MODE != kWeightsEnds || kSumRuleEnds))
        ok = ok && chain_sum_rules_ok<N, MODE == kWeightsEnds>(w, s.d, hi - lo, gsites, pe_all, MIXED ? (double)scale32 : 0.0);
    return ok;
}

// Fidelity of one sample - general path (runtime n).  g: this sample's 3n draws; d/e/za/zb: work vectors of n
// entries each (see tridiag_ql2_general for `Vec`).
template <typename Vec>
RC_HD double chain_fidelity_general(int n, const double* x, const double* h0d, const double* h0o,
                                    const double* g, int in, int out, Vec d, Vec e, Vec za, Vec zb) {
    for (int i = 0; i < n; ++i) {
        d[i] = x[i] + h0d[i] + g[3 * i];
        za[i] = (i == in) ? 1.0 : 0.0;
        zb[i] = (i == out) ? 1.0 : 0.0;
        e[i] = 0.0;
    }
    for (int i = 1; i < n; ++i) {
        const double re = h0o[i - 1] + g[3 * i + 1];
        const double im = g[3 * i + 2];
        const double h = fma(re, re, im * im);
        double r, rinv;
        sqrt_rsqrt(h, r, rinv);
        e[i - 1] = (h > 0.0) ? r : 0.0;
    }
    tridiag_ql2_general(n, d, e, za, zb);
    const double T = fabs(x[n]);
    double re = 0.0, im = 0.0;
    for (int k = 0; k < n; ++k) {
        double sk, ck;
        sincos_reduced(T * d[k], sk, ck);
        const double w = zb[k] * za[k];
        re = fma(w, ck, re);
        im = fma(-w, sk, im);
    }
    return fma(re, re, im * im);
}

}  // namespace rc
