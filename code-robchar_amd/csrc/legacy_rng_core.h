// NumPy's LEGACY normal stream (the RNG of the reference's MC path: `np.random.normal`, noise_model.py:114-115, consumed
// at noise_model.py:137-146 and mcsim.py:425) restated so that it can be produced on the GPU:
//
//   MT19937 (Matsumoto & Nishimura 1998; numpy/random/src/mt19937/mt19937.c): 624-word state, the recurrence
//       x[i+624] = x[i+397] ^ (((x[i] & 0x80000000) | (x[i+1] & 0x7fffffff)) >> 1) ^ ((x[i+1] & 1) ? 0x9908b0df : 0),
//       output = tempered word;
//   legacy_double (mt19937.h: mt19937_next_double): two consecutive outputs a = w0 >> 5, b = w1 >> 6,
//       u = (a * 2^26 + b) / 2^53;
//   legacy_gauss (numpy/random/src/legacy/legacy-distributions.c): Marsaglia's polar method - attempts of two uniforms
//       (FOUR words) x1 = 2u1 - 1, x2 = 2u2 - 1, r2 = x1^2 + x2^2, rejected when r2 >= 1 or r2 == 0; an accepted attempt
//       yields TWO normals, f x2 first and f x1 second (the "cached" one), f = sqrt(-2 ln(r2) / r2).
//
// Everything up to the accept / reject decision is exact integer or exactly rounded fp64 arithmetic (no fused
// multiply-add: NumPy's C code has none), so the uint32 stream, the attempt boundaries and therefore the generator
// state after any number of draws are BIT-IDENTICAL to NumPy's.  ln() is the C library's: round 5 restates the `log` of the
// image's glibc operation for operation (log_glibc_fma below), so that the NORMALS are NumPy's bit for bit as well.
//
// Plain C++ header shared by the HIP kernels and the host unit test (tests/host/host_core.cpp, checked against NumPy).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "glibc_log_data.h"

#if defined(__HIPCC__)
#define RCL_HD __host__ __device__ __forceinline__
#else
#define RCL_HD inline
#endif

namespace rcl {

constexpr int kMtN = 624;
constexpr int kMtM = 397;
constexpr int kMtChunk = kMtN - kMtM;          // 227 consecutive words of the recurrence are mutually independent

// x[i+624] from x[i], x[i+1], x[i+397]
RCL_HD uint32_t mt_next_word(uint32_t xi, uint32_t xi1, uint32_t xim) {
    const uint32_t y = (xi & 0x80000000u) | (xi1 & 0x7fffffffu);
    return xim ^ (y >> 1) ^ ((xi1 & 1u) ? 0x9908b0dfu : 0u);
}

RCL_HD uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// exactly rounded fp64 product / sum that the compiler must NOT contract into an fma (hipcc contracts by default, and
// HIP's __dmul_rn / __dadd_rn are plain operators): the pragma removes the `contract` flag from the operation itself,
// so it survives inlining.  (g++ builds of the host unit test pass -ffp-contract=off.)
#if defined(__clang__)
#define RCL_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define RCL_NO_CONTRACT
#endif
RCL_HD double mul_rn(double a, double b) {
    RCL_NO_CONTRACT
    return a * b;
}
RCL_HD double add_rn(double a, double b) {
    RCL_NO_CONTRACT
    return a + b;
}

// 2 u - 1 for the uniform made of two RAW (untempered) state words: exact (u = k / 2^53, 2u - 1 = (k - 2^52) / 2^52)
RCL_HD double mt_symmetric_uniform(uint32_t raw0, uint32_t raw1) {
    const int32_t a = (int32_t)(mt_temper(raw0) >> 5), b = (int32_t)(mt_temper(raw1) >> 6);
    const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    return 2.0 * u - 1.0;
}

// ---- the C library's log(), operation for operation -------------------------------------------------------------------
// glibc >= 2.28: the table-driven double-precision log of ARM's optimized-routines (sysdeps/ieee754/dbl-64/e_log.c; constants:
// glibc_log_data.h, read from the installed libm by scripts/glibc_log_table.py), in the `__log_fma` build that the ifunc
// resolver selects on every x86-64 CPU with FMA and AVX2.  That build is compiled with contraction on, so WHICH multiply-adds
// are fused is part of the function: the sequence below is the published algorithm with the pairing of the shipped code
//   r = fma(z, invc, -1);  w = fma(kd, ln2hi, logc);  hi = w + r;  lo = fma(kd, ln2lo, (w - hi) + r);
//   y = fma(r r2, fma(fma(r, A4, A3), r2, fma(r, A2, A1)), fma(r2, A0, lo)) + hi
// and, for 1 - 2^-4 <= x < 1 + 0x1.09p-4 (the branch that keeps the relative error small near 1),
//   w = r 2^27 (fused into rhi = fma(-2^27, r, fma(r, 2^27, r)));  hi = fma(rhi^2, B0, r);  lo = fma(rhi^2, B0, r - hi);
//   y = hi + fma(P(r), r^3, fma(B0 rlo, rhi + r, lo)),   P = B1 + r B2 + r2 B3 + r3 (B4 + r B5 + r2 B6 + r3 (B7 + r B8 + r2 B9 + r3 B10))
// with every remaining product / sum rounded on its own (RCL_NO_CONTRACT: the device compiler contracts by default).  Arguments:
// finite, positive, normal (the polar method's r2 lies in [2^-104, 1)); `tab` = the 128 x (invc, logc) table
// (RC_GLIBC_LOG_TAB_VALUES; LDS or constant memory on the device).  tests/test_host_core.py checks it against log() itself on
// 10^7 arguments; rc_legacy_log_is_host_exact() repeats a short form of that check on the host the library runs on.
#if defined(__clang__)
#define RCL_NO_CONTRACT_FN _Pragma("clang fp contract(off)")
#else
#define RCL_NO_CONTRACT_FN
#endif
RCL_HD double log_fma_op(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return fma(a, b, c);
#endif
}
RCL_HD double log_glibc_fma(double x, const double* tab) {
    RCL_NO_CONTRACT_FN
    const double A[5] = RC_GLIBC_LOG_POLY_A;
    const double B[11] = RC_GLIBC_LOG_POLY_B;
    uint64_t ix;
    memcpy(&ix, &x, 8);
    if (ix - 0x3fee000000000000ull < 0x308ffffffffffull + 1ull) {          // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        const double p12 = log_fma_op(r2, B[3], log_fma_op(r, B[2], B[1]));
        const double p45 = log_fma_op(r2, B[6], log_fma_op(r, B[5], B[4]));
        double p78 = log_fma_op(r2, B[9], log_fma_op(r, B[8], B[7]));
        p78 = log_fma_op(r3, B[10], p78);
        const double P = log_fma_op(log_fma_op(p78, r3, p45), r3, p12);
        const double rhi = log_fma_op(-0x1p27, r, log_fma_op(r, 0x1p27, r));
        const double rlo = r - rhi;
        const double rhi2 = rhi * rhi;
        const double hi = log_fma_op(rhi2, B[0], r);
        double lo = log_fma_op(rhi2, B[0], r - hi);
        lo = log_fma_op(B[0] * rlo, r + rhi, lo);
        return hi + log_fma_op(P, r3, lo);
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    double z;
    memcpy(&z, &iz, 8);
    const double invc = tab[2 * i], logc = tab[2 * i + 1];
    const double kd = (double)k;
    const double r = log_fma_op(z, invc, -1.0);
    const double w = log_fma_op(kd, RC_GLIBC_LN2HI, logc);
    const double hi = w + r;
    const double lo = log_fma_op(kd, RC_GLIBC_LN2LO, (w - hi) + r);
    const double r2 = r * r;
    const double q = log_fma_op(log_fma_op(r, A[4], A[3]), r2, log_fma_op(r, A[2], A[1]));
    return log_fma_op(r * r2, q, log_fma_op(r2, A[0], lo)) + hi;
}

// One attempt of the polar method on four consecutive RAW words; true when NumPy accepts it.
RCL_HD bool polar_attempt(uint32_t r0, uint32_t r1, uint32_t r2w, uint32_t r3, double& x1, double& x2, double& r2) {
    x1 = mt_symmetric_uniform(r0, r1);
    x2 = mt_symmetric_uniform(r2w, r3);
    r2 = add_rn(mul_rn(x1, x1), mul_rn(x2, x2));
    return !(r2 >= 1.0 || r2 == 0.0);
}

// ---- `directional_perturbation`'s consumption pattern (noise_model.py:183-189): np.random.randint(0, ndir), then two
// legacy normals = ONE accepted polar attempt.  rng = ndir - 1, mask = the smallest 2^k - 1 >= rng (RandomState.randint:
// _bounded_integers with use_masked, 32-bit outputs).
constexpr int kDirMaxLen = 250;                   // longest sample followed (probability of more: < 1e-30)

// tempered word & mask <= rng ?
RCL_HD bool dir_int_accept(uint32_t raw_word, uint32_t mask, uint32_t rng, uint32_t& v) {
    v = mt_temper(raw_word) & mask;
    return v <= rng;
}

// Number of raw words a sample STARTING at raw[p] consumes (1 .. kDirMaxLen); 0 = it runs off the buffer of W words;
// 255 = longer than kDirMaxLen.  rng == 0 (one direction): randint consumes nothing.
RCL_HD unsigned char dir_sample_len(const uint32_t* raw, long long p, long long W, uint32_t rng, uint32_t mask) {
    long long q = p;
    if (rng != 0) {
        for (;;) {
            if (q >= W) return 0;
            uint32_t v;
            const bool acc = dir_int_accept(raw[q++], mask, rng, v);
            if (acc) break;
            if (q - p > kDirMaxLen) return 255;
        }
    }
    for (;;) {
        if (q + 4 > W) return 0;
        double x1, x2, r2;
        const bool acc = polar_attempt(raw[q], raw[q + 1], raw[q + 2], raw[q + 3], x1, x2, r2);
        q += 4;
        if (acc) return (unsigned char)(q - p);
        if (q - p > kDirMaxLen) return 255;
    }
}

// words consumed by kDirGroup CONSECUTIVE samples starting at position p of the per-position length array (0: one of them
// is invalid or runs off the array): the sequential walk over the stream then costs one dependent load per GROUP
constexpr int kDirGroup = 8;
RCL_HD unsigned short dir_group_len(const unsigned char* len, long long p, long long npos) {
    long long q = p;
    for (int j = 0; j < kDirGroup; ++j) {
        if (q >= npos) return 0;
        const unsigned char l = len[q];
        if (l == 0 || l == 255) return 0;
        q += l;
    }
    return (unsigned short)(q - p);
}

// Where element `e` of the normal stream goes.  The stream is cut into `n_periods` periods of `period` elements; the
// first `skip` elements of every period are consumed but not stored (the burned draw of `rng(scale=sigma)`,
// mcsim.py:425 / gen_fig_8_arim_fcall_scaling.py:124); the rest of period p lands contiguously at
// out[p * (period - skip) ...], scaled by scales[p].  Returns -1 for a dropped element.
RCL_HD long long stream_slot(long long e, long long period, long long skip, long long* p_out) {
    const long long p = e / period;
    const long long o = e - p * period;
    *p_out = p;
    return (o < skip) ? -1 : p * (period - skip) + (o - skip);
}

}  // namespace rcl
