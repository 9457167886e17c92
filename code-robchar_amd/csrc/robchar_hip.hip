// librobchar_hip.so - HIP kernels (gfx950 / MI355X) and the C ABI declared in include/robchar_hip.h.
//
// Kernels
//   mc_fid_chain_kernel<N,M> one (controller, perturbation) sample per LANE, one wave per workgroup.  A wave
//                            owns tiles of 64 consecutive samples of ONE controller: the controller row is
//                            wave-uniform (scalar loads); a tile's 64*3N draws are one contiguous HBM run that
//                            is copied to LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, every HBM
//                            byte fetched exactly once, fully coalesced) and read back transposed (lane l
//                            takes its own 3N values).  Per lane: real symmetric tridiagonal implicit QL in
//                            registers with wave-uniform control flow (tridiag_core.h).
//   mc_fid_chain_anyn_kernel chains of 16 < N <= 32 spins: the general per-sample routine, work vectors in dynamic LDS.
//   mc_fid_jacobi_kernel     general complex Hermitian path (ring topology, cross-check): 8 or 4 samples per WAVE,
//                            dense matrix in LDS, round-robin cyclic Jacobi with the rotations of a round
//                            spread over the 64 lanes.
//   mc_fid_expm_kernel       dense complex, possibly non-Hermitian H (directional_perturbation): one WAVE per sample,
//                            Pade scaling-and-squaring expm in LDS - the reference's own algorithm shape on the device.
//   reduce_kernel            one workgroup per controller: RIM_1, std, min, Q(thr) for the centre / DKW-upper /
//                            DKW-lower variants in two passes over the K fidelities (fixed summation order).
//   sort_*_kernel            row sort for the ECDF: merge sort in LDS (K <= 16384), chunked bitonic network above.
//   philox_normal_kernel     counter-based Gaussian draws for sample spaces too large to draw on the host.
//
// Roofline: algorithmic HBM traffic is 24 N + 8 bytes per sample; the kernel is bound by
// fp64 VALU issue under the socket power cap, not by HBM.  See DESIGN.md.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/robchar_hip.h"
#include "tridiag_core.h"
#include "sort_core.h"
#include "legacy_rng_core.h"
#include "mt19937_jump_poly.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define RC_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(RC_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

struct StaticH {          // passed by value in the kernarg segment: no device allocation for 2N doubles
    double diag[RC_MAX_NSPIN];
    double off[RC_MAX_NSPIN];
};

struct FidParams {
    const double* ctrl;    // [C][N+1]
    const double* draws;   // [C][K][N][3]
    double* fid;           // [C][K]
    long long C, K;
    long long draw_cstride;     // elements between consecutive controllers' draw blocks (K*3N; 0 = shared set)
    long long tiles_per_ctrl;   // ceil(K / 64)
    long long ntiles;           // C * tiles_per_ctrl
    int in, out;
    int align16;                // draws base and every controller's run of K*3N doubles are 16-byte aligned
    StaticH h0;
    long long* stamps;          // diagnostic builds only (-DRC_STAMPS): [ntiles][8] s_memtime stamps
};

typedef __attribute__((address_space(1))) const void* rc_gptr_t;
typedef __attribute__((address_space(3))) void* rc_lptr_t;

// Staging geometry.  A wave's 64-sample tile is brought in through LDS in `fid_phases(N)` phases of
// 64/phases samples each, so that the per-wave LDS buffer (samples-per-phase * 3N doubles: 5.4 KiB at N = 7)
// never limits residency below what the registers allow (72 VGPRs at N = 7 -> 7 waves per SIMD).
// Weight mode and residency.  Measured on MI355X (kbench, 1e6 evaluations, N = 7): eigenvector rows 98 us, general
// adjugate 83 us (4 waves/SIMD: it keeps the original matrix through the QL phase), end-to-end adjugate 78 us at
// 5 waves/SIMD with 2 staging phases (more waves or phases change nothing: the kernel is bound by VALU instruction
// count at the clock the chip holds, not by latency).  The adjugate modes win at every N (2..16), so AUTO = adjugate
// (its end-to-end specialisation when {in,out} = {0,N-1}); the rows mode stays selectable as a cross-check.
#ifndef RC_WAVES_SMALL
#define RC_WAVES_SMALL 5
#endif
// Chosen from the ISA's VGPR need per instantiation (`make asm`; tests/test_asm_resources.py fails on any spill): a
// wave limit of W allows floor(512 / W) VGPRs (multiples of 8).  Residency above ~4 waves buys nothing,
// a spilled register costs scratch traffic in the innermost loop.  -DRC_WAVES_N=<n> -DRC_WAVES_W=<w> overrides one N
// (all modes) for A/B timing.
constexpr int fid_min_waves(int n, int mode) {
#if defined(RC_WAVES_N) && defined(RC_WAVES_W)
    if (n == RC_WAVES_N) return RC_WAVES_W;
#endif
    if (mode == rc::kWeightsAdjugate) return n <= 6 ? RC_WAVES_SMALL : (n <= 8 ? 4 : (n <= 12 ? 3 : 2));
    if (mode == rc::kWeightsRows) return n <= 8 ? RC_WAVES_SMALL : (n <= 12 ? 3 : 2);
    // kWeightsEnds
    return n <= 7 ? RC_WAVES_SMALL : (n <= 9 ? 4 : (n <= 11 ? 3 : (n <= 14 ? 2 : 1)));
}
// staging phases: the LDS buffer (64/phases * 3N doubles per wave) must not cap residency below the register limit
constexpr int fid_phases(int n, int mode) { return n <= 2 ? 1 : (n <= 8 ? 2 : 4); }

// (cos, sin)(2 pi k / 64), k = 0..63: source of the per-wave LDS copy that sincos_table reads
__device__ const double g_sincos_table[128] = {RC_SINCOS_TABLE_VALUES};

// Tiles with at least one sample that left the fast path (sweep cap / degenerate pair) since the last reset: a
// diagnostic counter, touched only on that rare path (rc_stats_general_tiles).
__device__ unsigned long long g_general_tiles = 0;

// Lane-strided view of an LDS work area: element i of this lane's vector lives at base[i * stride].
struct LdsVec {
    double* base;
    int stride;
    __device__ __forceinline__ double& operator[](int i) const { return base[i * stride]; }
};

// ------------------------------------------------------------------------------------------------
// fidelity kernel: chain topology, lane per sample, one wave per workgroup, one tile per wave
// ------------------------------------------------------------------------------------------------
template <int N, int MODE>
__global__ __launch_bounds__(64, fid_min_waves(N, MODE)) void mc_fid_chain_kernel(const FidParams p) {
    constexpr int G = 3 * N;                       // doubles per sample
    constexpr int PH = fid_phases(N, MODE);
    constexpr int SP = 64 / PH;                    // samples per staging phase
    constexpr int kPhaseBytes = SP * G * 8;
    __shared__ __attribute__((aligned(16))) double stage[SP * G];
    __shared__ __attribute__((aligned(16))) double sctab[128];   // sincos_table's table, one copy per wave (1 KiB)

    const int lane = threadIdx.x;
    const long long tile = blockIdx.x;             // wave-uniform
    if (rc::kTableSinCos) {                        // lane k copies entry k; consumed long after the staging waits
        const double2 ent = reinterpret_cast<const double2*>(g_sincos_table)[lane];
        reinterpret_cast<double2*>(sctab)[lane] = ent;
    }
    // The staging phase is a handful of instructions separated by memory latency; issued at raised priority it
    // is not starved by the older waves of the SIMD that are in their (VALU-dense) compute phase, so its
    // latency overlaps their arithmetic instead of stretching (measured: staging 31k -> 4k ticks per tile).
    __builtin_amdgcn_s_setprio(3);
#ifdef RC_STAMPS
    const long long t_begin = __builtin_amdgcn_s_memtime();
    const long long r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    const long long c = tile / p.tiles_per_ctrl;
    const long long kb = (tile - c * p.tiles_per_ctrl) * 64;
    const int nk = (int)((p.K - kb < 64) ? (p.K - kb) : 64);

    // controller row: wave-uniform -> scalar registers
    const double* xg = p.ctrl + c * (N + 1);
    double x[N + 1];
    bool pad = false;
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        x[i] = xg[i];
        pad |= (x[i] != x[i]);
    }
#ifdef RC_STAMPS
    long long t_ph[4] = {0, 0, 0, 0};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t_ctrl = __builtin_amdgcn_s_memtime();
#endif
    double* dst = p.fid + c * p.K + kb;
    if (pad) {                                     // NaN-padded controller (mcsim.py:442-443): no draws read
        if (lane < nk) dst[lane] = __builtin_nan("");
        return;
    }

    // HBM -> LDS -> registers.  The tile's draws are one contiguous run of nk*G doubles.  Each phase copies
    // SP samples into LDS by LDS-DMA (global_load_lds: no staging VGPRs, fully coalesced, every HBM byte
    // fetched once; 16-byte pieces when the run is 16-byte aligned and sized, 4-byte pieces otherwise) and
    // the SP lanes that own them read their G values back (the transposition).
    const char* src = (const char*)(p.draws + c * p.draw_cstride + kb * G);
    double gl[G];
#pragma unroll
    for (int i = 0; i < G; ++i) gl[i] = 0.0;
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
        const int first = ph * SP;
        if (first < nk) {                          // wave-uniform
            const int cnt = (nk - first < SP) ? (nk - first) : SP;
            const int bytes = cnt * G * 8;
            const char* ps = src + (long long)first * G * 8;
            if (p.align16 && !(cnt & 1)) {
#pragma unroll
                for (int it = 0; it < (kPhaseBytes + 1023) / 1024; ++it) {
                    const int off = it * 1024 + lane * 16;
                    if (off < bytes)
                        __builtin_amdgcn_global_load_lds((rc_gptr_t)(ps + off),
                                                         (rc_lptr_t)((char*)stage + it * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll 2
                for (int it = 0; it < (kPhaseBytes + 255) / 256; ++it) {
                    const int off = it * 256 + lane * 4;
                    if (off < bytes)
                        __builtin_amdgcn_global_load_lds((rc_gptr_t)(ps + off),
                                                         (rc_lptr_t)((char*)stage + it * 256), 4, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // DMA landed
#ifdef RC_STAMPS
            if (ph < 2) t_ph[2 * ph] = __builtin_amdgcn_s_memtime();
#endif
            const int rel = lane - first;
            if (rel >= 0 && rel < cnt) {
#pragma unroll
                for (int i = 0; i < G; ++i) gl[i] = stage[rel * G + i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the buffer is refilled
#ifdef RC_STAMPS
            if (ph < 2) t_ph[2 * ph + 1] = __builtin_amdgcn_s_memtime();
#endif
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if (rc::kTableSinCos) __syncthreads();         // the table copy has landed (one wave per workgroup: no wait)
#ifdef RC_STAMPS
    const long long t_loaded = __builtin_amdgcn_s_memtime();
#endif

    double f = 0.0;
    bool ok = true;
#ifdef RC_STAMPS
    long long t_in[2] = {0, 0};
    if (lane < nk)
        ok = rc::chain_fidelity_fast<N, MODE>(x, p.h0.diag, p.h0.off, [&gl](int i) { return gl[i]; }, p.in, p.out, sctab, f, t_in);
#else
    if (lane < nk)
        ok = rc::chain_fidelity_fast<N, MODE>(x, p.h0.diag, p.h0.off, [&gl](int i) { return gl[i]; }, p.in, p.out, sctab, f);
#endif
    const unsigned long long badmask = __ballot(lane < nk && !ok);
    if (badmask) {
        if (lane == 0) atomicAdd(&g_general_tiles, 1ull);
        // Rare: some samples of this tile hit the sweep cap or a degenerate pair.  Recompute THOSE samples with the
        // general per-sample routine, CH lanes at a time, with the work vectors (4N doubles per sample) in the LDS
        // staging buffer, which is free now; each such lane re-reads its draws straight from HBM.
        constexpr int CH = (SP * G) / (4 * N);
        const bool bad = (badmask >> lane) & 1ull;
        const int rank = __popcll(badmask & ((1ull << lane) - 1ull));     // position among the bad lanes
        const int nbad = __popcll(badmask);
#pragma unroll 1
        for (int c0 = 0; c0 < nbad; c0 += CH) {
            const int rel = rank - c0;
            if (bad && rel >= 0 && rel < CH) {
                const LdsVec vd{stage + rel, CH}, ve{stage + N * CH + rel, CH}, va{stage + 2 * N * CH + rel, CH},
                    vb{stage + 3 * N * CH + rel, CH};
                f = rc::chain_fidelity_general(N, xg, p.h0.diag, p.h0.off,
                                               (const double*)src + (long long)lane * G, p.in, p.out, vd, ve, va, vb);
            }
        }
    }
    if (lane < nk) dst[lane] = f;

#ifdef RC_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    const long long t_end = __builtin_amdgcn_s_memtime();
    if (lane == 0 && p.stamps) {
        p.stamps[blockIdx.x * 8 + 0] = t_begin;
        p.stamps[blockIdx.x * 8 + 1] = t_loaded;
        p.stamps[blockIdx.x * 8 + 2] = t_end;
        p.stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime() - r_begin;
        p.stamps[blockIdx.x * 8 + 4] = t_ctrl;
        p.stamps[blockIdx.x * 8 + 5] = t_ph[0];
        p.stamps[blockIdx.x * 8 + 6] = t_in[0];     // QL starts
        p.stamps[blockIdx.x * 8 + 7] = t_in[1];     // QL done
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// fidelity kernel for long chains (RC_MAX_NSPIN_FAST < N <= RC_MAX_NSPIN): the general per-sample routine for every
// sample, runtime N, the four work vectors of a lane in dynamic LDS (4 N doubles per lane, lane-strided), draws
// read straight from HBM.  Same tiling (one wave per 64 samples of one controller) and the same arithmetic as the
// general path of mc_fid_chain_kernel; two orders of magnitude slower than the register-resident kernels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mc_fid_chain_anyn_kernel(const FidParams p, int n) {
    extern __shared__ __attribute__((aligned(16))) double anyn_work[];
    const int lane = threadIdx.x;
    const long long tile = blockIdx.x;
    const long long c = tile / p.tiles_per_ctrl;
    const long long kb = (tile - c * p.tiles_per_ctrl) * 64;
    const int nk = (int)((p.K - kb < 64) ? (p.K - kb) : 64);
    const double* xg = p.ctrl + c * (n + 1);
    bool pad = false;
    for (int i = 0; i <= n; ++i) pad |= (xg[i] != xg[i]);
    double* dst = p.fid + c * p.K + kb;
    if (lane >= nk) return;
    if (pad) {                                     // NaN-padded controller (mcsim.py:442-443): no draws read
        dst[lane] = __builtin_nan("");
        return;
    }
    const double* g = p.draws + c * p.draw_cstride + (kb + lane) * 3 * n;
    const LdsVec vd{anyn_work + lane, 64}, ve{anyn_work + n * 64 + lane, 64}, va{anyn_work + 2 * n * 64 + lane, 64},
        vb{anyn_work + 3 * n * 64 + lane, 64};
    dst[lane] = rc::chain_fidelity_general(n, xg, p.h0.diag, p.h0.off, g, p.in, p.out, vd, ve, va, vb);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;                              // valid in lane 0
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}

// ------------------------------------------------------------------------------------------------
// fidelity kernel: general complex Hermitian (chain or ring), one WAVE per sample, cyclic Jacobi in LDS
// ------------------------------------------------------------------------------------------------
// The dense N x N complex128 Hamiltonian of a sample lives in LDS (re/im planes); the 64 lanes of the wave share
// the work of each Jacobi round: a round applies the N/2 disjoint plane rotations of a round-robin ordering,
// rotation parameters by lanes k < N/2, then the row update (J^H A), the column update (A J) and the update of
// the two needed eigenvector rows, each spread over the lanes.  LDS operations of one wave execute in order, so
// the phases are separated by wave-level fences only (no s_barrier).  Handles the ring topology
// (noise_model.py:83-85), where the tridiagonal gauge trick of the chain kernel does not apply, and serves as an
// independent on-device cross-check of the chain kernel.
constexpr int kJacWaves = 4;             // waves (= samples in flight) per workgroup
constexpr int kJacMaxSweeps = 20;

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct JacParams {
    const double* ctrl;
    const double* draws;
    double* fid;
    long long C, K;
    long long draw_cstride;
    int N, in, out, ring;
    StaticH h0;
};

// SUB lanes cooperate on one sample, 64 / SUB samples per wave (SUB = 8 for N <= 8, 16 for N <= 16): the phases of a
// Jacobi round are latency-bound (LDS round trips and fences), so sharing them among several samples multiplies the
// throughput.  All samples of a wave sweep in lock-step until every one of them has converged (further rotations of
// a converged matrix are identities).
template <int SUB, int NM>
__global__ __launch_bounds__(64 * kJacWaves) void mc_fid_jacobi_kernel(const JacParams p) {
    constexpr int SPW = 64 / SUB;                                // samples per wave
    constexpr int SLOTS = kJacWaves * SPW;
    __shared__ double sAr[SLOTS][NM * NM], sAi[SLOTS][NM * NM];
    __shared__ double sPar[SLOTS][3 * (NM / 2)];                 // (c, s_re, s_im) per pair of the round
    __shared__ double sV[SLOTS][4 * NM];                         // rows `in`, `out` of V: re/im
    const int N = p.N;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = lane / SUB, sl = lane % SUB;                   // sample slot in the wave, lane within the sample
    const int slot = wave * SPW + g;
    double* Ar = sAr[slot];
    double* Ai = sAi[slot];
    double* par = sPar[slot];
    double* vir = sV[slot];
    double* vii = vir + NM;
    double* vor = vir + 2 * NM;
    double* voi = vir + 3 * NM;
    auto sub_sum = [](double v) {
#pragma unroll
        for (int off = SUB / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;                                                // every lane of the sub-group holds the sum
    };
    const int npl = N + (N & 1);          // players of the round-robin (a dummy when N is odd)
    const int m = npl - 1;                // rounds per sweep
    const int npair = npl / 2;
    const long long total = p.C * p.K;
    const long long stride = (long long)gridDim.x * kJacWaves * SPW;

    for (long long s0 = ((long long)blockIdx.x * kJacWaves + wave) * SPW; s0 < total; s0 += stride) {   // wave-uniform
        const long long sidx = s0 + g;
        const bool valid = sidx < total;
        const long long c = valid ? sidx / p.K : 0;
        const double* x = p.ctrl + c * (N + 1);
        bool pad = false;
        for (int i = 0; i <= N; ++i) pad |= (x[i] != x[i]);
        const bool live = valid && !pad;                        // sub-group-uniform
        const double* gd = p.draws + c * p.draw_cstride + (sidx - c * p.K) * 3 * N;
        // ---- assemble H = HH + Z + diag(x)  (noise_model.py:79-85, :100-104, :122-147); idle slots hold zeros
        for (int e = sl; e < N * N; e += SUB) {
            const int i = e / N, j = e - i * N;
            double re = 0.0, im = 0.0;
            if (live) {
                if (i == j) re = x[i] + p.h0.diag[i] + gd[3 * i];
                else if (i == j + 1) { re = p.h0.off[j] + gd[3 * i + 1]; im = gd[3 * i + 2]; }
                else if (j == i + 1) { re = p.h0.off[i] + gd[3 * j + 1]; im = -gd[3 * j + 2]; }
                if (p.ring && N > 2 && ((i == N - 1 && j == 0) || (i == 0 && j == N - 1))) re += 1.0;
            }
            Ar[e] = re;
            Ai[e] = im;
        }
        for (int k = sl; k < N; k += SUB) {
            vir[k] = (k == p.in) ? 1.0 : 0.0;
            vii[k] = 0.0;
            vor[k] = (k == p.out) ? 1.0 : 0.0;
            voi[k] = 0.0;
        }
        wave_fence();
        // Frobenius norm (for the stopping test)
        double fro = 0.0;
        for (int e = sl; e < N * N; e += SUB) fro += Ar[e] * Ar[e] + Ai[e] * Ai[e];
        fro = sub_sum(fro);

        for (int sweep = 0; sweep < kJacMaxSweeps; ++sweep) {
            double off = 0.0;
            for (int e = sl; e < N * N; e += SUB) {
                const int i = e / N, j = e - i * N;
                if (i != j) off += Ar[e] * Ar[e] + Ai[e] * Ai[e];
            }
            off = sub_sum(off);
            // |offdiag| <= 3e-16 |A|: one sweep past 1e-8 gets here; the wave stops when all its samples have
            if (__all(off <= 1e-31 * fro)) break;
            for (int r = 0; r < m; ++r) {
                // ---- rotation parameters of this round's pairs
                if (sl < npair) {
                    int pp = (sl == 0) ? m : (r + sl) % m;
                    int qq = (sl == 0) ? r : (r - sl + m) % m;
                    double cs = 1.0, sr = 0.0, si = 0.0;
                    if (pp < N && qq < N) {
                        const double br = Ar[pp * N + qq], bi = Ai[pp * N + qq];
                        const double b2 = br * br + bi * bi;
                        if (b2 > 1e-290) {
                            const double babs = sqrt(b2);
                            const double tau = (Ar[qq * N + qq] - Ar[pp * N + pp]) / (2.0 * babs);
                            const double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
                            cs = 1.0 / sqrt(1.0 + t * t);
                            const double sc = t * cs / babs;           // s = sc * beta
                            sr = sc * br;
                            si = sc * bi;
                        }
                    }
                    par[3 * sl] = cs;
                    par[3 * sl + 1] = sr;
                    par[3 * sl + 2] = si;
                }
                wave_fence();
                // ---- rows:  a'_pj = c a_pj - s a_qj ;  a'_qj = conj(s) a_pj + c a_qj
                for (int w = sl; w < npair * N; w += SUB) {
                    const int k = w / N, j = w - k * N;
                    const int pp = (k == 0) ? m : (r + k) % m;
                    const int qq = (k == 0) ? r : (r - k + m) % m;
                    if (pp < N && qq < N) {
                        const double cs = par[3 * k], sr = par[3 * k + 1], si = par[3 * k + 2];
                        const double pr = Ar[pp * N + j], pi = Ai[pp * N + j];
                        const double qr = Ar[qq * N + j], qi = Ai[qq * N + j];
                        Ar[pp * N + j] = cs * pr - (sr * qr - si * qi);
                        Ai[pp * N + j] = cs * pi - (sr * qi + si * qr);
                        Ar[qq * N + j] = (sr * pr + si * pi) + cs * qr;
                        Ai[qq * N + j] = (sr * pi - si * pr) + cs * qi;
                    }
                }
                wave_fence();
                // ---- columns:  a'_ip = c a_ip - conj(s) a_iq ;  a'_iq = s a_ip + c a_iq   (same for the V rows)
                for (int w = sl; w < npair * (N + 2); w += SUB) {
                    const int k = w / (N + 2), i = w - k * (N + 2);
                    const int pp = (k == 0) ? m : (r + k) % m;
                    const int qq = (k == 0) ? r : (r - k + m) % m;
                    if (pp < N && qq < N) {
                        const double cs = par[3 * k], sr = par[3 * k + 1], si = par[3 * k + 2];
                        double *xr, *xi;
                        int ip, iq;
                        if (i < N) { xr = Ar; xi = Ai; ip = i * N + pp; iq = i * N + qq; }
                        else if (i == N) { xr = vir; xi = vii; ip = pp; iq = qq; }
                        else { xr = vor; xi = voi; ip = pp; iq = qq; }
                        const double pr = xr[ip], pi = xi[ip], qr = xr[iq], qi = xi[iq];
                        xr[ip] = cs * pr - (sr * qr + si * qi);
                        xi[ip] = cs * pi - (sr * qi - si * qr);
                        xr[iq] = (sr * pr - si * pi) + cs * qr;
                        xi[iq] = (sr * pi + si * pr) + cs * qi;
                    }
                }
                wave_fence();
                // annihilated elements are exactly zero in exact arithmetic: store that
                if (sl < npair) {
                    const int pp = (sl == 0) ? m : (r + sl) % m;
                    const int qq = (sl == 0) ? r : (r - sl + m) % m;
                    if (pp < N && qq < N && (par[3 * sl + 1] != 0.0 || par[3 * sl + 2] != 0.0)) {
                        Ar[pp * N + qq] = 0.0; Ai[pp * N + qq] = 0.0;
                        Ar[qq * N + pp] = 0.0; Ai[qq * N + pp] = 0.0;
                        Ai[pp * N + pp] = 0.0; Ai[qq * N + qq] = 0.0;
                    }
                }
                wave_fence();
            }
        }
        // ---- phi = sum_k V[out,k] exp(-i T lam_k) conj(V[in,k])
        const double T = fabs(x[N]);
        double re = 0.0, im = 0.0;
        for (int k = sl; k < N; k += SUB) {
            double sk, ck;
            rc::sincos_reduced(T * Ar[k * N + k], sk, ck);
            const double wr = vor[k] * vir[k] + voi[k] * vii[k];                 // V_out conj(V_in)
            const double wi = voi[k] * vir[k] - vor[k] * vii[k];
            re += wr * ck + wi * sk;                                             // (wr + i wi)(ck - i sk)
            im += wi * ck - wr * sk;
        }
        re = sub_sum(re);
        im = sub_sum(im);
        if (sl == 0 && valid) p.fid[sidx] = pad ? __builtin_nan("") : re * re + im * im;
        wave_fence();
    }
}

// ------------------------------------------------------------------------------------------------
// fidelity kernel: dense complex (possibly NON-Hermitian) Hamiltonian, one WAVE per sample, Pade expm in LDS
// ------------------------------------------------------------------------------------------------
// The reference's own algorithm shape on the device: U = expm(-i T H) by Pade approximation with scaling and
// squaring (orders 3/5/7/9/13, thresholds and coefficients of Higham 2005 - the published algorithm behind
// scipy.linalg.expm, noise_model.py:105), every matrix in LDS, the 64 lanes sharing each matrix product, the
// linear solve (partial pivoting) and the squarings; control flow is wave-uniform (one sample per wave).
// It exists for the perturbations the eigen-solver kernels cannot take: `directional_perturbation`
// (noise_model.py:150-201) writes a - ib on the DIAGONAL for its diagonal directions (the second assignment at
// :198-199 overwrites the first), i.e. a non-Hermitian H, passed here as an imaginary-diagonal plane next to the
// usual draws.  With diag_imag = NULL it is a third, algorithmically independent cross-check of the other kernels.
struct ExpmParams {
    const double* ctrl;
    const double* draws;        // [C][K][N][3] (stride draw_cstride per controller)
    const double* diag_imag;    // [C][K][N] or NULL: H[i][i] += 1j * diag_imag
    double* fid;
    long long C, K;
    long long draw_cstride, imag_cstride;
    int N, in, out, ring;
    StaticH h0;
};

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

constexpr int kExpmWaves = 2;
constexpr int kExpmBufs = 7;

__device__ __forceinline__ void mat_mul(int n, const cplx* A, const cplx* B, cplx* Cm, int lane) {
    for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e - i * n;
        double re = 0.0, im = 0.0;
        for (int k = 0; k < n; ++k) {
            const cplx a = A[i * n + k], b = B[k * n + j];
            re += a.re * b.re - a.im * b.im;
            im += a.re * b.im + a.im * b.re;
        }
        Cm[e] = {re, im};
    }
    wave_fence();
}

// Pade numerator coefficients of degree 3 / 5 / 7 / 9 (Higham 2005, table 10.4), one zero-padded row per degree
__device__ const double g_pade_low[4][10] = {
    {120, 60, 12, 1, 0, 0, 0, 0, 0, 0},
    {30240, 15120, 3360, 420, 30, 1, 0, 0, 0, 0},
    {17297280, 8648640, 1995840, 277200, 25200, 1512, 56, 1, 0, 0},
    {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.}};

__global__ __launch_bounds__(64 * kExpmWaves) void mc_fid_expm_kernel(const ExpmParams p) {
    extern __shared__ double lds_raw[];
    const int N = p.N, nn = N * N;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    cplx* base = (cplx*)lds_raw + (size_t)wave * kExpmBufs * nn;
    cplx *A = base, *A2 = base + nn, *A4 = base + 2 * nn, *A6 = base + 3 * nn, *U = base + 4 * nn, *V = base + 5 * nn,
         *W = base + 6 * nn;
    const double b13[] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                          129060195264000., 10559470521600., 670442572800., 33522128640., 1323241920.,
                          40840800., 960960., 16380., 182., 1.};
    const long long total = p.C * p.K;
    const long long stride = (long long)gridDim.x * kExpmWaves;
    for (long long sidx = (long long)blockIdx.x * kExpmWaves + wave; sidx < total; sidx += stride) {
        const long long c = sidx / p.K, k = sidx - c * p.K;
        const double* x = p.ctrl + c * (N + 1);
        bool pad = false;
        for (int i = 0; i <= N; ++i) pad |= (x[i] != x[i]);
        if (pad) {
            if (lane == 0) p.fid[sidx] = __builtin_nan("");
            continue;
        }
        const double* g = p.draws + c * p.draw_cstride + k * 3 * N;
        const double* gi = p.diag_imag ? p.diag_imag + c * p.imag_cstride + k * N : nullptr;
        const double T = fabs(x[N]);
        // A = -i T H,  H = HH + Z + diag(x)  (noise_model.py:79-85, :100-104, :122-147 / :150-201)
        for (int e = lane; e < nn; e += 64) {
            const int i = e / N, j = e - i * N;
            double re = 0.0, im = 0.0;
            if (i == j) { re = x[i] + p.h0.diag[i] + g[3 * i]; im = gi ? gi[i] : 0.0; }
            else if (i == j + 1) { re = p.h0.off[j] + g[3 * i + 1]; im = g[3 * i + 2]; }
            else if (j == i + 1) { re = p.h0.off[i] + g[3 * j + 1]; im = -g[3 * j + 2]; }
            if (p.ring && N > 2 && ((i == N - 1 && j == 0) || (i == 0 && j == N - 1))) re += 1.0;
            A[e] = {T * im, -T * re};                         // (-i T)(re + i im)
        }
        wave_fence();
        // 1-norm
        double colsum = 0.0;
        if (lane < N)
            for (int i = 0; i < N; ++i) colsum += sqrt(A[i * N + lane].re * A[i * N + lane].re + A[i * N + lane].im * A[i * N + lane].im);
        double nrm = colsum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm = fmax(nrm, __shfl_xor(nrm, off, 64));
        int m, sq = 0;
        if (nrm <= 1.495585217958292e-2) m = 3;
        else if (nrm <= 2.539398330063230e-1) m = 5;
        else if (nrm <= 9.504178996162932e-1) m = 7;
        else if (nrm <= 2.097847961257068e0) m = 9;
        else {
            m = 13;
            const double theta13 = 5.371920351148152e0;
            if (nrm > theta13) {
                sq = (int)ceil(log2(nrm / theta13));
                const double sc = ldexp(1.0, -sq);
                for (int e = lane; e < nn; e += 64) { A[e].re *= sc; A[e].im *= sc; }
                wave_fence();
            }
        }
        mat_mul(N, A, A, A2, lane);
        if (m == 13) {
            mat_mul(N, A2, A2, A4, lane);
            mat_mul(N, A4, A2, A6, lane);
            for (int e = lane; e < nn; e += 64) {
                W[e] = {b13[13] * A6[e].re + b13[11] * A4[e].re + b13[9] * A2[e].re,
                        b13[13] * A6[e].im + b13[11] * A4[e].im + b13[9] * A2[e].im};
            }
            wave_fence();
            mat_mul(N, A6, W, V, lane);                        // V used as scratch for the U polynomial
            for (int e = lane; e < nn; e += 64) {
                const int i = e / N, j = e - i * N;
                V[e].re += b13[7] * A6[e].re + b13[5] * A4[e].re + b13[3] * A2[e].re + ((i == j) ? b13[1] : 0.0);
                V[e].im += b13[7] * A6[e].im + b13[5] * A4[e].im + b13[3] * A2[e].im;
            }
            wave_fence();
            mat_mul(N, A, V, U, lane);
            for (int e = lane; e < nn; e += 64) {
                W[e] = {b13[12] * A6[e].re + b13[10] * A4[e].re + b13[8] * A2[e].re,
                        b13[12] * A6[e].im + b13[10] * A4[e].im + b13[8] * A2[e].im};
            }
            wave_fence();
            mat_mul(N, A6, W, V, lane);
            for (int e = lane; e < nn; e += 64) {
                const int i = e / N, j = e - i * N;
                V[e].re += b13[6] * A6[e].re + b13[4] * A4[e].re + b13[2] * A2[e].re + ((i == j) ? b13[0] : 0.0);
                V[e].im += b13[6] * A6[e].im + b13[4] * A4[e].im + b13[2] * A2[e].im;
            }
            wave_fence();
        } else {
            const double* b = g_pade_low[(m - 3) >> 1];        // wave-uniform row of a constant table: scalar loads, no scratch
            cplx* A8 = W;                                      // only needed for m == 9, W is free until then
            if (m >= 5) mat_mul(N, A2, A2, A4, lane);
            if (m >= 7) mat_mul(N, A4, A2, A6, lane);
            if (m >= 9) mat_mul(N, A6, A2, A8, lane);
            for (int e = lane; e < nn; e += 64) {
                const int i = e / N, j = e - i * N;
                double ur = b[3] * A2[e].re, ui = b[3] * A2[e].im, vr = b[2] * A2[e].re, vi = b[2] * A2[e].im;
                if (m >= 5) { ur += b[5] * A4[e].re; ui += b[5] * A4[e].im; vr += b[4] * A4[e].re; vi += b[4] * A4[e].im; }
                if (m >= 7) { ur += b[7] * A6[e].re; ui += b[7] * A6[e].im; vr += b[6] * A6[e].re; vi += b[6] * A6[e].im; }
                if (m >= 9) { ur += b[9] * A8[e].re; ui += b[9] * A8[e].im; vr += b[8] * A8[e].re; vi += b[8] * A8[e].im; }
                if (i == j) { ur += b[1]; vr += b[0]; }
                V[e] = {vr, vi};
                A4[e] = {ur, ui};                              // A4 (not needed any more) holds the U polynomial
            }
            wave_fence();
            mat_mul(N, A, A4, U, lane);
        }
        // solve (V - U) X = (V + U):  P := V - U in A2, X := V + U in A4
        cplx* P = A2;
        cplx* X = A4;
        for (int e = lane; e < nn; e += 64) {
            P[e] = {V[e].re - U[e].re, V[e].im - U[e].im};
            X[e] = {V[e].re + U[e].re, V[e].im + U[e].im};
        }
        wave_fence();
        for (int col = 0; col < N; ++col) {
            // pivot search (lanes = rows)
            double mag = -1.0;
            int row = lane;
            if (lane >= col && lane < N) mag = P[lane * N + col].re * P[lane * N + col].re + P[lane * N + col].im * P[lane * N + col].im;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double om = __shfl_xor(mag, off, 64);
                const int orow = __shfl_xor(row, off, 64);
                if (om > mag || (om == mag && orow < row)) { mag = om; row = orow; }
            }
            const int piv = row;                               // wave-uniform after the butterfly
            if (piv != col) {
                for (int j = lane; j < 2 * N; j += 64) {
                    cplx* M = (j < N) ? P : X;
                    const int jj = (j < N) ? j : j - N;
                    const cplx t = M[col * N + jj];
                    M[col * N + jj] = M[piv * N + jj];
                    M[piv * N + jj] = t;
                }
                wave_fence();
            }
            const cplx d = P[col * N + col];
            const double den = d.re * d.re + d.im * d.im;
            const cplx dinv = {d.re / den, -d.im / den};
            // eliminate below: work items (row r > col, column j of [P | X])
            const int rows = N - 1 - col;
            for (int w = lane; w < rows * 2 * N; w += 64) {
                const int r = col + 1 + w / (2 * N), j = w % (2 * N);
                cplx* M = (j < N) ? P : X;
                const int jj = (j < N) ? j : j - N;
                if (j < N && jj < col) continue;               // already zero
                const cplx f = cmul(P[r * N + col], dinv);
                const cplx t = cmul(f, M[col * N + jj]);
                if (!(j < N && jj == col)) { M[r * N + jj].re -= t.re; M[r * N + jj].im -= t.im; }
            }
            wave_fence();
            // the multipliers' column is zeroed last (every work item above read P[r][col])
            for (int r = col + 1 + lane; r < N; r += 64) P[r * N + col] = {0.0, 0.0};
            wave_fence();
        }
        // back substitution, lanes = columns of X
        for (int row = N - 1; row >= 0; --row) {
            const cplx d = P[row * N + row];
            const double den = d.re * d.re + d.im * d.im;
            const cplx dinv = {d.re / den, -d.im / den};
            if (lane < N) {
                cplx acc = X[row * N + lane];
                for (int k2 = row + 1; k2 < N; ++k2) {
                    const cplx t = cmul(P[row * N + k2], X[k2 * N + lane]);
                    acc.re -= t.re;
                    acc.im -= t.im;
                }
                X[row * N + lane] = cmul(acc, dinv);
            }
            wave_fence();
        }
        // squarings
        cplx* E = X;
        cplx* Tm = U;
        for (int q = 0; q < sq; ++q) {
            mat_mul(N, E, E, Tm, lane);
            cplx* sw = E; E = Tm; Tm = sw;
        }
        if (lane == 0) {
            const cplx phi = E[p.out * N + p.in];
            p.fid[sidx] = phi.re * phi.re + phi.im * phi.im;
        }
        wave_fence();
    }
}

// ------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------
constexpr int kMaxQ = 8;
constexpr int kRedThreads = 512;
constexpr int kRedWaves = kRedThreads / 64;
constexpr int kRedCache = 32;          // fidelities a thread keeps in registers: rows up to 16384 are read once

struct RedParams {
    const double* fid;   // [C][K]
    long long C, K;
    int nq;
    double thr[kMaxQ];
    double eps;
    double *rim1, *stdv, *minf, *q;   // variant-major, may be null
};

__device__ __forceinline__ double clip01(double v) { return fmin(fmax(v, 0.0), 1.0); }

// Block-wide sum with a fixed combination order (wave shuffle tree, then the waves in index order): bitwise
// reproducible run to run.  Every thread returns the total.  (Used by the small kernels below.)
__device__ __forceinline__ double block_sum(double v, double* scratch /*[kRedWaves]*/) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int w = 1; w < kRedWaves; ++w) r += scratch[w];
    return r;
}

// One workgroup per controller.  The row is read from HBM once and kept in registers (K <= 16384; longer rows
// are re-read, from L2, for the second pass).  Pass 1: sums / min / NaN flag / threshold counts of the three
// DKW variants - all partials of a wave go to LDS together, ONE barrier, every thread combines them in wave
// order (deterministic).  Pass 2: centred second moments (np.std is the two-pass population form).
template <int NQ>
__global__ __launch_bounds__(kRedThreads) void reduce_kernel(const RedParams p) {
    constexpr int NV = 5;                                  // sum[3], min, nan
    constexpr int NC = 3 * NQ;                             // threshold counts cnt[3][NQ]: integers (exact, half the registers)
    constexpr int kCache = (NQ <= 2) ? kRedCache : 4;       // the many-threshold variant has no registers to spare
    __shared__ double part[kRedWaves][NV];
    __shared__ unsigned int partc[kRedWaves][NC > 0 ? NC : 1];
    __shared__ double part2[kRedWaves][3];
    const long long c = blockIdx.x;
    const double* row = p.fid + c * p.K;
    const double K = (double)p.K;
    const bool cached = p.K <= (long long)kCache * kRedThreads;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    double val[kCache];
    double acc[NV];
    unsigned int cnt[NC > 0 ? NC : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NC; ++i) cnt[i] = 0u;
    acc[3] = INFINITY;
    auto pass1 = [&](double f) {
        const double fv[3] = {f, clip01(f - p.eps), clip01(f + p.eps)};
        acc[4] += (f != f) ? 1.0 : 0.0;
        acc[3] = fmin(acc[3], f);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            acc[v] += fv[v];
#pragma unroll
            for (int j = 0; j < NQ; ++j) cnt[v * NQ + j] += (fv[v] >= p.thr[j]) ? 1u : 0u;   // < 2^32 per thread (K < 2^41)
        }
    };
    if (cached) {
#pragma unroll
        for (int i = 0; i < kCache; ++i) {
            const long long k = (long long)i * kRedThreads + threadIdx.x;
            val[i] = (k < p.K) ? row[k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < kCache; ++i)
            if ((long long)i * kRedThreads + threadIdx.x < p.K) pass1(val[i]);
    } else {
        // long rows (K > kCache * kRedThreads): 8 loads in flight per thread, then the accumulation
        long long k = threadIdx.x;
        for (; k + 7 * kRedThreads < p.K; k += 8 * kRedThreads) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[k + u * kRedThreads];
#pragma unroll
            for (int u = 0; u < 8; ++u) pass1(v[u]);
        }
        for (; k < p.K; k += kRedThreads) pass1(row[k]);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double r = (i == 3) ? wave_min(acc[i]) : wave_sum(acc[i]);
        if (lane == 0) part[wave][i] = r;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        unsigned int r = cnt[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
        if (lane == 0) partc[wave][i] = r;
    }
    __syncthreads();
    double tot[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double r = part[0][i];
#pragma unroll
        for (int w = 1; w < kRedWaves; ++w) r = (i == 3) ? fmin(r, part[w][i]) : r + part[w][i];
        tot[i] = r;
    }
    const bool has_nan = tot[4] != 0.0;
    const double mean[3] = {tot[0] / K, tot[1] / K, tot[2] / K};

    double ss[3] = {0, 0, 0};
    if (p.stdv) {
        auto pass2 = [&](double f) {
            const double fv[3] = {f, clip01(f - p.eps), clip01(f + p.eps)};
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double dlt = fv[v] - mean[v];
                ss[v] = fma(dlt, dlt, ss[v]);
            }
        };
        if (cached) {
#pragma unroll
            for (int i = 0; i < kCache; ++i)
                if ((long long)i * kRedThreads + threadIdx.x < p.K) pass2(val[i]);
        } else {
            long long k = threadIdx.x;
            for (; k + 7 * kRedThreads < p.K; k += 8 * kRedThreads) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = row[k + u * kRedThreads];
#pragma unroll
                for (int u = 0; u < 8; ++u) pass2(v[u]);
            }
            for (; k < p.K; k += kRedThreads) pass2(row[k]);
        }
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const double r = wave_sum(ss[v]);
            if (lane == 0) part2[wave][v] = r;
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            double r = part2[0][v];
#pragma unroll
            for (int w = 1; w < kRedWaves; ++w) r += part2[w][v];
            ss[v] = r;
        }
    }
    if (threadIdx.x == 0) {
        const double nanv = __builtin_nan("");
        const double mins[3] = {tot[3], clip01(tot[3] - p.eps), clip01(tot[3] + p.eps)};
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            // RIM_1 = W1(F, delta(x-1)) = mean(1 - F)   (wd_sortof_fast_implementation.py:82-116)
            if (p.rim1) p.rim1[v * p.C + c] = has_nan ? nanv : 1.0 - mean[v];
            if (p.stdv) p.stdv[v * p.C + c] = has_nan ? nanv : sqrt(ss[v] / K);
            if (p.minf) p.minf[v * p.C + c] = has_nan ? nanv : mins[v];
            if (p.q) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    if (j < p.nq) {
                        unsigned long long n = 0;
#pragma unroll
                        for (int w = 0; w < kRedWaves; ++w) n += partc[w][v * NQ + j];
                        p.q[((long long)v * p.nq + j) * p.C + c] = (double)n / K;
                    }
                }
            }
        }
    }
}

// Short rows (K <= 2048), many of them - the paper-scale layout (L x C = 11 000 rows of 100 draws per algorithm,
// mcsim.py:204-207) and the ARIM scan (checkpoints x controllers x levels rows of 100, gen_fig_8...py:37-69): one
// WAVE per row, 4 rows per workgroup, the row in registers (<= 32 values per lane), butterfly reductions (every lane
// ends with the total: no LDS, no barrier), same two-pass arithmetic and outputs as reduce_kernel.
constexpr int kWaveRowMaxK = 2048;
template <typename T>
__device__ __forceinline__ T wave_allsum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int NQ>
__global__ __launch_bounds__(256) void reduce_rows_wave_kernel(const RedParams p) {
    constexpr int kC = kWaveRowMaxK / 64;
    constexpr int NC = 3 * NQ;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long c = (long long)blockIdx.x * 4 + wave;
    if (c >= p.C) return;                                   // wave-uniform
    const double* row = p.fid + c * p.K;
    const int Ki = (int)p.K;
    const double K = (double)p.K;
    double val[kC];
#pragma unroll
    for (int i = 0; i < kC; ++i) {
        const int k = i * 64 + lane;
        val[i] = (i * 64 < Ki && k < Ki) ? row[k] : 0.0;
    }
    double sum[3] = {0, 0, 0}, mn = INFINITY, nan = 0.0;
    unsigned int cnt[NC > 0 ? NC : 1];
#pragma unroll
    for (int i = 0; i < NC; ++i) cnt[i] = 0u;
#pragma unroll
    for (int i = 0; i < kC; ++i) {
        if (i * 64 < Ki && i * 64 + lane < Ki) {
            const double f = val[i];
            const double fv[3] = {f, clip01(f - p.eps), clip01(f + p.eps)};
            nan += (f != f) ? 1.0 : 0.0;
            mn = fmin(mn, f);
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                sum[v] += fv[v];
#pragma unroll
                for (int j = 0; j < NQ; ++j) cnt[v * NQ + j] += (fv[v] >= p.thr[j]) ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) sum[v] = wave_allsum(sum[v]);
    nan = wave_allsum(nan);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off, 64));
#pragma unroll
    for (int i = 0; i < NC; ++i) cnt[i] = wave_allsum(cnt[i]);
    const bool has_nan = nan != 0.0;
    const double mean[3] = {sum[0] / K, sum[1] / K, sum[2] / K};
    double ss[3] = {0, 0, 0};
    if (p.stdv) {
#pragma unroll
        for (int i = 0; i < kC; ++i) {
            if (i * 64 < Ki && i * 64 + lane < Ki) {
                const double f = val[i];
                const double fv[3] = {f, clip01(f - p.eps), clip01(f + p.eps)};
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const double dlt = fv[v] - mean[v];
                    ss[v] = fma(dlt, dlt, ss[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 3; ++v) ss[v] = wave_allsum(ss[v]);
    }
    if (lane == 0) {
        const double nanv = __builtin_nan("");
        const double mins[3] = {mn, clip01(mn - p.eps), clip01(mn + p.eps)};
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            if (p.rim1) p.rim1[v * p.C + c] = has_nan ? nanv : 1.0 - mean[v];
            if (p.stdv) p.stdv[v * p.C + c] = has_nan ? nanv : sqrt(ss[v] / K);
            if (p.minf) p.minf[v * p.C + c] = has_nan ? nanv : mins[v];
            if (p.q) {
#pragma unroll
                for (int j = 0; j < NQ; ++j)
                    if (j < p.nq) p.q[((long long)v * p.nq + j) * p.C + c] = (double)cnt[v * NQ + j] / K;
            }
        }
    }
}

// p-RIM (wd_sortof_fast_implementation.py:147-174): (mean_k (1 - f_k)^p)^(1/p), one workgroup per controller.
__global__ __launch_bounds__(kRedThreads) void rim_p_kernel(const double* fid, long long C, long long K, double pw,
                                                            double* out) {
    __shared__ double sd[kRedWaves];
    const long long c = blockIdx.x;
    const double* row = fid + c * K;
    double acc = 0.0;
    for (long long k = threadIdx.x; k < K; k += kRedThreads) acc += pow(1.0 - row[k], pw);
    acc = block_sum(acc, sd);
    if (threadIdx.x == 0) out[c] = pow(acc / (double)K, 1.0 / pw);
}

// Row sort (ECDF).  K <= 16384: sort_rows_merge_kernel - one fused launch, merge sort in LDS, no padding.  Longer rows:
// bitonic network on rows padded with +inf to P = 2^k: sort_chunk16_kernel per 16384-element chunk of a workspace [C][P]
// + sort_global_fused_kernel for the strides >= 16384 (K = 10^5, BASELINE config 4: 7 launches).  NaN rows (padded
// controllers) are detected and copied through unchanged.
constexpr int kSortChunk = 16384;
constexpr int kSortThreads = 1024;

// register building blocks of the sort kernels: sign flip (descending segments run through the ascending network
// on sign-flipped keys) and the strides <= 8 of a bitonic network over 16 registers
__device__ __forceinline__ double sort_flip(double v) {
    return __hiloint2double(__double2hiint(v) ^ (int)0x80000000, __double2loint(v));
}
__device__ __forceinline__ void sort_regs16(double (&v)[16], int first_stride) {
#pragma unroll
    for (int stride = 8; stride >= 1; stride >>= 1) {
        if (stride > first_stride) continue;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if ((j & stride) == 0) {
                const double a = v[j], b = v[j + stride];
                v[j] = fmin(a, b);
                v[j + stride] = fmax(a, b);
            }
        }
    }
}
// Rows of up to 16384 samples, any K: merge sort (sort_core.h).  Thread t sorts its 16 elements in registers, then
// log2(K/16) merge-path levels through one padded LDS buffer; no power-of-two padding (K = 10 000 costs 10 000, not
// 16 384), ~30 dependent LDS reads per thread and level instead of the bitonic network's ~160 LDS operations.
__global__ __launch_bounds__(kSortThreads) void sort_rows_merge_kernel(const double* fid, double* out, long long K, int n) {
    extern __shared__ double buf[];                                  // pad(n) + 1 doubles
    const long long c = blockIdx.x;
    const double* row = fid + c * K;
    const int t = threadIdx.x;
    const bool active = 16 * t < n;
    double v[16];
    int bad = 0;
    if (active) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long long i = 16LL * t + j;
            v[j] = (i < K) ? row[i] : INFINITY;
            bad |= (v[j] != v[j]);
        }
    }
    if (__syncthreads_or(bad)) {                                     // NaN row (padded controller): copied through
        for (long long i = t; i < K; i += blockDim.x) out[c * K + i] = row[i];
        return;
    }
    if (active) {                                                    // 16-element run, ascending (bitonic in registers)
#pragma unroll
        for (int size = 2; size <= 16; size <<= 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (size < 16 && (j & size) != 0) v[j] = sort_flip(v[j]);
            sort_regs16(v, size >> 1);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (size < 16 && (j & size) != 0) v[j] = sort_flip(v[j]);
        }
    }
    for (int L = 16; L < n; L <<= 1) {
        __syncthreads();                                             // readers of the previous level are done
        if (active) {
#pragma unroll
            for (int j = 0; j < 16; ++j) buf[17 * t + j] = v[j];
        }
        __syncthreads();
        if (active) rcs::merge_level16(buf, n, L, t, v);
    }
    if (active) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long long i = 16LL * t + j;
            if (i < K) out[c * K + i] = v[j];
        }
    }
}

// Long rows (P > 16384): the same register / butterfly scheme per 16384-element chunk of a workspace row, for the
// network sizes [size_lo, size_hi] restricted to strides < 16384 (larger strides: sort_global_fused_kernel).  The
// first pass reads the caller's row (padding with +inf, flagging NaN rows), the last one writes the caller's output.
__global__ __launch_bounds__(kSortThreads) void sort_chunk16_kernel(const double* fid, double* work, double* out,
                                                                    int* nanflag, long long K, long long P,
                                                                    long long size_lo, long long size_hi, int first,
                                                                    int last) {
    extern __shared__ double buf[];                                  // 16384 * 17 / 16 doubles
    constexpr int CH = kSortChunk;
    const long long c = blockIdx.x;
    const long long gbase = (long long)blockIdx.y * CH;
    const int t = threadIdx.x;
    double v[16];
    if (first) {
        int bad = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long long i = gbase + 16LL * t + j;
            v[j] = (i < K) ? fid[c * K + i] : INFINITY;
            bad |= (v[j] != v[j]);
        }
        if (__syncthreads_or(bad) && t == 0) atomicOr(&nanflag[c], 1);
#pragma unroll
        for (int size = 2; size <= 16; size <<= 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if ((size < 16) ? ((j & size) != 0) : ((t & 1) != 0)) v[j] = sort_flip(v[j]);
            sort_regs16(v, size >> 1);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if ((size < 16) ? ((j & size) != 0) : ((t & 1) != 0)) v[j] = sort_flip(v[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = work[c * P + gbase + 16LL * t + j];
    }
    for (long long size = (size_lo < 32 ? 32 : size_lo); size <= size_hi; size <<= 1) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) buf[17 * t + j] = v[j];
        int hi = (int)((size >> 1) < (CH >> 1) ? (size >> 1) : (CH >> 1));
        while (hi >= 16) {
            const int nleft = 31 - __builtin_clz(hi) - 3;
            const int take = nleft >= 4 ? 4 : nleft;
            const int S = hi >> (take - 1);
            const int lgS = 31 - __builtin_clz(S);
            __syncthreads();
            const int base = (t & (S - 1)) | ((t >> lgS) << (lgS + 4));
            int pos[16];
            bool down[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int e = base + k * S;
                pos[k] = e + (e >> 4);
                down[k] = ((gbase + e) & size) != 0;
                const double x = buf[pos[k]];
                v[k] = down[k] ? sort_flip(x) : x;
            }
            switch (take) {
                case 4: sort_regs16(v, 8); break;
                case 3: sort_regs16(v, 4); break;
                case 2: sort_regs16(v, 2); break;
                default: sort_regs16(v, 1); break;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) buf[pos[k]] = down[k] ? sort_flip(v[k]) : v[k];
            hi = S >> 1;
        }
        __syncthreads();
        const bool dn = ((gbase + 16 * t) & size) != 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double x = buf[17 * t + j];
            v[j] = dn ? sort_flip(x) : x;
        }
        sort_regs16(v, 8);
        if (dn) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = sort_flip(v[j]);
        }
    }
    if (last) {
        const bool nanrow = nanflag[c] != 0;                         // NaN row (padded controller): copied through
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long long i = gbase + 16LL * t + j;
            if (i < K) out[c * K + i] = nanrow ? fid[c * K + i] : v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) work[c * P + gbase + 16LL * t + j] = v[j];
    }
}

// All network steps of one size whose stride is >= 16384, up to four per launch: a thread gathers the 2^TAKE elements
// base + k S that the steps S 2^(TAKE-1) .. S couple, runs them in registers and writes them back (one HBM round trip
// instead of TAKE).
template <int TAKE>
__global__ __launch_bounds__(256) void sort_global_fused_kernel(double* work, long long P, long long size, long long S) {
    constexpr int R = 1 << TAKE;
    const long long c = blockIdx.x;
    double* row = work + c * P;
    const int lgS = 63 - __builtin_clzll((unsigned long long)S);
    for (long long g = (long long)blockIdx.y * 256 + threadIdx.x; g < (P >> TAKE); g += (long long)gridDim.y * 256) {
        const long long base = (g & (S - 1)) | ((g >> lgS) << (lgS + TAKE));
        const bool down = (base & size) != 0;                        // bit above every coupled stride: uniform
        double v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double x = row[base + k * S];
            v[k] = down ? sort_flip(x) : x;
        }
#pragma unroll
        for (int stride = R >> 1; stride >= 1; stride >>= 1) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if ((j & stride) == 0) {
                    const double a = v[j], b = v[j + stride];
                    v[j] = fmin(a, b);
                    v[j + stride] = fmax(a, b);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) row[base + k * S] = down ? sort_flip(v[k]) : v[k];
    }
}


// ------------------------------------------------------------------------------------------------
// counter-based Gaussian draws (explicitly NOT the reference's RNG: for sample spaces too large to draw on the
// host, e.g. BASELINE config 4 = 2.1e9 draws).  Philox4x32-10 keyed by `seed`; element e of the stream comes from
// counter (e >> 1): two 53-bit uniforms -> Box-Muller pair, element parity picks cos / sin.  Any element can be
// regenerated independently (oracle/philox_host.py does, for the parity tests).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                              unsigned int k0, unsigned int k1, unsigned int (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        const unsigned int n3 = (unsigned int)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// (ln c_k, 1 / c_k), c_k = 1/2 + (k + 1) / 256, k = 0..127: ln f = ln c_k + log1p((f - c_k) / c_k) for f in [1/2, 1)
#define RC_LN_TABLE_VALUES \
    -0.6853650401178903, 1.9844961240310077, -0.6776429940239801, 1.9692307692307693, \
    -0.6699801212784109, 1.9541984732824427, -0.6623755218931916, 1.9393939393939394, \
    -0.6548283162578087, 1.9248120300751879, -0.6473376445286511, 1.9104477611940298, \
    -0.639902666041133, 1.8962962962962964, -0.6325225587435105, 1.8823529411764706, \
    -0.6251965186514375, 1.8686131386861313, -0.6179237593223578, 1.855072463768116, \
    -0.6107035113488707, 1.841726618705036, -0.6035350218702582, 1.8285714285714285, \
    -0.5964175541013942, 1.8156028368794326, -0.5893503868783018, 1.8028169014084507, \
    -0.5823328142196552, 1.7902097902097902, -0.5753641449035618, 1.7777777777777777, \
    -0.5684437020589881, 1.7655172413793103, -0.561570822771226, 1.7534246575342465, \
    -0.5547448577008262, 1.7414965986394557, -0.5479651707154474, 1.7297297297297298, \
    -0.5412311385341033, 1.7181208053691275, -0.5345421503833068, 1.7066666666666668, \
    -0.5278976076646381, 1.695364238410596, -0.5212969236332861, 1.6842105263157894, \
    -0.514739523087127, 1.673202614379085, -0.5082248420659333, 1.6623376623376624, \
    -0.5017523275603158, 1.6516129032258065, -0.4953214372300254, 1.641025641025641, \
    -0.4889316391312544, 1.6305732484076434, -0.48258241145259567, 1.620253164556962, \
    -0.47627324225933093, 1.610062893081761, -0.4700036292457356, 1.6, \
    -0.4637730794950995, 1.5900621118012421, -0.4575811092471784, 1.5802469135802468, \
    -0.4514272436728001, 1.5705521472392638, -0.44531101665536404, 1.5609756097560976, \
    -0.4392319705789819, 1.5515151515151515, -0.43318965612301924, 1.5421686746987953, \
    -0.42718363206280735, 1.532934131736527, -0.42121346507630353, 1.5238095238095237, \
    -0.415278729556489, 1.514792899408284, -0.4093790074293007, 1.5058823529411764, \
    -0.40351388797690263, 1.4970760233918128, -0.39768296766610944, 1.4883720930232558, \
    -0.39188584998178355, 1.4797687861271676, -0.38612214526503347, 1.471264367816092, \
    -0.38039147055604844, 1.4628571428571429, -0.3746934494414107, 1.4545454545454546, \
    -0.36902771190573336, 1.4463276836158192, -0.3633938941874773, 1.4382022471910112, \
    -0.3577916386388075, 1.4301675977653632, -0.3522205935893521, 1.4222222222222223, \
    -0.3466804132137367, 1.4143646408839778, -0.34117075740276714, 1.4065934065934067, \
    -0.33569129163814154, 1.3989071038251366, -0.33024168687057687, 1.391304347826087, \
    -0.32482161940123766, 1.3837837837837839, -0.3194307707663612, 1.3763440860215055, \
    -0.31406882762497584, 1.3689839572192513, -0.3087354816496133, 1.3617021276595744, \
    -0.3034304294199201, 1.3544973544973544, -0.29815337231907635, 1.3473684210526315, \
    -0.2929040164329326, 1.3403141361256545, -0.2876820724517809, 1.3333333333333333, \
    -0.2824872555746769, 1.3264248704663213, -0.27731928541623435, 1.3195876288659794, \
    -0.27217788591581565, 1.3128205128205128, -0.26706278524904525, 1.3061224489795917, \
    -0.26197371574157396, 1.299492385786802, -0.2569104137850272, 1.292929292929293, \
    -0.2518726197550701, 1.2864321608040201, -0.24686007793152578, 1.28, \
    -0.24187253642048673, 1.2736318407960199, -0.2369097470783577, 1.2673267326732673, \
    -0.23197146543777514, 1.2610837438423645, -0.22705745063534608, 1.2549019607843137, \
    -0.2221674653411543, 1.248780487804878, -0.2173012756899814, 1.2427184466019416, \
    -0.2124586512141934, 1.2367149758454106, -0.2076393647782445, 1.2307692307692308, \
    -0.20284319251475147, 1.2248803827751196, -0.1980699137620938, 1.2190476190476192, \
    -0.19331931100349597, 1.2132701421800949, -0.18859116980755003, 1.2075471698113207, \
    -0.18388527877013736, 1.2018779342723005, -0.179201429457711, 1.1962616822429906, \
    -0.17453941635189968, 1.1906976744186046, -0.16989903679539747, 1.1851851851851851, \
    -0.16528009093910292, 1.1797235023041475, -0.16068238169047347, 1.1743119266055047, \
    -0.15610571466306167, 1.1689497716894977, -0.15154989812720093, 1.1636363636363636, \
    -0.14701474296180966, 1.158371040723982, -0.14250006260728304, 1.1531531531531531, \
    -0.13800567301944372, 1.147982062780269, -0.13353139262452263, 1.1428571428571428, \
    -0.12907704227514236, 1.1377777777777778, -0.1246424452072766, 1.1327433628318584, \
    -0.1202274269981598, 1.1277533039647578, -0.1158318155251217, 1.1228070175438596, \
    -0.11145544092532282, 1.1179039301310043, -0.1070981355563671, 1.1130434782608696, \
    -0.10275973395776894, 1.1082251082251082, -0.09844007281325252, 1.103448275862069, \
    -0.09413899091386191, 1.0987124463519313, -0.08985632912186105, 1.0940170940170941, \
    -0.08559193033540351, 1.0893617021276596, -0.0813456394539524, 1.0847457627118644, \
    -0.07711730334443129, 1.080168776371308, -0.07290677080808779, 1.0756302521008403, \
    -0.06871389254805181, 1.0711297071129706, -0.06453852113757118, 1.0666666666666667, \
    -0.06038051098890748, 1.062240663900415, -0.05623971832287608, 1.0578512396694215, \
    -0.05211600113901402, 1.0534979423868314, -0.048009219186360606, 1.0491803278688525, \
    -0.04391923393483549, 1.0448979591836736, -0.039845908547199674, 1.0406504065040652, \
    -0.03578910785158528, 1.0364372469635628, -0.0317486983145803, 1.032258064516129, \
    -0.027724548014854862, 1.0281124497991967, -0.023716526617316044, 1.024, \
    -0.01972450534777859, 1.0199203187250996, -0.015748356968139168, 1.0158730158730158, \
    -0.01178795575204224, 1.0118577075098814, -0.007843177461025893, 1.0078740157480315, \
    -0.003913899321136329, 1.003921568627451, 0.0, 1.0
__device__ const double g_ln_table[256] = {RC_LN_TABLE_VALUES};

// ln u for u in (0, 1]: u = 2^e f, f in [1/2, 1); ln f = ln c_k + log1p((f - c_k) / c_k) with the 128-entry (ln c, 1/c)
// table above (in LDS) and a degree-7 series on |r| <= 1/128.  A few ulp from libm.
__device__ __forceinline__ double ln_table(double u, const double* lntab) {
    const double f = __builtin_amdgcn_frexp_mant(u);
    const int ex = __builtin_amdgcn_frexp_exp(u);
    const int k = (int)((__double2hiint(f) >> 13) & 127);            // top 7 fraction bits
    const double ck = 0.5 + (double)(k + 1) * 0x1.0p-8;
    const double r = (f - ck) * lntab[2 * k + 1];                    // in [-1/128, 0)
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(r, p, 0.2);
    p = fma(r, p, -0.25);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    p = fma(r * r, p, r);                                            // log1p(r)
    return fma((double)ex, 6.93147180559945286227e-01, lntab[2 * k] + p);
}

// One thread per Box-Muller PAIR (counter): one Philox call, one log / sqrt, one sin/cos -> elements 2 ctr (cos) and
// 2 ctr + 1 (sin).  The three library calls are replaced by table-driven routines (LDS reads are cheap next to fp64
// VALU work): ln u through a 128-entry (ln c, 1/c) table + a degree-7 log1p series on |r| <= 1/128
// (c_127 = 1 exactly, so u -> 1 keeps full relative accuracy), sqrt through the v_rsq_f64 seed + one third-order
// step, sin/cos(2 pi u) through rc::sincos_table (64 u is exact).  Each agrees with libm to a few ulp
// (tests: |device - numpy| < 1e-15 on 0.05-scaled draws).  3.3x the throughput of the per-element libm version.
__global__ __launch_bounds__(256) void philox_normal_kernel(unsigned long long seed, unsigned long long offset,
                                                            long long n, double scale, double* out) {
    __shared__ __attribute__((aligned(16))) double sctab[128];
    __shared__ __attribute__((aligned(16))) double lntab[256];
    if (threadIdx.x < 64)
        reinterpret_cast<double2*>(sctab)[threadIdx.x] = reinterpret_cast<const double2*>(g_sincos_table)[threadIdx.x];
    if (threadIdx.x < 128)
        reinterpret_cast<double2*>(lntab)[threadIdx.x] = reinterpret_cast<const double2*>(g_ln_table)[threadIdx.x];
    __syncthreads();
    const unsigned long long first = offset >> 1;                        // first counter touched
    const unsigned long long last = (offset + (unsigned long long)n - 1) >> 1;
    const long long npairs = (long long)(last - first + 1);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < npairs; t += (long long)gridDim.x * 256) {
        const unsigned long long ctr = first + (unsigned long long)t;
        unsigned int w[4];
        philox4x32_10((unsigned int)ctr, (unsigned int)(ctr >> 32), 0u, 0u, (unsigned int)seed,
                      (unsigned int)(seed >> 32), w);
        const unsigned long long a = (((unsigned long long)w[1] << 32) | w[0]) >> 11;
        const unsigned long long b = (((unsigned long long)w[3] << 32) | w[2]) >> 11;
        const double u1 = ((double)a + 0.5) * 0x1.0p-53;           // (0, 1)
        const double u2 = ((double)b + 0.5) * 0x1.0p-53;
        const double lnu = ln_table(u1, lntab);
        double rad, rinv;
        rc::sqrt_rsqrt(-2.0 * lnu, rad, rinv);
        double sn, cs;
        rc::sincos_table(64.0 * u2, sctab, sn, cs);
        const double amp = scale * rad;
        const unsigned long long e0 = ctr << 1;
        if (e0 >= offset) out[e0 - offset] = amp * cs;
        if (e0 + 1 < offset + (unsigned long long)n && e0 + 1 >= offset) out[e0 + 1 - offset] = amp * sn;
    }
}

// ------------------------------------------------------------------------------------------------
// NumPy's legacy normal stream on the device (legacy_rng_core.h): the reference's RNG without the host
// ------------------------------------------------------------------------------------------------
// Stage 1 - raw MT19937 words.  raw[0 .. 624) holds a state block (the host's key, or the carry block of the previous
// segment); this kernel appends the following blocks.  The recurrence x[i] = next(x[i-624], x[i-623], x[i-227]) makes
// 227 consecutive words independent of each other and dependent on the chunk before: ONE wave walks the chunks (4 words
// per lane), the last 2048 words in an LDS ring, wave-level fences between chunks.  Sequential by nature, ~0.3 ns per
// word - an order of magnitude faster than NumPy's scalar generator on the host, and the words are born in HBM.
// LDS-only ordering point of ONE wave: the LDS operations of a wave execute in order, so draining the LDS counter is
// all that is needed between a chunk's writes and the next chunk's reads.  (A full release/acquire fence would also
// wait for the chunk's GLOBAL stores - hundreds of cycles per chunk on a purely sequential kernel.)
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// One chunk of the recurrence for a wave: ring index `c` (this lane's first word), the 12 LDS reads issued together.
__device__ __forceinline__ void mt_chunk(const unsigned int* ring, int mask, int c, unsigned int (&v)[4]) {
    unsigned int a[4], b[4], m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = c + 64 * j;
        a[j] = ring[(i - 624) & mask];
        b[j] = ring[(i - 623) & mask];
        m[j] = ring[(i - 227) & mask];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = rcl::mt_next_word(a[j], b[j], m[j]);
}

// Sub-stream p (one wave per workgroup, P workgroups in parallel) starts from the 624-word window seeds[p] - the
// generator's window at global word p * kMtJumpWords (seeds[0] = the caller's block) - and writes the kMtJumpWords words
// that FOLLOW its window, raw[624 + p B ... 624 + (p + 1) B): the concatenation over p is the sequential stream.
// `raw` must hold `total` words (a multiple of 624); the last sub-stream stops there.
__global__ __launch_bounds__(64) void mt19937_raw_kernel(const unsigned int* seeds, unsigned int* raw, long long total) {
    __shared__ unsigned int ring[2048];
    const int lane = threadIdx.x;
    const long long p = blockIdx.x;
    const unsigned int* seed = seeds + p * rcl::kMtN;
    for (int i = lane; i < rcl::kMtN; i += 64) {
        ring[i] = seed[i];
        if (p == 0) raw[i] = seed[i];
    }
    wave_lds_fence();
    // this sub-stream's share of the `total` words of the segment (the last one may be short)
    long long mine = total - rcl::kMtN - p * kMtJumpWords;
    mine = mine < 0 ? 0 : (mine > kMtJumpWords ? kMtJumpWords : mine);
    const long long full = mine / rcl::kMtChunk;
    const int rest = (int)(mine - full * rcl::kMtChunk);
    const bool tail = lane + 192 < rcl::kMtChunk;
    unsigned int* dst = raw + rcl::kMtN + p * kMtJumpWords + lane;
    int c = rcl::kMtN + lane;                      // ring index (mod 2048) of this lane's first word of the chunk
    for (long long n = 0; n < full; ++n) {
        unsigned int v[4];
        mt_chunk(ring, 2047, c, v);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            ring[(c + 64 * j) & 2047] = v[j];
            dst[64 * j] = v[j];                    // fire and forget: nothing in this kernel reads `raw` back
        }
        if (tail) {
            ring[(c + 192) & 2047] = v[3];
            dst[192] = v[3];
        }
        wave_lds_fence();                          // this chunk's words are visible to the next chunk's reads
        c = (c + rcl::kMtChunk) & 2047;
        dst += rcl::kMtChunk;
    }
    if (rest) {                                    // last, partial chunk of the sub-stream
        unsigned int v[4];
        mt_chunk(ring, 2047, c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < rest) dst[64 * j] = v[j];
    }
}

// Start windows of the sub-streams by jump-ahead (scripts/mt_jump_poly.py, mt19937_jump_poly.h): with g(x) = x^B mod
// phi(x), the window at distance B is  window_B[j] = XOR over the set coefficients i of g of  x[i + j].  One launch per
// jump (window p from window p - 1), kJumpWgs workgroups of it: each regenerates the 19937 + 624 words behind the old
// window into its LDS (wave 0, 88 chunks, ~10 us) and produces 78 of the 624 new words, its ~9900 XOR terms per word
// split over three thread groups (consecutive lanes read consecutive LDS words: conflict-free; the LDS read rate of a
// CU is the limit, hence several CUs).  Word 0 of a jumped window is exact only in its top bit - the only bit of it
// the recurrence uses; as an OUTPUT that word belongs to the sub-stream before.
constexpr int kJumpSeq = 19937 + rcl::kMtN;        // words of the stream a jump needs
constexpr int kJumpSeqPad = 20736;                 // >= kJumpSeq + 256 (whole chunks), LDS words
constexpr int kJumpWgs = 8;
constexpr int kJumpWords = rcl::kMtN / kJumpWgs;   // 78 window words per workgroup
constexpr int kJumpGroups = 3;                     // term groups per word
constexpr int kJumpThreads = 256;
constexpr int kJumpLdsWords = kJumpSeqPad + kJumpGroups * kJumpWords;
static_assert(kJumpWords * kJumpWgs == rcl::kMtN && kJumpGroups * kJumpWords <= kJumpThreads, "jump geometry");
__device__ const unsigned short g_mt_jump_idx[kMtJumpTerms] = {RC_MT_JUMP_IDX_VALUES};

__global__ __launch_bounds__(kJumpThreads) void mt19937_jump_step_kernel(unsigned int* seeds, int p) {
    extern __shared__ unsigned int xs[];           // kJumpLdsWords words
    unsigned int* part = xs + kJumpSeqPad;
    const int t = threadIdx.x;
    const unsigned int* prev = seeds + (long long)(p - 1) * rcl::kMtN;
    for (int i = t; i < rcl::kMtN; i += kJumpThreads) xs[i] = prev[i];
    __syncthreads();
    if (t < 64) {                                  // wave 0: the stream after the old window
        const bool tail = t + 192 < rcl::kMtChunk;
        for (int c = rcl::kMtN + t; c - t < kJumpSeq; c += rcl::kMtChunk) {
            unsigned int v[4];
            mt_chunk(xs, 0xffff, c, v);            // flat array (indices < 65536): no wrap-around
#pragma unroll
            for (int j = 0; j < 3; ++j) xs[c + 64 * j] = v[j];
            if (tail) xs[c + 192] = v[3];
            wave_lds_fence();
        }
    }
    __syncthreads();
    const int grp = t / kJumpWords, wj = t - grp * kJumpWords;
    if (grp < kJumpGroups) {
        const unsigned int* base = xs + blockIdx.x * kJumpWords + wj;
        unsigned int acc = 0;
        int k = grp;
        for (; k + 7 * kJumpGroups < kMtJumpTerms; k += 8 * kJumpGroups) {
            unsigned int w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = base[g_mt_jump_idx[k + u * kJumpGroups]];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= w[u];
        }
        for (; k < kMtJumpTerms; k += kJumpGroups) acc ^= base[g_mt_jump_idx[k]];
        part[grp * kJumpWords + wj] = acc;
    }
    __syncthreads();
    if (t < kJumpWords) {
        unsigned int acc = part[t];
#pragma unroll
        for (int g = 1; g < kJumpGroups; ++g) acc ^= part[g * kJumpWords + t];
        seeds[(long long)p * rcl::kMtN + blockIdx.x * kJumpWords + t] = acc;
    }
}

// Stage 2 - polar-method attempts.  Attempt t reads raw words [w0 + 4t, w0 + 4t + 4) of the segment; a workgroup owns
// kLgAttempts consecutive attempts (8 per thread).  Pass A counts the accepted attempts per workgroup, a one-block
// scan turns the counts into ranks, pass B recomputes the attempts and writes the two normals of accepted attempt number
// r (counted over the WHOLE stream) to stream elements e_shift + 2r (f x2) and e_shift + 2r + 1 (f x1), mapped through
// the period / skip / scale pattern of rcl::stream_slot.
constexpr int kLgThreads = 256;
constexpr int kLgPerThread = 8;
constexpr int kLgAttempts = kLgThreads * kLgPerThread;

struct LegacyParams {
    const unsigned int* raw;          // segment words; raw[0] is global word g0
    long long w_first;                // index INTO raw of the first word of attempt t_first
    long long t_first, t_count;       // attempts [t_first, t_first + t_count) are processed by this launch
    long long rank_base;              // accepted attempts before t_first
    long long pairs_needed;           // accepted attempts to emit in all
    long long e_shift, n_total;       // stream elements in front of the first generated one (0 | 1); total wanted
    long long period, skip;
    const double* scales;             // [n_periods] device
    double* out;
    unsigned long long* wg_counts;    // [nwg + 1]
    long long* last;                  // [0] attempt index of the last needed pair, [1..4] its raw words
};

__device__ __forceinline__ bool legacy_attempt(const LegacyParams& p, long long t, double& x1, double& x2, double& r2,
                                               unsigned int (&w)[4]) {
    const unsigned int* src = p.raw + p.w_first + 4 * (t - p.t_first);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = src[i];
    return rcl::polar_attempt(w[0], w[1], w[2], w[3], x1, x2, r2);
}

__global__ __launch_bounds__(kLgThreads) void legacy_count_kernel(const LegacyParams p) {
    __shared__ unsigned int wsum[kLgThreads / 64];
    const long long base = p.t_first + (long long)blockIdx.x * kLgAttempts + (long long)threadIdx.x * kLgPerThread;
    unsigned int n = 0;
#pragma unroll
    for (int j = 0; j < kLgPerThread; ++j) {
        const long long t = base + j;
        if (t < p.t_first + p.t_count) {
            double x1, x2, r2;
            unsigned int w[4];
            n += legacy_attempt(p, t, x1, x2, r2, w) ? 1u : 0u;
        }
    }
    n = wave_allsum(n);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) p.wg_counts[blockIdx.x] = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive prefix sum of counts[0 .. n) in place, total to counts[n]; one workgroup (n is a few 10^4 at most)
__global__ __launch_bounds__(1024) void legacy_scan_kernel(unsigned long long* counts, long long n) {
    __shared__ unsigned long long part[1024];
    const long long per = (n + 1023) / 1024;
    const long long lo = (long long)threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    unsigned long long s = 0;
    for (long long i = lo; i < hi; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned long long v = part[i];
            part[i] = run;
            run += v;
        }
        counts[n] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (long long i = lo; i < hi; ++i) {
        const unsigned long long v = counts[i];
        counts[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(kLgThreads) void legacy_emit_kernel(const LegacyParams p) {
    __shared__ unsigned int wsum[kLgThreads / 64];
    __shared__ __attribute__((aligned(16))) double lntab[256];
    if (threadIdx.x < 128)
        reinterpret_cast<double2*>(lntab)[threadIdx.x] = reinterpret_cast<const double2*>(g_ln_table)[threadIdx.x];
    const long long base = p.t_first + (long long)blockIdx.x * kLgAttempts + (long long)threadIdx.x * kLgPerThread;
    const long long t_end = p.t_first + p.t_count;
    // thread-local count, then the thread's exclusive offset inside the workgroup
    unsigned int mask = 0, n = 0;
#pragma unroll
    for (int j = 0; j < kLgPerThread; ++j) {
        const long long t = base + j;
        if (t < t_end) {
            double x1, x2, r2;
            unsigned int w[4];
            if (legacy_attempt(p, t, x1, x2, r2, w)) {
                mask |= 1u << j;
                ++n;
            }
        }
    }
    unsigned int incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += o;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned int wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += wsum[w];
    long long rank = p.rank_base + (long long)p.wg_counts[blockIdx.x] + wave_off + (incl - n);
#pragma unroll
    for (int j = 0; j < kLgPerThread; ++j) {
        if (!((mask >> j) & 1u)) continue;
        const long long t = base + j;
        if (rank < p.pairs_needed) {
            double x1, x2, r2;
            unsigned int w[4];
            legacy_attempt(p, t, x1, x2, r2, w);
            const double f = __dsqrt_rn(__ddiv_rn(rcl::mul_rn(-2.0, ln_table(r2, lntab)), r2));
            const double val[2] = {rcl::mul_rn(f, x2), rcl::mul_rn(f, x1)};         // returned first, cached second
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long long e = p.e_shift + 2 * rank + h;
                if (e < p.n_total) {
                    long long pi;
                    const long long slot = rcl::stream_slot(e, p.period, p.skip, &pi);
                    if (slot >= 0) p.out[slot] = rcl::add_rn(0.0, rcl::mul_rn(p.scales[pi], val[h]));   // loc + scale * g
                }
            }
            if (rank == p.pairs_needed - 1) {
                p.last[0] = t;
#pragma unroll
                for (int i = 0; i < 4; ++i) p.last[1 + i] = (long long)w[i];
            }
        }
        ++rank;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
std::mutex g_cfg_mu;                 // guards g_default_kernel only
int g_default_kernel = RC_KERNEL_AUTO;
long long* g_stamps = nullptr;      // diagnostic builds only

// Per-device state.  The C ABI takes a `device` argument everywhere, so nothing here is process-wide: the blocking
// entry points serialise PER DEVICE (two devices run concurrently, e.g. from the threads of
// rc_mc_metrics_sharded_f64), and the kernel attributes that must be raised before a launch
// (hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property) are tracked per device.
constexpr int kMaxDevices = 64;
enum { kAttrExpm = 0, kAttrAnyN, kAttrSortMerge, kAttrSortChunk, kAttrMtJump, kAttrCount };
struct DeviceCtx {
    std::mutex mu;                   // blocking entry points: one at a time per device (they share `stream` and `ws`)
    hipStream_t stream = nullptr;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    std::atomic<bool> attr[kAttrCount];
    DeviceCtx() {
        for (auto& a : attr) a.store(false);
    }
};
DeviceCtx g_ctx[kMaxDevices];

int device_in_range(int device) {
    int n = 0;
    RC_HIP_CHECK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n || device >= kMaxDevices) return fail(RC_EINVAL, "device index out of range");
    return RC_OK;
}

// caller holds g_ctx[device].mu
int get_ctx(int device, DeviceCtx** out) {
    if (int rc = device_in_range(device)) return rc;
    RC_HIP_CHECK(hipSetDevice(device));
    if (!g_ctx[device].stream) RC_HIP_CHECK(hipStreamCreateWithFlags(&g_ctx[device].stream, hipStreamNonBlocking));
    *out = &g_ctx[device];
    return RC_OK;
}

// Raises a kernel's dynamic-LDS limit once per DEVICE (the current one).  Racing callers may both set it: idempotent.
int ensure_func_attr(int which, const void* func, int bytes) {
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(RC_EINVAL, "device index out of range");
    if (g_ctx[dev].attr[which].load(std::memory_order_acquire)) return RC_OK;
    RC_HIP_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_ctx[dev].attr[which].store(true, std::memory_order_release);
    return RC_OK;
}

int ensure_ws(DeviceCtx* ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return RC_OK;
    if (ctx->ws) RC_HIP_CHECK(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    RC_HIP_CHECK(hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return RC_OK;
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // plain host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice;
}

int check_common(int N, int in, int out, long long C, long long K) {
    if (N < 2 || N > RC_MAX_NSPIN) return fail(RC_EINVAL, "N must be in [2, 32]");
    if (in < 0 || in >= N || out < 0 || out >= N) return fail(RC_EINVAL, "in/out spin index out of range");
    if (C < 0 || K < 0) return fail(RC_EINVAL, "C and K must be non-negative");
    return RC_OK;
}

template <int N, int MODE>
int launch_chain(hipStream_t s, const FidParams& p) {
    const long long blocks = p.ntiles;
    if (blocks > 0x7fffffffLL) return fail(RC_EINVAL, "too many tiles for one launch");
    hipLaunchKernelGGL((mc_fid_chain_kernel<N, MODE>), dim3((unsigned)blocks), dim3(64), 0, s, p);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int enqueue_expm(hipStream_t s, int N, int in, int out, const double* h0_diag, const double* h0_offdiag, int ring,
                 const double* ctrl, const double* draws, long long draw_cstride, const double* diag_imag,
                 long long imag_cstride, long long C, long long K, double* fid) {
    if (N > RC_MAX_NSPIN_FAST) return fail(RC_EINVAL, "the dense kernels (ring topology, expm) support N <= 16");
    ExpmParams p{};
    p.ctrl = ctrl;
    p.draws = draws;
    p.diag_imag = diag_imag;
    p.fid = fid;
    p.C = C;
    p.K = K;
    p.draw_cstride = draw_cstride;
    p.imag_cstride = imag_cstride;
    p.N = N;
    p.in = in;
    p.out = out;
    p.ring = ring ? 1 : 0;
    for (int i = 0; i < RC_MAX_NSPIN; ++i) {
        p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
        p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
    }
    const size_t lds = (size_t)kExpmWaves * kExpmBufs * N * N * sizeof(cplx);
    if (int rc = ensure_func_attr(kAttrExpm, (const void*)mc_fid_expm_kernel,
                                  kExpmWaves * kExpmBufs * RC_MAX_NSPIN_FAST * RC_MAX_NSPIN_FAST * (int)sizeof(cplx)))
        return rc;
    const long long total = C * K;
    long long blocks = (total + kExpmWaves - 1) / kExpmWaves;
    if (blocks > 256LL * 16) blocks = 256LL * 16;              // grid-stride loop inside; every wave exits
    hipLaunchKernelGGL(mc_fid_expm_kernel, dim3((unsigned)blocks), dim3(64 * kExpmWaves), lds, s, p);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int enqueue_fidelity(hipStream_t s, int kernel, int N, int in, int out, const double* h0_diag,
                     const double* h0_offdiag, int ring, const double* ctrl, const double* draws,
                     long long draw_cstride, long long C, long long K, double* fid) {
    if (draw_cstride < 0) draw_cstride = K * N * 3;          // default: private draws per controller
    if (draw_cstride != 0 && draw_cstride < K * N * 3) return fail(RC_EINVAL, "draws_ctrl_stride overlaps controllers");
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (C == 0 || K == 0) return RC_OK;
    if (!ctrl || !draws || !fid) return fail(RC_EINVAL, "NULL array pointer");
    const bool ends = (in == 0 && out == N - 1) || (in == N - 1 && out == 0);
    if (kernel == RC_KERNEL_AUTO) kernel = ring ? RC_KERNEL_JACOBI : RC_KERNEL_TRIDIAG_ADJ;
    if (kernel == RC_KERNEL_TRIDIAG_QL || kernel == RC_KERNEL_TRIDIAG_ADJ) {
        const int mode = (kernel == RC_KERNEL_TRIDIAG_QL) ? rc::kWeightsRows
                                                          : (ends ? rc::kWeightsEnds : rc::kWeightsAdjugate);
        if (ring) return fail(RC_EINVAL, "the tridiagonal QL kernel handles chain topology only");
        FidParams p{};
        p.ctrl = ctrl;
        p.draws = draws;
        p.fid = fid;
        p.C = C;
        p.K = K;
        p.draw_cstride = draw_cstride;
        p.tiles_per_ctrl = (K + 63) / 64;
        p.ntiles = C * p.tiles_per_ctrl;
        p.in = in;
        p.out = out;
        p.align16 = (((uintptr_t)draws & 15) == 0 && (((size_t)draw_cstride * 8) & 15) == 0) ? 1 : 0;
        p.stamps = g_stamps;
        for (int i = 0; i < RC_MAX_NSPIN; ++i) {
            p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
            p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
        }
        if (N > RC_MAX_NSPIN_FAST) {
            const long long blocks = p.ntiles;
            if (blocks > 0x7fffffffLL) return fail(RC_EINVAL, "too many tiles for one launch");
            const size_t lds = (size_t)4 * N * 64 * sizeof(double);
            if (int rc = ensure_func_attr(kAttrAnyN, (const void*)mc_fid_chain_anyn_kernel, 4 * RC_MAX_NSPIN * 64 * (int)sizeof(double)))
                return rc;
            hipLaunchKernelGGL(mc_fid_chain_anyn_kernel, dim3((unsigned)blocks), dim3(64), lds, s, p, N);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
        switch (N) {
#define RC_CASE(n)                                                                        \
    case n:                                                                               \
        return mode == rc::kWeightsRows ? launch_chain<n, rc::kWeightsRows>(s, p)         \
               : (mode == rc::kWeightsEnds ? launch_chain<n, rc::kWeightsEnds>(s, p)      \
                                           : launch_chain<n, rc::kWeightsAdjugate>(s, p));
            RC_CASE(2) RC_CASE(3) RC_CASE(4) RC_CASE(5) RC_CASE(6) RC_CASE(7) RC_CASE(8) RC_CASE(9)
            RC_CASE(10) RC_CASE(11) RC_CASE(12) RC_CASE(13) RC_CASE(14) RC_CASE(15) RC_CASE(16)
#undef RC_CASE
        }
        return fail(RC_EINVAL, "unsupported N");
    }
    if (N > RC_MAX_NSPIN_FAST) return fail(RC_EINVAL, "the dense kernels (ring topology, expm) support N <= 16");
    if (kernel == RC_KERNEL_EXPM)
        return enqueue_expm(s, N, in, out, h0_diag, h0_offdiag, ring, ctrl, draws, draw_cstride, nullptr, 0, C, K, fid);
    if (kernel == RC_KERNEL_JACOBI) {
        JacParams p{};
        p.ctrl = ctrl;
        p.draws = draws;
        p.fid = fid;
        p.C = C;
        p.K = K;
        p.draw_cstride = draw_cstride;
        p.N = N;
        p.in = in;
        p.out = out;
        p.ring = ring ? 1 : 0;
        for (int i = 0; i < RC_MAX_NSPIN; ++i) {
            p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
            p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
        }
        const long long total = C * K;
        const int spw = (N <= 8) ? 8 : 4;                       // samples per wave
        long long blocks = (total + kJacWaves * spw - 1) / (kJacWaves * spw);
        if (blocks > 256LL * 16) blocks = 256LL * 16;          // grid-stride loop inside; every wave exits
        if (N <= 8)
            hipLaunchKernelGGL((mc_fid_jacobi_kernel<8, 8>), dim3((unsigned)blocks), dim3(64 * kJacWaves), 0, s, p);
        else
            hipLaunchKernelGGL((mc_fid_jacobi_kernel<16, 16>), dim3((unsigned)blocks), dim3(64 * kJacWaves), 0, s, p);
        RC_HIP_CHECK(hipGetLastError());
        return RC_OK;
    }
    return fail(RC_EINVAL, "unknown kernel id");
}

int enqueue_reduce(hipStream_t s, const double* fid, long long C, long long K, const double* thr, int nq,
                   double eps, double* rim1, double* stdv, double* minf, double* q, double* sorted_out) {
    if (C < 0 || K < 0) return fail(RC_EINVAL, "C and K must be non-negative");
    if (nq < 0 || nq > kMaxQ) return fail(RC_EINVAL, "nq must be in [0, 8]");
    if (nq > 0 && !thr) return fail(RC_EINVAL, "q_thresholds is NULL");
    if (C == 0) return RC_OK;
    if (K == 0) return fail(RC_EINVAL, "K must be positive for a reduction");
    if (!fid) return fail(RC_EINVAL, "NULL fid pointer");
    if (C > 0x7fffffffLL) return fail(RC_EINVAL, "too many controllers for one launch");
    RedParams p{};
    p.fid = fid;
    p.C = C;
    p.K = K;
    p.nq = nq;
    for (int j = 0; j < nq; ++j) p.thr[j] = thr[j];
    p.eps = eps;
    p.rim1 = rim1;
    p.stdv = stdv;
    p.minf = minf;
    p.q = q;
    if (rim1 || stdv || minf || q) {
        if (K <= kWaveRowMaxK && C >= 64) {                   // many short rows: one wave per row
            const dim3 grid((unsigned)((C + 3) / 4));
            if (nq == 0)
                hipLaunchKernelGGL(reduce_rows_wave_kernel<0>, grid, dim3(256), 0, s, p);
            else if (nq <= 2)
                hipLaunchKernelGGL(reduce_rows_wave_kernel<2>, grid, dim3(256), 0, s, p);
            else
                hipLaunchKernelGGL(reduce_rows_wave_kernel<kMaxQ>, grid, dim3(256), 0, s, p);
        } else if (nq == 0)
            hipLaunchKernelGGL(reduce_kernel<0>, dim3((unsigned)C), dim3(kRedThreads), 0, s, p);
        else if (nq <= 2)
            hipLaunchKernelGGL(reduce_kernel<2>, dim3((unsigned)C), dim3(kRedThreads), 0, s, p);
        else
            hipLaunchKernelGGL(reduce_kernel<kMaxQ>, dim3((unsigned)C), dim3(kRedThreads), 0, s, p);
        RC_HIP_CHECK(hipGetLastError());
    }
    if (sorted_out) {
        long long P = 2;
        while (P < K) P <<= 1;
        if (C * P > (1LL << 34)) return fail(RC_EINVAL, "sorted_out: workspace would exceed 128 GiB");
        if (K <= kSortChunk) {                                        // one fused launch, no workspace: merge sort
            if (int rc = ensure_func_attr(kAttrSortMerge, (const void*)sort_rows_merge_kernel, (kSortChunk / 16 * 17 + 1) * 8))
                return rc;
            const int T = (int)((K + 15) / 16), n = 16 * T;
            const int threads = ((T + 63) / 64) * 64;
            hipLaunchKernelGGL(sort_rows_merge_kernel, dim3((unsigned)C), dim3(threads), (size_t)(n / 16 * 17 + 1) * sizeof(double),
                               s, fid, sorted_out, K, n);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
        // workspace of THIS call, allocated and released in stream order (two streams sorting long rows at the same time
        // each get their own; nothing is shared between calls)
        struct SortWs {
            double* work = nullptr;
            int* flags = nullptr;
        } wsv, *ws = &wsv;
        RC_HIP_CHECK(hipMallocAsync((void**)&wsv.work, (size_t)C * P * sizeof(double), s));
        if (hipError_t e = hipMallocAsync((void**)&wsv.flags, (size_t)C * sizeof(int), s); e != hipSuccess) {
            (void)hipFreeAsync(wsv.work, s);
            return fail(RC_EHIP, std::string("hipMallocAsync(sort flags): ") + hipGetErrorString(e));
        }
        struct Release {
            SortWs* w;
            hipStream_t st;
            ~Release() {
                (void)hipFreeAsync(w->work, st);
                (void)hipFreeAsync(w->flags, st);
            }
        } release{ws, s};
        RC_HIP_CHECK(hipMemsetAsync(ws->flags, 0, (size_t)C * sizeof(int), s));
        {
            // long rows (K > 16384): chunk sort -> for each larger size one fused pass over the strides >= 16384 (up to four per
            // launch) and one chunk pass over the rest; the first pass reads `fid`, the last writes `sorted_out`
            if (int rc = ensure_func_attr(kAttrSortChunk, (const void*)sort_chunk16_kernel, kSortChunk / 16 * 17 * 8)) return rc;
            const size_t lds16 = (size_t)(kSortChunk / 16 * 17) * sizeof(double);
            const dim3 cgrid((unsigned)C, (unsigned)(P / kSortChunk));
            hipLaunchKernelGGL(sort_chunk16_kernel, cgrid, dim3(kSortThreads), lds16, s, fid, ws->work, sorted_out, ws->flags,
                               K, P, 2LL, (long long)kSortChunk, 1, 0);
            for (long long size = 2LL * kSortChunk; size <= P; size <<= 1) {
                long long hi = size >> 1;
                while (hi >= kSortChunk) {
                    int nleft = 0;
                    for (long long x = hi; x >= kSortChunk; x >>= 1) ++nleft;
                    const int take = nleft >= 4 ? 4 : nleft;
                    const long long S = hi >> (take - 1);
                    const dim3 ggrid((unsigned)C, 128);
                    switch (take) {
                        case 4: hipLaunchKernelGGL(sort_global_fused_kernel<4>, ggrid, dim3(256), 0, s, ws->work, P, size, S); break;
                        case 3: hipLaunchKernelGGL(sort_global_fused_kernel<3>, ggrid, dim3(256), 0, s, ws->work, P, size, S); break;
                        case 2: hipLaunchKernelGGL(sort_global_fused_kernel<2>, ggrid, dim3(256), 0, s, ws->work, P, size, S); break;
                        default: hipLaunchKernelGGL(sort_global_fused_kernel<1>, ggrid, dim3(256), 0, s, ws->work, P, size, S); break;
                    }
                    hi = S >> 1;
                }
                hipLaunchKernelGGL(sort_chunk16_kernel, cgrid, dim3(kSortThreads), lds16, s, fid, ws->work, sorted_out,
                                   ws->flags, K, P, size, size, 0, size == P ? 1 : 0);
            }
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
        RC_HIP_CHECK(hipGetLastError());
    }
    return RC_OK;
}

// ------------------------------------------------------------------------------------------------
// single-process multi-device driver (rc_mc_fidelity_sharded_f64 / rc_mc_metrics_sharded_f64)
// ------------------------------------------------------------------------------------------------
// The (controller x perturbation) sample space is split by CONTROLLER into contiguous balanced blocks, one per device
// (per-controller vectors and reductions stay device-local; the reference's only parallel construct is
// the dead Pool at mcsim.py:451-455).  One host thread per device: its controllers are processed in chunks of at
// most kShardChunkBytes of draws through the device's workspace - H2D of the chunk's draws (or Philox generation on the
// device), fidelity kernel, reduction, D2H of fidelities / metric rows straight into the caller's host arrays.  No
// collective is needed for a host-resident result.
constexpr size_t kShardChunkBytes = (size_t)4 << 30;

struct ShardJob {
    int device, kernel, N, in, out, ring;
    const double *h0d, *h0o, *ctrl, *draws;          // host; draws may be null (Philox)
    unsigned long long seed, offset;
    double sigma;
    long long C, K, c0, c1;                          // this device owns controllers [c0, c1)
    const double* thr;
    int nq;
    double eps;
    double *rim1, *stdv, *minf, *q, *fid_out;        // host, FULL arrays ([3][C], [3][nq][C], [C][K]); may be null
    int rc = RC_OK;
    std::string err;
};

int run_shard_locked(ShardJob* j) {
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(j->device, &ctx)) return rc;
    const long long G = 3LL * j->N, K = j->K, Cl = j->c1 - j->c0;
    if (Cl <= 0 || K == 0) return RC_OK;
    const bool want_red = j->rim1 || j->stdv || j->minf || (j->q && j->nq);
    long long cc_max = (long long)(kShardChunkBytes / ((size_t)K * G * sizeof(double)));
    if (cc_max < 1) cc_max = 1;
    if (cc_max > Cl) cc_max = Cl;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nb_ctrl = up((size_t)cc_max * (j->N + 1) * sizeof(double));
    const size_t nb_draw = up((size_t)cc_max * K * G * sizeof(double));
    const size_t nb_fid = up((size_t)cc_max * K * sizeof(double));
    const size_t nb_c3 = up((size_t)3 * cc_max * sizeof(double));
    const size_t nb_q = up((size_t)3 * (j->nq > 0 ? j->nq : 1) * cc_max * sizeof(double));
    if (int rc = ensure_ws(ctx, nb_ctrl + nb_draw + nb_fid + 3 * nb_c3 + nb_q)) return rc;
    char* w = (char*)ctx->ws;
    double* d_ctrl = (double*)w; w += nb_ctrl;
    double* d_draw = (double*)w; w += nb_draw;
    double* d_fid = (double*)w;  w += nb_fid;
    double* d_rim = (double*)w;  w += nb_c3;
    double* d_std = (double*)w;  w += nb_c3;
    double* d_min = (double*)w;  w += nb_c3;
    double* d_q = (double*)w;
    hipStream_t st = ctx->stream;
    for (long long a = j->c0; a < j->c1; a += cc_max) {
        const long long cc = (j->c1 - a < cc_max) ? (j->c1 - a) : cc_max;
        RC_HIP_CHECK(hipMemcpyAsync(d_ctrl, j->ctrl + a * (j->N + 1), (size_t)cc * (j->N + 1) * sizeof(double),
                                    hipMemcpyHostToDevice, st));
        if (j->draws) {
            RC_HIP_CHECK(hipMemcpyAsync(d_draw, j->draws + a * K * G, (size_t)cc * K * G * sizeof(double),
                                        hipMemcpyHostToDevice, st));
        } else {
            // element ((c K + k) N + i) 3 + slot of the stream: independent of how the controllers are sharded
            if (int rc = rc_draws_philox_f64_async(j->device, st, j->seed, j->offset + (unsigned long long)(a * K * G),
                                                   cc * K * G, j->sigma, d_draw))
                return rc;
        }
        if (int rc = enqueue_fidelity(st, j->kernel, j->N, j->in, j->out, j->h0d, j->h0o, j->ring, d_ctrl, d_draw, -1, cc, K,
                                      d_fid))
            return rc;
        if (want_red) {
            if (int rc = enqueue_reduce(st, d_fid, cc, K, j->thr, j->nq, j->eps, j->rim1 ? d_rim : nullptr,
                                        j->stdv ? d_std : nullptr, j->minf ? d_min : nullptr,
                                        (j->q && j->nq) ? d_q : nullptr, nullptr))
                return rc;
            // device rows [3][cc] -> columns [a, a+cc) of the caller's [3][C]
            const size_t wbytes = (size_t)cc * sizeof(double), dpitch = (size_t)j->C * sizeof(double);
            if (j->rim1) RC_HIP_CHECK(hipMemcpy2DAsync(j->rim1 + a, dpitch, d_rim, wbytes, wbytes, 3, hipMemcpyDeviceToHost, st));
            if (j->stdv) RC_HIP_CHECK(hipMemcpy2DAsync(j->stdv + a, dpitch, d_std, wbytes, wbytes, 3, hipMemcpyDeviceToHost, st));
            if (j->minf) RC_HIP_CHECK(hipMemcpy2DAsync(j->minf + a, dpitch, d_min, wbytes, wbytes, 3, hipMemcpyDeviceToHost, st));
            if (j->q && j->nq)
                RC_HIP_CHECK(hipMemcpy2DAsync(j->q + a, dpitch, d_q, wbytes, wbytes, (size_t)3 * j->nq, hipMemcpyDeviceToHost, st));
        }
        if (j->fid_out)
            RC_HIP_CHECK(hipMemcpyAsync(j->fid_out + a * K, d_fid, (size_t)cc * K * sizeof(double), hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipStreamSynchronize(st));       // the workspace is reused by the next chunk
    }
    return RC_OK;
}

void run_shard(ShardJob* j) {
    std::lock_guard<std::mutex> lk(g_ctx[j->device].mu);
    j->rc = run_shard_locked(j);
    if (j->rc) j->err = g_last_error;            // thread-local of THIS worker: hand it to the caller
}

int run_sharded(int ndev, const int* devices, ShardJob proto) {
    if (int rc = check_common(proto.N, proto.in, proto.out, proto.C, proto.K)) return rc;
    if (ndev < 1 || ndev > kMaxDevices) return fail(RC_EINVAL, "ndev must be in [1, 64]");
    if (proto.nq < 0 || proto.nq > kMaxQ) return fail(RC_EINVAL, "nq must be in [0, 8]");
    if (proto.nq > 0 && !proto.thr) return fail(RC_EINVAL, "q_thresholds is NULL");
    if (proto.C == 0 || proto.K == 0) return RC_OK;
    if (!proto.ctrl) return fail(RC_EINVAL, "NULL controllers pointer");
    if (!proto.fid_out && !proto.rim1 && !proto.stdv && !proto.minf && !(proto.q && proto.nq))
        return fail(RC_EINVAL, "no output requested");
    for (int r = 0; r < ndev; ++r) {
        const int dev = devices ? devices[r] : r;
        if (int rc = device_in_range(dev)) return rc;
        for (int r2 = 0; r2 < r; ++r2)
            if ((devices ? devices[r2] : r2) == dev) return fail(RC_EINVAL, "a device is listed twice");
    }
    std::vector<ShardJob> jobs(ndev, proto);
    const long long base = proto.C / ndev, extra = proto.C % ndev;
    long long start = 0;
    for (int r = 0; r < ndev; ++r) {
        jobs[r].device = devices ? devices[r] : r;
        jobs[r].c0 = start;
        start += base + (r < extra ? 1 : 0);
        jobs[r].c1 = start;
    }
    std::vector<std::thread> th;
    for (int r = 1; r < ndev; ++r) th.emplace_back(run_shard, &jobs[r]);
    run_shard(&jobs[0]);
    for (auto& t : th) t.join();
    for (int r = 0; r < ndev; ++r)
        if (jobs[r].rc) return fail(jobs[r].rc, "device " + std::to_string(jobs[r].device) + ": " + jobs[r].err);
    return RC_OK;
}

// ------------------------------------------------------------------------------------------------
// driver of the device-side legacy stream (rc_draws_legacy_f64)
// ------------------------------------------------------------------------------------------------
// The stream is produced in SEGMENTS of at most kLegacySegWords raw words: block-aligned word range [g0, g1) of the
// generator's output sequence, the first block being known (the caller's key / the previous segment's last block).
// After every segment the accepted-attempt count comes back to the host, which decides whether more words are needed.
constexpr long long kLegacySegWords = 1LL << 27;      // 512 MiB of raw words per segment

struct StreamFree {
    void* p;
    hipStream_t s;
    ~StreamFree() {
        if (p) (void)hipFreeAsync(p, s);
    }
};

int legacy_normal_stream(hipStream_t st, rc_mt19937_state* state, long long n_periods, long long period, long long skip,
                         const double* scales_host, double* out_dev) {
    const long long n_total = n_periods * period;
    const long long e_shift = state->has_gauss ? 1 : 0;
    const long long n_need = n_total - e_shift;
    const long long pairs = (n_need + 1) / 2;
    double* d_scales = nullptr;
    RC_HIP_CHECK(hipMallocAsync((void**)&d_scales, (size_t)n_periods * sizeof(double), st));
    StreamFree free_scales{d_scales, st};
    RC_HIP_CHECK(hipMemcpyAsync(d_scales, scales_host, (size_t)n_periods * sizeof(double), hipMemcpyHostToDevice, st));
    if (e_shift) {                                    // element 0 of the stream is the host's cached normal
        long long pi;
        const long long slot = rcl::stream_slot(0, period, skip, &pi);
        if (slot >= 0) {
            const double v = 0.0 + scales_host[pi] * state->gauss;
            RC_HIP_CHECK(hipMemcpyAsync(out_dev + slot, &v, sizeof(double), hipMemcpyHostToDevice, st));
            RC_HIP_CHECK(hipStreamSynchronize(st));   // `v` is a stack temporary
        }
        state->has_gauss = 0;
        state->gauss = 0.0;
    }
    if (pairs == 0) return RC_OK;

    long long* d_last = nullptr;
    RC_HIP_CHECK(hipMallocAsync((void**)&d_last, 5 * sizeof(long long), st));
    StreamFree free_last{d_last, st};
    RC_HIP_CHECK(hipMemsetAsync(d_last, 0xff, 5 * sizeof(long long), st));

    // global word coordinates: word 0 = key[0] of the caller's state; the next unread word is `pos`
    std::vector<unsigned int> carry(state->key, state->key + rcl::kMtN);   // block at g0
    long long g0 = 0;                                 // first global word of the current segment (multiple of 624)
    const long long w0 = state->pos;                  // first word of attempt 0
    long long t_next = 0, rank = 0;                   // attempts processed so far, accepted among them
    long long last[5] = {-1, 0, 0, 0, 0};
    unsigned int* raw = nullptr;
    long long raw_words = 0;
    StreamFree free_raw{nullptr, st};
    while (rank < pairs) {
        // attempts still expected (acceptance pi/4, eight-sigma margin), capped by the segment size
        const long long missing = pairs - rank;
        long long t_want = (long long)((double)missing / 0.78539816339744831 + 12.0 * sqrt((double)missing)) + 64;
        const long long first_word = w0 + 4 * t_next;                // global index; lies in the block at g0 or the one after
        long long words = (first_word - g0) + 4 * t_want;
        if (words > kLegacySegWords) words = kLegacySegWords;
        words = ((words + rcl::kMtN - 1) / rcl::kMtN) * rcl::kMtN;
        if (words < 2 * rcl::kMtN) words = 2 * rcl::kMtN;
        const long long t_count = (g0 + words - first_word) / 4;     // attempts wholly inside [g0, g0 + words)
        // P parallel sub-streams of kMtJumpWords words each behind the segment's first block
        const int P = (int)((words - rcl::kMtN + kMtJumpWords - 1) / kMtJumpWords);
        const long long cap = words;
        if (raw_words < cap) {
            if (raw) (void)hipFreeAsync(raw, st);
            raw = nullptr;
            free_raw.p = nullptr;
            RC_HIP_CHECK(hipMallocAsync((void**)&raw, (size_t)cap * sizeof(unsigned int), st));
            free_raw.p = raw;
            raw_words = cap;
        }
        unsigned int* d_seeds = nullptr;
        RC_HIP_CHECK(hipMallocAsync((void**)&d_seeds, (size_t)P * rcl::kMtN * sizeof(unsigned int), st));
        StreamFree free_seeds{d_seeds, st};
        RC_HIP_CHECK(hipMemcpyAsync(d_seeds, carry.data(), rcl::kMtN * sizeof(unsigned int), hipMemcpyHostToDevice, st));
        if (P > 1) {
            if (int rc = ensure_func_attr(kAttrMtJump, (const void*)mt19937_jump_step_kernel, kJumpLdsWords * (int)sizeof(unsigned int)))
                return rc;
            for (int q = 1; q < P; ++q)             // window q from window q - 1: a chain of short launches
                hipLaunchKernelGGL(mt19937_jump_step_kernel, dim3(kJumpWgs), dim3(kJumpThreads),
                                   kJumpLdsWords * sizeof(unsigned int), st, d_seeds, q);
        }
        hipLaunchKernelGGL(mt19937_raw_kernel, dim3((unsigned)P), dim3(64), 0, st, (const unsigned int*)d_seeds, raw, words);
        const long long nwg = (t_count + kLgAttempts - 1) / kLgAttempts;
        unsigned long long* d_counts = nullptr;
        RC_HIP_CHECK(hipMallocAsync((void**)&d_counts, (size_t)(nwg + 1) * sizeof(unsigned long long), st));
        StreamFree free_counts{d_counts, st};
        LegacyParams p{};
        p.raw = raw;
        p.w_first = first_word - g0;
        p.t_first = t_next;
        p.t_count = t_count;
        p.rank_base = rank;
        p.pairs_needed = pairs;
        p.e_shift = e_shift;
        p.n_total = n_total;
        p.period = period;
        p.skip = skip;
        p.scales = d_scales;
        p.out = out_dev;
        p.wg_counts = d_counts;
        p.last = d_last;
        hipLaunchKernelGGL(legacy_count_kernel, dim3((unsigned)nwg), dim3(kLgThreads), 0, st, p);
        hipLaunchKernelGGL(legacy_scan_kernel, dim3(1), dim3(1024), 0, st, d_counts, nwg);
        hipLaunchKernelGGL(legacy_emit_kernel, dim3((unsigned)nwg), dim3(kLgThreads), 0, st, p);
        RC_HIP_CHECK(hipGetLastError());
        unsigned long long accepted = 0;
        RC_HIP_CHECK(hipMemcpyAsync(&accepted, d_counts + nwg, sizeof(accepted), hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipMemcpyAsync(last, d_last, sizeof(last), hipMemcpyDeviceToHost, st));
        // the segment's last block is the next segment's first (an attempt may straddle the boundary)
        RC_HIP_CHECK(hipMemcpyAsync(carry.data(), raw + words - rcl::kMtN, rcl::kMtN * sizeof(unsigned int),
                                    hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipStreamSynchronize(st));
        rank += (long long)accepted;
        if (rank >= pairs) {
            // the generator stands right after the last needed attempt: global word wf
            const long long wf = w0 + 4 * (last[0] + 1);
            long long blk = wf / rcl::kMtN, pos = wf % rcl::kMtN;
            if (pos == 0) {                           // NumPy's representation of a block boundary: pos = 624 of the block before
                blk -= 1;
                pos = rcl::kMtN;
            }
            if (blk * rcl::kMtN < g0 || (blk + 1) * rcl::kMtN > g0 + words) return fail(RC_EHIP, "legacy stream: final block outside the segment");
            RC_HIP_CHECK(hipMemcpy(state->key, raw + (blk * rcl::kMtN - g0), rcl::kMtN * sizeof(unsigned int), hipMemcpyDeviceToHost));
            state->pos = (int)pos;
            if (n_need & 1) {
                // odd count: the second normal of the last attempt stays cached - computed HERE with the host's libm,
                // exactly as NumPy computes it (legacy-distributions.c: f = sqrt(-2 log(r2) / r2); gauss = f * x1)
                double x1, x2, r2;
                rcl::polar_attempt((unsigned)last[1], (unsigned)last[2], (unsigned)last[3], (unsigned)last[4], x1, x2, r2);
                const double f = sqrt(-2.0 * log(r2) / r2);
                state->gauss = f * x1;
                state->has_gauss = 1;
            }
            break;
        }
        t_next += t_count;
        g0 += words - rcl::kMtN;
    }
    return RC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

#ifdef RC_STAMPS
// diagnostic builds only: device buffer of [ntiles][4] int64 receiving per-wave s_memtime stamps
int rc_debug_set_stamps(long long* dev_buf) { g_stamps = dev_buf; return 0; }
#endif

int rc_version(void) { return RC_ABI_VERSION; }

int rc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* rc_last_error(void) { return g_last_error.c_str(); }

long long rc_stats_general_tiles(int device, int reset) {
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EHIP, "hipSetDevice failed");
    }
    unsigned long long v = 0;
    void* addr = nullptr;
    if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&addr, HIP_SYMBOL(g_general_tiles)) != hipSuccess ||
        hipMemcpy(&v, addr, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EHIP, "reading the general-path counter failed");
    }
    if (reset && hipMemset(addr, 0, sizeof(v)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EHIP, "resetting the general-path counter failed");
    }
    return (long long)v;
}

int rc_set_fidelity_kernel(int kernel) {
    if (kernel < RC_KERNEL_AUTO || kernel > RC_KERNEL_EXPM) return fail(RC_EINVAL, "unknown kernel id");
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    g_default_kernel = kernel;
    return RC_OK;
}

int rc_mc_fidelity_f64_async(int device, void* stream, int kernel, int N, int in, int out,
                             const double* h0_diag, const double* h0_offdiag, int ring,
                             const double* controllers_dev, const double* draws_dev, long long C,
                             long long K, double* fid_out_dev) {
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_fidelity((hipStream_t)stream, kernel, N, in, out, h0_diag, h0_offdiag, ring,
                            controllers_dev, draws_dev, -1, C, K, fid_out_dev);
}

int rc_mc_fidelity_ex_f64_async(int device, void* stream, int kernel, int N, int in, int out,
                                const double* h0_diag, const double* h0_offdiag, int ring,
                                const double* controllers_dev, const double* draws_dev, long long draws_ctrl_stride,
                                long long C, long long K, double* fid_out_dev) {
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_fidelity((hipStream_t)stream, kernel, N, in, out, h0_diag, h0_offdiag, ring,
                            controllers_dev, draws_dev, draws_ctrl_stride, C, K, fid_out_dev);
}

int rc_mc_fidelity_nh_f64_async(int device, void* stream, int N, int in, int out, const double* h0_diag,
                                const double* h0_offdiag, int ring, const double* controllers_dev,
                                const double* draws_dev, const double* diag_imag_dev, long long C, long long K,
                                double* fid_out_dev) {
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (C == 0 || K == 0) return RC_OK;
    if (!controllers_dev || !draws_dev || !fid_out_dev) return fail(RC_EINVAL, "NULL array pointer");
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_expm((hipStream_t)stream, N, in, out, h0_diag, h0_offdiag, ring, controllers_dev, draws_dev,
                        K * N * 3, diag_imag_dev, K * N, C, K, fid_out_dev);
}

int rc_mc_fidelity_f64(int device, int N, int in, int out, const double* h0_diag,
                       const double* h0_offdiag, int ring, const double* controllers, const double* draws,
                       long long C, long long K, double* fid_out) {
    int kernel;
    {
        std::lock_guard<std::mutex> lk(g_cfg_mu);
        kernel = g_default_kernel;
    }
    return rc_mc_fidelity_kernel_f64(device, kernel, N, in, out, h0_diag, h0_offdiag, ring, controllers, draws, C, K,
                                     fid_out);
}

int rc_mc_fidelity_kernel_f64(int device, int kernel, int N, int in, int out, const double* h0_diag,
                              const double* h0_offdiag, int ring, const double* controllers, const double* draws,
                              long long C, long long K, double* fid_out) {
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (C == 0 || K == 0) return RC_OK;
    if (!controllers || !draws || !fid_out) return fail(RC_EINVAL, "NULL array pointer");
    if (int rc = device_in_range(device)) return rc;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(device, &ctx)) return rc;
    const size_t nb_ctrl = (size_t)C * (N + 1) * sizeof(double);
    const size_t nb_draw = (size_t)C * K * N * 3 * sizeof(double);
    const size_t nb_fid = (size_t)C * K * sizeof(double);
    const bool dc = is_device_ptr(controllers), dd = is_device_ptr(draws), df = is_device_ptr(fid_out);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t need = (dc ? 0 : up(nb_ctrl)) + (dd ? 0 : up(nb_draw)) + (df ? 0 : up(nb_fid));
    if (need) {
        if (int rc = ensure_ws(ctx, need)) return rc;
    }
    char* w = (char*)ctx->ws;
    const double* d_ctrl = controllers;
    const double* d_draw = draws;
    double* d_fid = fid_out;
    if (!dc) {
        RC_HIP_CHECK(hipMemcpyAsync(w, controllers, nb_ctrl, hipMemcpyHostToDevice, ctx->stream));
        d_ctrl = (const double*)w;
        w += up(nb_ctrl);
    }
    if (!dd) {
        RC_HIP_CHECK(hipMemcpyAsync(w, draws, nb_draw, hipMemcpyHostToDevice, ctx->stream));
        d_draw = (const double*)w;
        w += up(nb_draw);
    }
    if (!df) d_fid = (double*)w;
    if (int rc = enqueue_fidelity(ctx->stream, kernel, N, in, out, h0_diag, h0_offdiag, ring, d_ctrl, d_draw, -1, C, K,
                                  d_fid))
        return rc;
    if (!df) RC_HIP_CHECK(hipMemcpyAsync(fid_out, d_fid, nb_fid, hipMemcpyDeviceToHost, ctx->stream));
    RC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_reduce_f64_async(int device, void* stream, const double* fid_dev, long long C, long long K,
                        const double* q_thresholds, int nq, double dkw_eps, double* rim1_dev,
                        double* std_dev, double* minf_dev, double* q_dev, double* sorted_out_dev) {
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_reduce((hipStream_t)stream, fid_dev, C, K, q_thresholds, nq, dkw_eps, rim1_dev, std_dev,
                          minf_dev, q_dev, sorted_out_dev);
}

int rc_reduce_f64(int device, const double* fid, long long C, long long K, const double* q_thresholds,
                  int nq, double dkw_eps, double* rim1, double* std_, double* minf, double* q,
                  double* sorted_out) {
    if (C < 0 || K < 0) return fail(RC_EINVAL, "C and K must be non-negative");
    if (nq < 0 || nq > kMaxQ) return fail(RC_EINVAL, "nq must be in [0, 8]");
    if (C == 0) return RC_OK;
    if (!fid) return fail(RC_EINVAL, "NULL fid pointer");
    if (int rc = device_in_range(device)) return rc;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(device, &ctx)) return rc;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nb_fid = (size_t)C * K * sizeof(double);
    const size_t nb_c3 = (size_t)3 * C * sizeof(double);
    const size_t nb_q = (size_t)3 * (nq > 0 ? nq : 1) * C * sizeof(double);
    const bool df = is_device_ptr(fid);
    // outputs are always staged (they are small), the sorted tensor only when it is a host pointer
    const bool ds = sorted_out && is_device_ptr(sorted_out);
    size_t need = (df ? 0 : up(nb_fid)) + 3 * up(nb_c3) + up(nb_q) + ((sorted_out && !ds) ? up(nb_fid) : 0);
    if (int rc = ensure_ws(ctx, need)) return rc;
    char* w = (char*)ctx->ws;
    const double* d_fid = fid;
    if (!df) {
        RC_HIP_CHECK(hipMemcpyAsync(w, fid, nb_fid, hipMemcpyHostToDevice, ctx->stream));
        d_fid = (const double*)w;
        w += up(nb_fid);
    }
    double* d_rim = (double*)w; w += up(nb_c3);
    double* d_std = (double*)w; w += up(nb_c3);
    double* d_min = (double*)w; w += up(nb_c3);
    double* d_q = (double*)w;   w += up(nb_q);
    double* d_sorted = nullptr;
    if (sorted_out) d_sorted = ds ? sorted_out : (double*)w;
    if (int rc = enqueue_reduce(ctx->stream, d_fid, C, K, q_thresholds, nq, dkw_eps, rim1 ? d_rim : nullptr,
                                std_ ? d_std : nullptr, minf ? d_min : nullptr, (q && nq) ? d_q : nullptr,
                                d_sorted))
        return rc;
    if (rim1) RC_HIP_CHECK(hipMemcpyAsync(rim1, d_rim, nb_c3, hipMemcpyDefault, ctx->stream));
    if (std_) RC_HIP_CHECK(hipMemcpyAsync(std_, d_std, nb_c3, hipMemcpyDefault, ctx->stream));
    if (minf) RC_HIP_CHECK(hipMemcpyAsync(minf, d_min, nb_c3, hipMemcpyDefault, ctx->stream));
    if (q && nq) RC_HIP_CHECK(hipMemcpyAsync(q, d_q, (size_t)3 * nq * C * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (sorted_out && !ds) RC_HIP_CHECK(hipMemcpyAsync(sorted_out, d_sorted, nb_fid, hipMemcpyDeviceToHost, ctx->stream));
    RC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_rim_p_f64_async(int device, void* stream, const double* fid_dev, long long C, long long K, double p,
                       double* out_dev) {
    if (C < 0 || K <= 0) return fail(RC_EINVAL, "C must be >= 0 and K > 0");
    if (!(p > 0.0)) return fail(RC_EINVAL, "p must be positive (RIM_0 = 1 is a host constant)");
    if (C == 0) return RC_OK;
    if (!fid_dev || !out_dev) return fail(RC_EINVAL, "NULL array pointer");
    if (C > 0x7fffffffLL) return fail(RC_EINVAL, "too many controllers for one launch");
    RC_HIP_CHECK(hipSetDevice(device));
    hipLaunchKernelGGL(rim_p_kernel, dim3((unsigned)C), dim3(kRedThreads), 0, (hipStream_t)stream, fid_dev, C, K,
                       p, out_dev);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_rim_p_f64(int device, const double* fid, long long C, long long K, double p, double* out) {
    if (C < 0 || K <= 0) return fail(RC_EINVAL, "C must be >= 0 and K > 0");
    if (!(p > 0.0)) return fail(RC_EINVAL, "p must be positive (RIM_0 = 1 is a host constant)");
    if (C == 0) return RC_OK;
    if (!fid || !out) return fail(RC_EINVAL, "NULL array pointer");
    if (int rc = device_in_range(device)) return rc;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(device, &ctx)) return rc;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nb_fid = (size_t)C * K * sizeof(double), nb_out = (size_t)C * sizeof(double);
    const bool df = is_device_ptr(fid);
    if (int rc = ensure_ws(ctx, (df ? 0 : up(nb_fid)) + up(nb_out))) return rc;
    char* w = (char*)ctx->ws;
    const double* d_fid = fid;
    if (!df) {
        RC_HIP_CHECK(hipMemcpyAsync(w, fid, nb_fid, hipMemcpyHostToDevice, ctx->stream));
        d_fid = (const double*)w;
        w += up(nb_fid);
    }
    if (int rc = rc_rim_p_f64_async(device, ctx->stream, d_fid, C, K, p, (double*)w)) return rc;
    RC_HIP_CHECK(hipMemcpyAsync(out, w, nb_out, hipMemcpyDefault, ctx->stream));
    RC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_draws_philox_f64_async(int device, void* stream, unsigned long long seed, unsigned long long offset,
                              long long n, double scale, double* out_dev) {
    if (n < 0) return fail(RC_EINVAL, "n must be non-negative");
    if (n == 0) return RC_OK;
    if (!out_dev) return fail(RC_EINVAL, "NULL output pointer");
    RC_HIP_CHECK(hipSetDevice(device));
    long long blocks = (n / 2 + 1 + 255) / 256;               // one thread per Box-Muller pair
    if (blocks > 256LL * 32) blocks = 256LL * 32;
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seed, offset,
                       n, scale, out_dev);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_draws_philox_f64(int device, unsigned long long seed, unsigned long long offset, long long n, double scale,
                        double* out) {
    if (n < 0) return fail(RC_EINVAL, "n must be non-negative");
    if (n == 0) return RC_OK;
    if (!out) return fail(RC_EINVAL, "NULL output pointer");
    if (int rc = device_in_range(device)) return rc;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(device, &ctx)) return rc;
    const bool dev_out = is_device_ptr(out);
    double* d_out = out;
    if (!dev_out) {
        if (int rc = ensure_ws(ctx, (size_t)n * sizeof(double))) return rc;
        d_out = (double*)ctx->ws;
    }
    if (int rc = rc_draws_philox_f64_async(device, ctx->stream, seed, offset, n, scale, d_out)) return rc;
    if (!dev_out) RC_HIP_CHECK(hipMemcpyAsync(out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_mc_fidelity_sharded_f64(int ndev, const int* devices, int kernel, int N, int in, int out, const double* h0_diag,
                               const double* h0_offdiag, int ring, const double* controllers, const double* draws,
                               long long C, long long K, double* fid_out) {
    if (C > 0 && K > 0 && (!draws || !fid_out)) return fail(RC_EINVAL, "NULL array pointer");
    ShardJob j{};
    j.kernel = kernel; j.N = N; j.in = in; j.out = out; j.ring = ring;
    j.h0d = h0_diag; j.h0o = h0_offdiag; j.ctrl = controllers; j.draws = draws;
    j.C = C; j.K = K; j.fid_out = fid_out;
    return run_sharded(ndev, devices, j);
}

int rc_mc_metrics_sharded_f64(int ndev, const int* devices, int kernel, int N, int in, int out, const double* h0_diag,
                              const double* h0_offdiag, int ring, const double* controllers, const double* draws,
                              unsigned long long philox_seed, unsigned long long philox_offset, double sigma,
                              long long C, long long K, const double* q_thresholds, int nq, double dkw_eps,
                              double* rim1, double* std_, double* minf, double* q, double* fid_out) {
    ShardJob j{};
    j.kernel = kernel; j.N = N; j.in = in; j.out = out; j.ring = ring;
    j.h0d = h0_diag; j.h0o = h0_offdiag; j.ctrl = controllers; j.draws = draws;
    j.seed = philox_seed; j.offset = philox_offset; j.sigma = sigma;
    j.C = C; j.K = K; j.thr = q_thresholds; j.nq = nq; j.eps = dkw_eps;
    j.rim1 = rim1; j.stdv = std_; j.minf = minf; j.q = q; j.fid_out = fid_out;
    return run_sharded(ndev, devices, j);
}

int rc_draws_legacy_f64(int device, void* stream, rc_mt19937_state* state, long long n_periods, long long period,
                        long long skip, const double* scales, double* out_dev) {
    if (!state) return fail(RC_EINVAL, "NULL generator state");
    if (state->pos < 0 || state->pos > 624) return fail(RC_EINVAL, "generator state: pos must be in [0, 624]");
    if (n_periods < 0 || period < 0 || skip < 0 || skip > period) return fail(RC_EINVAL, "need n_periods, period >= 0 and 0 <= skip <= period");
    if (n_periods == 0 || period == 0) return RC_OK;
    if (!scales) return fail(RC_EINVAL, "NULL scales pointer");
    if (skip < period && !out_dev) return fail(RC_EINVAL, "NULL output pointer");
    if (int rc = device_in_range(device)) return rc;
    RC_HIP_CHECK(hipSetDevice(device));
    return legacy_normal_stream((hipStream_t)stream, state, n_periods, period, skip, scales, out_dev);
}

}  // extern "C"
