// Lane-per-sample fidelity kernels of the chain topology: mc_fid_chain_kernel<N, MODE> (N <= 16, register-resident
// tridiagonal QL, LDS-DMA staging) and mc_fid_chain_anyn_kernel (16 < N <= 32, work vectors in LDS); wave reductions.
//
// Part of ONE translation unit: this file is #included by robchar_hip.hip INSIDE its anonymous namespace (after the
// shared parameter structs); it is not a stand-alone header.
// Staging geometry.  A wave's 64-sample tile is brought in through LDS in `fid_phases(N)` phases of
// 64/phases samples each, so that the per-wave LDS buffer (samples-per-phase * 3N doubles: 5.4 KiB at N = 7)
// never limits residency below what the registers allow (98 VGPRs at N = 7, 4 waves per SIMD).
// Weight mode and residency.  Measured on MI355X (kbench, 1e6 evaluations, N = 7; round 1, all-fp64 QL): eigenvector
// rows 98 us, general adjugate 83 us, end-to-end adjugate 78 us (more waves or staging phases change nothing: the kernel
// is bound by the energy of its VALU work, not by latency).  The adjugate modes win at every N (2..16), so AUTO =
// adjugate (its end-to-end specialisation when {in,out} = {0,N-1}); the rows mode stays selectable as a cross-check.
// Round 2: the eigenvalue-only modes at N = 3..13 compute their eigenvalues in mixed precision (tridiag_core.h:
// fp32 QL rotations + one fp64 Halley step): N = 7 end-to-end 52-56 us, general adjugate 55-59 us.
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
    // (general adjugate at N = 13: 268 registers with the mixed-precision state - one wave; N >= 14 runs the all-fp64 QL.
    // With all THREE moment rules of the a-posteriori guard (-DRC_SUM_RULE_MOMENTS=3) four picked matrix entries stay alive
    // through the eigenvalue phase: N = 8 then needs 3 waves instead of 4, N = 14..16 one instead of 2)
    // (N = 17 .. 24, round 5: one wave - 14 N doubles of state and the unrolled recurrences of the weights)
    if (mode == rc::kWeightsAdjugate)
        return n >= 17 ? 1
               : rc::kSumRuleMoments >= 3 ? (n <= 6 ? RC_WAVES_SMALL : (n <= 7 ? 4 : (n <= 8 ? 3 : (n >= 13 ? 1 : 2))))
                                          : (n <= 6 ? RC_WAVES_SMALL : (n <= 8 ? 4 : (n == 13 ? 1 : 2)));
    if (mode == rc::kWeightsRows) return n <= 8 ? RC_WAVES_SMALL : (n <= 12 ? 3 : 2);
    // kWeightsEnds
    return n <= 6 ? RC_WAVES_SMALL : (n <= 8 ? 4 : (n <= 10 ? 3 : (n <= 14 ? 2 : 1)));
}
// staging phases: the LDS buffer (64/phases * 3N doubles per wave) must not cap residency below the register limit
constexpr int fid_phases(int n, int mode) { return n <= 2 ? 1 : (n <= 8 ? 2 : 4); }

// (cos, sin)(2 pi k / 64), k = 0..63: source of the per-wave LDS copy that sincos_table reads
__device__ const double g_sincos_table[128] = {RC_SINCOS_TABLE_VALUES};

// Tiles with at least one sample that left the fast path (sweep cap / degenerate pair) since the last reset: a
// diagnostic counter, touched only on that rare path (rc_stats_general_tiles).
__device__ unsigned long long g_general_tiles = 0;
// Tiles of the mixed-precision path that needed more than its one Halley step (close eigenvalue pair somewhere in the
// tile): also rare-path only (rc_stats_polish_tiles).
// 64 slots (tile index mod 64), summed by the reader: ~9 % of the tiles bump it - one address would serialise them in L2.
__device__ unsigned long long g_polish_tiles[64] = {0};

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
    // (no initialisation: every lane < nk reads its G values in exactly one phase, lanes >= nk never use theirs -
    // zeroing them was 2 G v_mov_b32 per wave)
    double gl[G];
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
            int rel = lane - first;
            // opaque to the optimiser: otherwise `- first * G * 8` is folded into every read's address (a negative offset
            // does not fit the ds_read immediate: one v_mov + v_mad per read); this way ONE base address + immediates
            asm volatile("" : "+v"(rel));
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
    int extra = 0;
    if (lane < nk)
        ok = rc::chain_fidelity_fast<N, MODE>(x, p.h0.diag, p.h0.off, [&gl](int i) { return gl[i]; }, p.in, p.out, sctab, f,
                                              nullptr, &extra);
    if (extra && lane == 0) atomicAdd(&g_polish_tiles[blockIdx.x & 63u], 1ull);
#endif
    // (N >= 17: no rows-mode QL in registers - a bad lane goes straight to the LDS routine below)
    constexpr bool kRepairInRegisters = MODE != rc::kWeightsRows && N <= RC_MAX_NSPIN_FAST;
    unsigned long long badmask = __ballot(lane < nk && !ok);
    if (RC_UNLIKELY(badmask != 0ull)) {
        if (lane == 0) atomicAdd(&g_general_tiles, 1ull);
        // Rare: some samples of this tile left the fast path (degenerate eigenvalue pair - the eigenvalue-only weights need
        // distinct eigenvalues -, sweep cap, overflow).  REPAIR, step 1: those lanes alone run the register-resident QL
        // with eigenvector rows (the rows-mode fast path: no condition on the gaps), re-reading their draws from HBM - a
        // ~10 us detour for the tile instead of the ~100 us single-lane straggler the LDS routine below is at N >= 10.
        if (kRepairInRegisters) {
            const bool bad = (badmask >> lane) & 1ull;
            bool ok2 = true;
            if (bad) {
                const double* gsrc = (const double*)src + (long long)lane * G;
                double f2;
                // (the controller row is re-read through `xg`: keeping the fast path's copy alive for this path costs
                // registers at every N where the scalar file is full)
                ok2 = rc::chain_fidelity_fast<N, rc::kWeightsRows>(xg, p.h0.diag, p.h0.off, [gsrc](int i) { return gsrc[i]; },
                                                                   p.in, p.out, sctab, f2);
                if (ok2) f = f2;
            }
            badmask = __ballot(bad && !ok2);
        }
    }
    if (RC_UNLIKELY(badmask != 0ull)) {
        // Step 2 (last resort: the rows-mode QL hit its sweep cap too - not observed): the textbook per-sample routine,
        // CH lanes at a time, work vectors (4N doubles per sample) in the LDS staging buffer, which is free now.
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
// fidelity kernel: RING topology, lane per sample (hermitian_core.h), N = 3 .. kRingMaxN
// ------------------------------------------------------------------------------------------------
// Same tiling and staging as mc_fid_chain_kernel (one wave per 64-sample tile of one controller, draws by LDS-DMA);
// per lane the complex Hermitian matrix in registers -> Householder tridiagonalisation with rows in / out of Q -> the
// shared QL iteration with two complex rows.  Samples that hit the QL sweep cap are recomputed with the general
// routine, their vectors (6 N doubles per sample) in the free LDS staging buffer.
// (round 5) N = 11 .. 16 through the folded band reduction (hermitian_core.h: ring_fold_*): 10 N doubles of state instead of N^2
constexpr int kRingMaxN = 16;
constexpr int ring_min_waves(int n) { return n <= 4 ? 5 : (n <= 5 ? 4 : (n <= 6 ? 3 : (n <= 8 ? 2 : 1))); }

// HBM -> LDS -> registers for one tile (see mc_fid_chain_kernel): the nk * G doubles at `src` through the `stage` buffer in
// PH phases; lane l < nk ends with its G values in gl.
template <int G, int PH>
__device__ __forceinline__ void stage_tile_draws(const char* src, int nk, int lane, int align16, double* stage, double (&gl)[G]) {
    constexpr int SP = 64 / PH;
    constexpr int kPhaseBytes = SP * G * 8;
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
        const int first = ph * SP;
        if (first < nk) {                          // wave-uniform
            const int cnt = (nk - first < SP) ? (nk - first) : SP;
            const int bytes = cnt * G * 8;
            const char* ps = src + (long long)first * G * 8;
            if (align16 && !(cnt & 1)) {
#pragma unroll
                for (int it = 0; it < (kPhaseBytes + 1023) / 1024; ++it) {
                    const int off = it * 1024 + lane * 16;
                    if (off < bytes)
                        __builtin_amdgcn_global_load_lds((rc_gptr_t)(ps + off), (rc_lptr_t)((char*)stage + it * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll 2
                for (int it = 0; it < (kPhaseBytes + 255) / 256; ++it) {
                    const int off = it * 256 + lane * 4;
                    if (off < bytes)
                        __builtin_amdgcn_global_load_lds((rc_gptr_t)(ps + off), (rc_lptr_t)((char*)stage + it * 256), 4, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int rel = lane - first;
            asm volatile("" : "+v"(rel));                             // one base address + immediate offsets (see chain kernel)
            if (rel >= 0 && rel < cnt) {
#pragma unroll
                for (int i = 0; i < G; ++i) gl[i] = stage[rel * G + i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// One tile of the ring topology through the all-fp64 route (Householder with rows + QL with rows; hermitian_core.h).
// `sctab` must have been filled; one wave.
template <int N>
__device__ __forceinline__ void ring_tile_fp64(const FidParams& p, const double corner, const long long tile, double* stage,
                                               const double* sctab) {
    constexpr int G = 3 * N;
    constexpr int PH = fid_phases(N, rc::kWeightsRows);
    constexpr int SP = 64 / PH;
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    const long long c = tile / p.tiles_per_ctrl;
    const long long kb = (tile - c * p.tiles_per_ctrl) * 64;
    const int nk = (int)((p.K - kb < 64) ? (p.K - kb) : 64);
    const double* xg = p.ctrl + c * (N + 1);
    double x[N + 1];
    bool pad = false;
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        x[i] = xg[i];
        pad |= (x[i] != x[i]);
    }
    double* dst = p.fid + c * p.K + kb;
    if (pad) {                                     // NaN-padded controller (mcsim.py:442-443): no draws read
        if (lane < nk) dst[lane] = __builtin_nan("");
        __builtin_amdgcn_s_setprio(0);
        return;
    }
    const char* src = (const char*)(p.draws + c * p.draw_cstride + kb * G);
    double gl[G];                                  // (not initialised: see mc_fid_chain_kernel)
    stage_tile_draws<G, PH>(src, nk, lane, p.align16, stage, gl);
    __builtin_amdgcn_s_setprio(0);

    double f = 0.0;
    bool ok = true;
    if (lane < nk)
        ok = rc::ring_fidelity_fast<N>(x, p.h0.diag, p.h0.off, corner, [&gl](int i) { return gl[i]; }, p.in, p.out, sctab, f);
    const unsigned long long badmask = __ballot(lane < nk && !ok);
    if (badmask) {
        if (lane == 0) atomicAdd(&g_general_tiles, 1ull);
        constexpr int CH = (SP * G) / (6 * N);
        const bool bad = (badmask >> lane) & 1ull;
        const int rank = __popcll(badmask & ((1ull << lane) - 1ull));
        const int nbad = __popcll(badmask);
#pragma unroll 1
        for (int c0 = 0; c0 < nbad; c0 += CH) {
            const int rel = rank - c0;
            if (bad && rel >= 0 && rel < CH) {
                const LdsVec vd{stage + rel, CH}, ve{stage + N * CH + rel, CH};
                LdsVec vz[4] = {{stage + 2 * N * CH + rel, CH}, {stage + 3 * N * CH + rel, CH}, {stage + 4 * N * CH + rel, CH},
                                {stage + 5 * N * CH + rel, CH}};
                // the sample's draws are re-read from HBM (rare path): keeping them in registers across the fast path
                // would cost 6 N VGPRs of its residency
                const double* gsrc = (const double*)src + (long long)lane * G;
                f = rc::ring_fidelity_general<N>(x, p.h0.diag, p.h0.off, corner, [gsrc](int i) { return gsrc[i]; }, p.in, p.out,
                                                 vd, ve, vz);
            }
        }
    }
    if (lane < nk) dst[lane] = f;
}

template <int N>
__global__ __launch_bounds__(64, ring_min_waves(N)) void mc_fid_ring_kernel(const FidParams p, const double corner) {
    constexpr int G = 3 * N;
    constexpr int SP = 64 / fid_phases(N, rc::kWeightsRows);
    __shared__ __attribute__((aligned(16))) double stage[SP * G];
    __shared__ __attribute__((aligned(16))) double sctab[128];
    if (rc::kTableSinCos) {
        const double2 ent = reinterpret_cast<const double2*>(g_sincos_table)[threadIdx.x];
        reinterpret_cast<double2*>(sctab)[threadIdx.x] = ent;
        __syncthreads();                           // (one wave per workgroup: no wait)
    }
    ring_tile_fp64<N>(p, corner, blockIdx.x, stage, sctab);
}

// ------------------------------------------------------------------------------------------------
// RING topology, mixed-precision eigenvalue route (hermitian_core.h: ring_fidelity_mixed) - the AUTO choice for N <= 10.
// Same tiling and staging.  A SAMPLE the route does not trust itself with (an eigenvalue pair closer than 1e-3 of the
// scale, or one the fp32-start / fp64-Halley scheme cannot settle) gets NaN and is appended to a list (one atomic per
// wave); mc_fid_ring_repair_kernel, enqueued right behind on the same stream, recomputes the listed samples with the
// all-fp64 route, lane per sample - the bad samples of ALL tiles packed into full waves.  Keeping that route out of this
// kernel is what lets it run at 3-5 waves per SIMD (the all-fp64 route needs twice the registers).
// ------------------------------------------------------------------------------------------------
// (the guard's accumulators cost N = 5 and N = 7 one wave of residency: 5 -> 4, 4 -> 3 - measured together with the m = 0
// rule at +1.7 % for both sizes; with all three moment rules, -DRC_SUM_RULE_MOMENTS=3, N = 9 goes 3 -> 2 as well)
constexpr int ring_mixed_min_waves(int n) {
#if defined(RC_RWAVES_N) && defined(RC_RWAVES_W)
    if (n == RC_RWAVES_N) return RC_RWAVES_W;      // residency experiments (scripts/build_variant.sh)
#endif
    return n <= 4 ? 5 : (n <= 6 ? 4 : (n <= 8 ? 3 : (n == 9 ? (rc::kSumRuleMoments >= 3 ? 2 : 3) : (n <= 13 ? 2 : 1))));
}

// (RingRepairList: kernel_params.h)

template <int N>
__global__ __launch_bounds__(64, ring_mixed_min_waves(N)) void mc_fid_ring_mixed_kernel(const FidParams p, const double corner,
                                                                                        const RingRepairList rl) {
    constexpr int G = 3 * N;
    // (N = 16: eight staging phases instead of four - with the same <G, PH> staging instantiation in this kernel and in
    // mc_fid_ring_kernel<16> the gfx950 backend of ROCm 7.2 stops with "Illegal instruction detected: Operand has incorrect
    // register class ... V_CMP_NE_U32_e32 0, $src_shared_base"; either kernel alone, or any other N, compiles)
    constexpr int PH = (N == 16) ? 8 : fid_phases(N, rc::kWeightsEnds);
    constexpr int SP = 64 / PH;
    __shared__ __attribute__((aligned(16))) double stage[SP * G];
    __shared__ __attribute__((aligned(16))) double sctab[128];
    const int lane = threadIdx.x;
    const long long tile = blockIdx.x;
    // the list counters alternate between calls: this call appends to `count` and empties the one the next call will use
    // (whoever read that one - the previous call's repair kernel - finished before this kernel started: stream order)
    if (blockIdx.x == 0 && lane == 0) *rl.clear = 0ull;
    if (rc::kTableSinCos) {
        const double2 ent = reinterpret_cast<const double2*>(g_sincos_table)[lane];
        reinterpret_cast<double2*>(sctab)[lane] = ent;
    }
    __builtin_amdgcn_s_setprio(3);
    const long long c = tile / p.tiles_per_ctrl;
    const long long kb = (tile - c * p.tiles_per_ctrl) * 64;
    const int nk = (int)((p.K - kb < 64) ? (p.K - kb) : 64);
    const double* xg = p.ctrl + c * (N + 1);
    double x[N + 1];
    bool pad = false;
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        x[i] = xg[i];
        pad |= (x[i] != x[i]);
    }
    double* dst = p.fid + c * p.K + kb;
    if (pad) {                                     // NaN-padded controller (mcsim.py:442-443): no draws read
        if (lane < nk) dst[lane] = __builtin_nan("");
        return;
    }
    const char* src = (const char*)(p.draws + c * p.draw_cstride + kb * G);
    double gl[G];
    stage_tile_draws<G, PH>(src, nk, lane, p.align16, stage, gl);
    __builtin_amdgcn_s_setprio(0);
    if (rc::kTableSinCos) __syncthreads();
    double f = 0.0;
    bool ok = true;
    int extra = 0;
    if (lane < nk)
        ok = rc::ring_fidelity_mixed<N>(x, p.h0.diag, p.h0.off, corner, [&gl](int i) { return gl[i]; }, p.in, p.out, sctab, f,
                                        &extra);
    if (extra && lane == 0) atomicAdd(&g_polish_tiles[blockIdx.x & 63u], 1ull);
    const bool bad = lane < nk && !ok;
    const unsigned long long badmask = __ballot(bad);
    if (badmask) {                                 // list the bad samples: one atomic per wave
        const int first = __ffsll((long long)badmask) - 1;
        unsigned long long base = 0;
        if (lane == first) base = atomicAdd(rl.count, (unsigned long long)__popcll(badmask));
        base = __shfl(base, first, 64);
        if (bad) {
            rl.samples[base + __popcll(badmask & ((1ull << lane) - 1ull))] = c * p.K + kb + lane;
            f = __builtin_nan("");
        }
    }
    if (lane < nk) dst[lane] = f;
}

// The listed samples through the all-fp64 route (Householder with rows + QL with rows), lane per sample: every lane reads
// ITS controller row and draws straight from HBM.  Grid-stride over the list; every wave ends when the list is exhausted.
// A sample whose QL hits the sweep cap (not observed) takes the textbook routine with its vectors in LDS (6 N doubles per lane).
template <int N>
__global__ __launch_bounds__(64, 1) void mc_fid_ring_repair_kernel(const FidParams p, const double corner, const RingRepairList rl) {
    constexpr int G = 3 * N;
    __shared__ __attribute__((aligned(16))) double work[6 * N * 64];
    __shared__ __attribute__((aligned(16))) double sctab[128];
    const int lane = threadIdx.x;
    if (rc::kTableSinCos) {
        const double2 ent = reinterpret_cast<const double2*>(g_sincos_table)[lane];
        reinterpret_cast<double2*>(sctab)[lane] = ent;
        __syncthreads();
    }
    const long long count = (long long)*rl.count;
#pragma unroll 1
    for (long long i0 = (long long)blockIdx.x * 64; i0 < count; i0 += (long long)gridDim.x * 64) {
        const long long i = i0 + lane;
        if (lane == 0) atomicAdd(&g_general_tiles, 1ull);              // (rc_stats_general_tiles: repaired waves of 64 ring samples)
        const long long sidx = (i < count) ? rl.samples[i] : -1;
        if (sidx >= 0 && sidx < p.C * p.K) {           // (the range check: a list left behind by an aborted call cannot reach outside)
            const long long c = sidx / p.K, k = sidx - c * p.K;
            const double* xg = p.ctrl + c * (N + 1);
            double x[N + 1];
#pragma unroll
            for (int j = 0; j <= N; ++j) x[j] = xg[j];
            const double* gsrc = p.draws + c * p.draw_cstride + k * G;
            double f;
            // (<N, true>: per-lane sweeps - a listed sample's result does not depend on which other samples were listed with it)
            const bool ok = rc::ring_fidelity_fast<N, true>(x, p.h0.diag, p.h0.off, corner, [gsrc](int j) { return gsrc[j]; }, p.in,
                                                            p.out, sctab, f);
            if (!ok) {
                const LdsVec vd{work + lane, 64}, ve{work + N * 64 + lane, 64};
                LdsVec vz[4] = {{work + 2 * N * 64 + lane, 64}, {work + 3 * N * 64 + lane, 64}, {work + 4 * N * 64 + lane, 64},
                                {work + 5 * N * 64 + lane, 64}};
                f = rc::ring_fidelity_general<N>(x, p.h0.diag, p.h0.off, corner, [gsrc](int j) { return gsrc[j]; }, p.in, p.out, vd,
                                                 ve, vz);
            }
            p.fid[sidx] = f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fidelity kernel for long chains (RC_MAX_NSPIN_CHAIN < N <= RC_MAX_NSPIN; the rows mode from N = 17): the general per-sample routine for every
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
