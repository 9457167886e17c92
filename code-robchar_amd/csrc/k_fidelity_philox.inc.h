// mc_fid_chain_philox_kernel<N, MODE>: the chain fidelity kernel with the counter-based draws generated WHERE THEY ARE CONSUMED
// (round 4; SURVEY.md 8(d): "in philox mode draws are not read").  Same tiling and the same per-sample arithmetic as
// mc_fid_chain_kernel (k_fidelity_chain.inc.h) - one sample per lane, one 64-sample tile of ONE controller per wave - but no
// draw tensor exists: lane (c, k) regenerates its 3N normals from the stream of rc_draws_philox_f64,
//     element  offset + ((c K + k) N + i) 3 + s   of stream `seed`, scaled by sigma (per controller row: all sigma levels of an
//     algorithm go through one launch with the controller rows tiled L times),
// with the SAME routine philox_normal_kernel uses (philox_pair): the fidelities are bit-identical to the two-kernel route
// (tests/test_gpu_rng.py), and oracle/philox_host.py still regenerates any element on the host.
// Cost (DESIGN.md 8 xiii): 3N/2 + 1 Box-Muller pairs per sample (the pair grid straddles samples: one output of the first or the
// last pair belongs to a neighbour), ~150 VALU instructions each - about what the fidelity itself costs at N = 7; what it saves
// is the generator's 16-byte store per pair, the fidelity kernel's read of it, and the 24 N bytes per sample of HBM capacity
// (BASELINE config 4: 16.8 GB per sigma level).
//
// Part of ONE translation unit: #included by robchar_hip.hip inside its anonymous namespace, after k_draws.inc.h.
struct PhiloxDraws {
    unsigned long long seed;
    unsigned long long offset;        // stream element of sample (c = 0, k = 0), site 0, slot 0
    const double* sigma_rows;         // [C] scale per controller row, or NULL: `sigma` for all
    double sigma;
};

// element `e` of the stream on its own (rare paths only: one Philox call per element)
__device__ __forceinline__ double philox_element(unsigned long long seed, unsigned long long e, double scale, const double* lntab,
                                                 const double* sctab) {
    double amp, cs, sn;
    philox_pair(seed, e >> 1, scale, lntab, sctab, amp, cs, sn);
    double v = (e & 1ull) ? amp * sn : amp * cs;
    asm volatile("" : "+v"(v));                    // rounded product, never contracted into its consumer (see the kernel)
    return v;
}

// (one wave less than the staging kernel where that one sits at its register limit: the pair values live beside the matrix
// while it is being formed; residency above ~3 waves buys nothing)
constexpr int fid_philox_min_waves(int n, int mode) {
    const int w = fid_min_waves(n, mode);
    return n >= 14 ? 1 : (w > 3 ? 3 : w);
}

template <int N, int MODE>
__global__ __launch_bounds__(64, fid_philox_min_waves(N, MODE)) void mc_fid_chain_philox_kernel(const FidParams p, const PhiloxDraws q) {
    constexpr int G = 3 * N;                       // doubles per sample
    constexpr int NP = G / 2 + 1;                  // Box-Muller pairs that cover G consecutive elements from either parity
    constexpr int CH = 4;                          // lanes per pass of the last-resort routine (work vectors in LDS)
    __shared__ __attribute__((aligned(16))) double sctab[128];
    __shared__ __attribute__((aligned(16))) double lntab[256];
    __shared__ __attribute__((aligned(16))) double work[(4 * N + G) * CH];

    const int lane = threadIdx.x;
    const long long tile = blockIdx.x;             // wave-uniform
    reinterpret_cast<double2*>(sctab)[lane] = reinterpret_cast<const double2*>(g_sincos_table)[lane];
    reinterpret_cast<double2*>(lntab)[lane] = reinterpret_cast<const double2*>(g_ln_table)[lane];
    reinterpret_cast<double2*>(lntab)[lane + 64] = reinterpret_cast<const double2*>(g_ln_table)[lane + 64];
    __syncthreads();                               // (one wave per workgroup: no wait)
    const long long c = tile / p.tiles_per_ctrl;
    const long long kb = (tile - c * p.tiles_per_ctrl) * 64;
    const int nk = (int)((p.K - kb < 64) ? (p.K - kb) : 64);

    const double* xg = p.ctrl + c * (N + 1);       // controller row: wave-uniform -> scalar registers
    double x[N + 1];
    bool pad = false;
#pragma unroll
    for (int i = 0; i <= N; ++i) {
        x[i] = xg[i];
        pad |= (x[i] != x[i]);
    }
    double* dst = p.fid + c * p.K + kb;
    if (pad) {                                     // NaN-padded controller (mcsim.py:442-443): its draws are not generated
        if (lane < nk) dst[lane] = __builtin_nan("");
        return;
    }
    const double sigma = q.sigma_rows ? q.sigma_rows[c] : q.sigma;
    // this lane's G elements start at E; the pairs (2 ctr, 2 ctr + 1) that cover them start at ctr = E >> 1
    const unsigned long long E = q.offset + (unsigned long long)(c * p.K + kb + lane) * (unsigned long long)G;
    double gl[G];
    if (lane < nk) {
        // pair t = counter (E >> 1) + t holds the elements (2 t, 2 t + 1) after E when E is even and (2 t - 1, 2 t) when it is
        // odd: every gl[i] is a select between two VALUES of neighbouring pairs (written this way - not as v[i + odd] on an
        // array of pair values - because a select between array elements becomes a load from a selected address: scratch)
        const unsigned long long c0 = E >> 1;
        const bool odd = (E & 1ull) != 0ull;
        double sn_prev = 0.0;
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            double amp, cs, sn;
            philox_pair(q.seed, c0 + (unsigned long long)t, sigma, lntab, sctab, amp, cs, sn);
            // (rounded products, as philox_normal_kernel stores them: left to the compiler they would be contracted into the
            // fma that forms the matrix entry - a different rounding than the two-kernel route's)
            double a = amp * cs, b = amp * sn;
            asm volatile("" : "+v"(a), "+v"(b));      // (opaque: __dmul_rn is a plain multiply to the optimiser)
            if (2 * t < G) gl[2 * t] = odd ? b : a;
            if (t >= 1 && 2 * t - 1 < G) gl[2 * t - 1] = odd ? a : sn_prev;
            sn_prev = b;
        }
    }

    double f = 0.0;
    bool ok = true;
    int extra = 0;
    if (lane < nk)
        ok = rc::chain_fidelity_fast<N, MODE>(x, p.h0.diag, p.h0.off, [&gl](int i) { return gl[i]; }, p.in, p.out, sctab, f, nullptr,
                                              &extra);
    if (extra && lane == 0) atomicAdd(&g_polish_tiles[blockIdx.x & 63u], 1ull);
    unsigned long long badmask = __ballot(lane < nk && !ok);
    if (badmask) {
        if (lane == 0) atomicAdd(&g_general_tiles, 1ull);
        // rare: the eigenvector-rows route for the lanes the eigenvalue-only weights cannot take; their draws are regenerated
        // element by element (nothing of the fast path has to stay alive)
        const bool bad = (badmask >> lane) & 1ull;
        bool ok2 = true;
        if (bad) {
            double f2;
            const unsigned long long seed = q.seed;
            const double* lt = lntab;
            const double* st = sctab;
            ok2 = rc::chain_fidelity_fast<N, rc::kWeightsRows>(
                xg, p.h0.diag, p.h0.off, [seed, E, sigma, lt, st](int i) { return philox_element(seed, E + (unsigned long long)i, sigma, lt, st); },
                p.in, p.out, sctab, f2);
            if (ok2) f = f2;
        }
        badmask = __ballot(bad && !ok2);
    }
    if (badmask) {
        // last resort (the rows-mode QL at its sweep cap): the textbook per-sample routine, CH lanes at a time, draws and
        // work vectors in LDS
        const bool bad = (badmask >> lane) & 1ull;
        const int rank = __popcll(badmask & ((1ull << lane) - 1ull));
        const int nbad = __popcll(badmask);
#pragma unroll 1
        for (int c0 = 0; c0 < nbad; c0 += CH) {
            const int rel = rank - c0;
            if (bad && rel >= 0 && rel < CH) {
                double* g = work + 4 * N * CH + rel * G;
                for (int i = 0; i < G; ++i) g[i] = philox_element(q.seed, E + (unsigned long long)i, sigma, lntab, sctab);
                const LdsVec vd{work + rel, CH}, ve{work + N * CH + rel, CH}, va{work + 2 * N * CH + rel, CH}, vb{work + 3 * N * CH + rel, CH};
                f = rc::chain_fidelity_general(N, xg, p.h0.diag, p.h0.off, g, p.in, p.out, vd, ve, va, vb);
            }
        }
    }
    if (lane < nk) dst[lane] = f;
}

template <int N, int MODE>
int launch_chain_philox(hipStream_t s, const FidParams& p, const PhiloxDraws& q) {
    if (p.ntiles > 0x7fffffffLL) return fail(RC_EINVAL, "too many tiles for one launch");
    hipLaunchKernelGGL((mc_fid_chain_philox_kernel<N, MODE>), dim3((unsigned)p.ntiles), dim3(64), 0, s, p, q);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}
