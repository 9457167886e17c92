// `directional_perturbation` (noise_model.py:150-201) evaluated straight from what its RNG consumption leaves per sample:
// a direction index and two normals (20 bytes) - no (C, K, N, 3) draw tensor, no imaginary-diagonal plane, no host or
// torch step between the generator and the fidelity (round 4; round 3 scattered the dense layout with ~12 torch indexing
// kernels and read 168 bytes per sample of which 8-16 were non-zero).
//
// Part of ONE translation unit: this file is #included by robchar_hip.hip INSIDE its anonymous namespace.
//
// A sample perturbs ONE element pair (p, q) of the Hamiltonian: z[p,q] = a + ib, z[q,p] = a - ib (noise_model.py:190-199).
//   * bond directions (|p - q| = 1, two thirds of the list): a Hermitian perturbation of one coupling - the sample is the
//     controller's tridiagonal matrix with ONE modified bond; chain topology: the real tridiagonal routes of
//     tridiag_core.h (mixed-precision eigenvalues, eigenvalue-only weights, a-posteriori guard, rows-mode repair);
//   * diagonal directions (p = q): the second assignment overwrites the first, H[p,p] += a - ib - a complex diagonal
//     entry, non-Hermitian: the complex symmetric QL route of csym_core.h.
// The two classes run different code of very different cost (~1 400 against ~6 000 instructions per wave at N = 7), so
// the samples are first PARTITIONED by class - stably, by a count / offset / scatter pass with a fixed block order, so
// that the composition of every wave (and with it every wave-uniform decision of the mixed-precision route) is the same
// in every run - and each class is evaluated by its own kernel over its compacted list, lane per sample, every lane
// reading ITS controller row.  A sample neither route settles (not observed) is listed and recomputed by the dense
// Pade-expm kernel, the reference's own algorithm shape (mc_fid_expm_kernel in list mode).

struct DirParams {
    const double* ctrl;        // [C][N+1]
    const int* idx;            // [n] direction index per sample (the reference's `directions` list order)
    const double* ab;          // [n][2] the two normals, scaled by sigma
    double* fid;               // [n]
    long long K, n;
    long long first;           // global index of this call's sample 0 (idx / ab / fid are offset by it): controller = (first + s) / K
    int in, out;
    int* list;                 // [n] sample indices: bond class from the front, diagonal class from the back
    unsigned int* blk_counts;  // [nblocks] bond samples per partition block
    unsigned int* counts;      // [0] bond samples in all, [1] samples handed to the expm pass
    int* marked;               // [n] samples for the expm pass
    StaticH h0;
};

// direction index t -> class (0 bond / 1 diagonal), the site whose draw slots it fills in the structured layout
// (bond: slots (3 site + 1, 3 site + 2) = (a, sign * b); diagonal: slot 3 site = a, imaginary diagonal = -b), and that sign.
// Order of the list (noise_model.py:160-167): (0,0), (N-1,N-1), then (d,d-1), (d,d), (d,d+1) for d = 1..N-2, then (0,1),
// (1,0), (N-2,N-1), (N-1,N-2).  z[p][p-1] = a + ib is the lower element of bond p: (a, +b); z[p][p+1] = a + ib makes the
// lower element z[p+1][p] = a - ib: (a, -b) at site p + 1.
__device__ __forceinline__ void dir_decode(int t, int N, int& cls, int& site, double& sign) {
    sign = 1.0;
    if (t < 2) {
        cls = 1;
        site = t ? N - 1 : 0;
        return;
    }
    const int u = t - 2;
    if (u < 3 * (N - 2)) {
        const int d = 1 + u / 3, o = u - 3 * (d - 1);          // o = 0, 1, 2 <-> (d,d-1), (d,d), (d,d+1)
        cls = (o == 1) ? 1 : 0;
        site = (o == 2) ? d + 1 : d;
        sign = (o == 2) ? -1.0 : 1.0;
        return;
    }
    const int v = u - 3 * (N - 2);                             // (0,1), (1,0), (N-2,N-1), (N-1,N-2)
    cls = 0;
    site = (v < 2) ? 1 : N - 1;
    sign = (v == 0 || v == 2) ? -1.0 : 1.0;
}

constexpr int kDirPartThreads = 1024;

// pass 1: bond samples per block of kDirPartThreads consecutive samples
__global__ __launch_bounds__(kDirPartThreads) void dir_class_count_kernel(const DirParams p, int N) {
    __shared__ unsigned int wsum[kDirPartThreads / 64];
    const long long s = (long long)blockIdx.x * kDirPartThreads + threadIdx.x;
    bool bond = false;
    if (s < p.n) {
        int cls, site;
        double sign;
        dir_decode(p.idx[s], N, cls, site, sign);
        bond = cls == 0;
    }
    const unsigned long long m = __ballot(bond);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int w = 0; w < kDirPartThreads / 64; ++w) t += wsum[w];
        p.blk_counts[blockIdx.x] = t;
    }
}

// pass 2: every block adds up the counts of the blocks before it (<= 8 192 of them per call: the driver chunks), then
// scatters its samples: bond class to list[offset ...] in sample order, diagonal class to list[n - 1 - ...]
__global__ __launch_bounds__(kDirPartThreads) void dir_class_scatter_kernel(const DirParams p, int N) {
    __shared__ unsigned int red[kDirPartThreads / 64];
    __shared__ unsigned int wbond[kDirPartThreads / 64];
    __shared__ unsigned int base_s;
    unsigned int part = 0;
    for (unsigned int b = threadIdx.x; b < blockIdx.x; b += kDirPartThreads) part += p.blk_counts[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    const long long s = (long long)blockIdx.x * kDirPartThreads + threadIdx.x;
    bool bond = false, valid = s < p.n;
    if (valid) {
        int cls, site;
        double sign;
        dir_decode(p.idx[s], N, cls, site, sign);
        bond = cls == 0;
    }
    const unsigned long long m = __ballot(bond);
    if ((threadIdx.x & 63) == 0) wbond[threadIdx.x >> 6] = (unsigned int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int w = 0; w < kDirPartThreads / 64; ++w) t += red[w];
        base_s = t;
        if (blockIdx.x == gridDim.x - 1) {                     // the last block knows the total
            unsigned int tot = t;
            for (int w = 0; w < kDirPartThreads / 64; ++w) tot += wbond[w];
            p.counts[0] = tot;
        }
    }
    __syncthreads();
    if (!valid) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned int before_bond = 0;
    for (int k = 0; k < w; ++k) before_bond += wbond[k];
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned int rank_bond = base_s + before_bond + (unsigned int)__popcll(m & below);
    // diagonal samples before this one = samples before it (s) - bond samples before it
    if (bond) p.list[rank_bond] = (int)s;
    else p.list[p.n - 1 - (s - (long long)rank_bond)] = (int)s;
}

constexpr int kDirMaxN = 12;                                    // = kCsymMaxN: the diagonal class needs the complex symmetric route
constexpr int dir_bond_min_waves(int n) { return n <= 6 ? 4 : (n <= 8 ? 3 : 2); }

// bond class: lane per listed sample, the chain routes of tridiag_core.h.  MODE = kWeightsEnds / kWeightsAdjugate.
template <int N, int MODE>
__global__ __launch_bounds__(64, dir_bond_min_waves(N)) void mc_fid_dir_bond_kernel(const DirParams p) {
    __shared__ __attribute__((aligned(16))) double sctab[128];
    const int lane = threadIdx.x;
    if (rc::kTableSinCos) {
        const double2 ent = reinterpret_cast<const double2*>(g_sincos_table)[lane];
        reinterpret_cast<double2*>(sctab)[lane] = ent;
        __syncthreads();
    }
    const long long nb = (long long)p.counts[0];
    const long long i = (long long)blockIdx.x * 64 + lane;
    if ((long long)blockIdx.x * 64 >= nb) return;               // wave-uniform: the grid is sized for the worst case
    const bool live = i < nb;
    const long long s = live ? p.list[i] : 0;
    const long long c = (p.first + s) / p.K;
    const double* xg = p.ctrl + c * (N + 1);
    bool pad = false;
#pragma unroll
    for (int j = 0; j <= N; ++j) pad |= (xg[j] != xg[j]);
    int cls, site;
    double sign;
    dir_decode(p.idx[s], N, cls, site, sign);
    const double a = p.ab[2 * s], b = sign * p.ab[2 * s + 1];
    const int ja = 3 * site + 1, jb = 3 * site + 2;
    auto lg = [ja, jb, a, b](int j) { return (j % 3 == 0) ? 0.0 : ((j == ja) ? a : ((j == jb) ? b : 0.0)); };
    double f = 0.0;
    bool ok = true;
    int extra = 0;
    const bool run = live && !pad;
    if (run) ok = rc::chain_fidelity_fast<N, MODE>(xg, p.h0.diag, p.h0.off, lg, p.in, p.out, sctab, f, nullptr, &extra);
    if (extra && lane == 0) atomicAdd(&g_polish_tiles[blockIdx.x & 63u], 1ull);
    unsigned long long badmask = __ballot(run && !ok);
    if (badmask) {                                              // rare: the eigenvector-rows route for those lanes
        if (lane == 0) atomicAdd(&g_general_tiles, 1ull);
        const bool bad = (badmask >> lane) & 1ull;
        bool ok2 = true;
        if (bad) {
            double f2;
            ok2 = rc::chain_fidelity_fast<N, rc::kWeightsRows>(xg, p.h0.diag, p.h0.off, lg, p.in, p.out, sctab, f2);
            if (ok2) f = f2;
        }
        if (bad && !ok2) {                                      // (sweep cap of the rows-mode QL too: not observed) -> expm pass
            p.marked[atomicAdd(&p.counts[1], 1u)] = (int)s;
            f = __builtin_nan("");
        }
    }
    if (live) p.fid[s] = pad ? __builtin_nan("") : f;
}

// diagonal class: lane per listed sample, the complex symmetric QL route; list read from the back
template <int N>
__global__ __launch_bounds__(64, csym_min_waves(N)) void mc_fid_dir_diag_kernel(const DirParams p) {
    __shared__ __attribute__((aligned(16))) double sctab[128];
    const int lane = threadIdx.x;
    if (rc::kTableSinCos) {
        const double2 ent = reinterpret_cast<const double2*>(g_sincos_table)[lane];
        reinterpret_cast<double2*>(sctab)[lane] = ent;
        __syncthreads();
    }
    const long long nd = p.n - (long long)p.counts[0];
    const long long i = (long long)blockIdx.x * 64 + lane;
    if ((long long)blockIdx.x * 64 >= nd) return;
    if (i >= nd) return;
    const long long s = p.list[p.n - 1 - i];
    const long long c = (p.first + s) / p.K;
    const double* xg = p.ctrl + c * (N + 1);
    double x[N + 1];
    bool pad = false;
#pragma unroll
    for (int j = 0; j <= N; ++j) {
        x[j] = xg[j];
        pad |= (x[j] != x[j]);
    }
    double f = __builtin_nan("");
    if (!pad) {
        int cls, site;
        double sign;
        dir_decode(p.idx[s], N, cls, site, sign);
        const double a = p.ab[2 * s], mb = -p.ab[2 * s + 1];    // z[p,p] = a + ib, then overwritten by a - ib (noise_model.py:196-199)
        const int ja = 3 * site;
        double fv;
        const bool ok = rc::csym_fidelity<N>(x, p.h0.diag, p.h0.off, [ja, a](int j) { return (j == ja) ? a : 0.0; },
                                             [site, mb](int k) { return (k == site) ? mb : 0.0; }, p.in, p.out, sctab, fv);
        if (ok) f = fv;
        else p.marked[atomicAdd(&p.counts[1], 1u)] = (int)s;    // breakdown of a complex-orthogonal rotation -> expm pass
    }
    p.fid[s] = f;
}
