// Per-controller reductions (reduce_kernel, reduce_rows_wave_kernel, rim_p_kernel) and the row sort for the ECDF
// (merge-path sort in LDS, chunked bitonic network for long rows).
//
// Part of ONE translation unit: this file is #included by robchar_hip.hip INSIDE its anonymous namespace (after the
// shared parameter structs); it is not a stand-alone header.
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
// THREADS: 512 for rows the cache needs that many for (K > 8192) and for the many-threshold variant; 128 / 256 for rows of up to
// 4096 / 8192 values (late round 4) - a 512-thread workgroup is a latency chain (load, reduce, barrier, second pass) of which
// two fit a CU: 11 000 rows of 2049 values took 325 us where the wave-per-row kernel below takes 44 us for rows of 2048
// (profiles/r04_reduce_sweep.txt); narrower workgroups keep ten rows in flight per CU.  The choice depends on K alone, so a
// row's result does not depend on how many rows are reduced with it.
// CACHE: values a thread keeps in registers.
template <int NQ, int THREADS = kRedThreads, int CACHE = kRedCache>
__global__ __launch_bounds__(THREADS) void reduce_kernel(const RedParams p) {
    constexpr int kWaves = THREADS / 64;
    constexpr int NV = 5;                                  // sum[3], min, nan
    constexpr int NC = 3 * NQ;                             // threshold counts cnt[3][NQ]: integers (exact, half the registers)
    constexpr int kCache = (NQ <= 2) ? CACHE : 4;           // the many-threshold variant has no registers to spare
    __shared__ double part[kWaves][NV];
    __shared__ unsigned int partc[kWaves][NC > 0 ? NC : 1];
    __shared__ double part2[kWaves][3];
    const long long c = blockIdx.x;
    const double* row = p.fid + c * p.K;
    const double K = (double)p.K;
    const bool cached = p.K <= (long long)kCache * THREADS;
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
            const long long k = (long long)i * THREADS + threadIdx.x;
            val[i] = (k < p.K) ? row[k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < kCache; ++i)
            if ((long long)i * THREADS + threadIdx.x < p.K) pass1(val[i]);
    } else {
        // long rows (K > kCache * THREADS): 8 loads in flight per thread, then the accumulation
        long long k = threadIdx.x;
        for (; k + 7 * THREADS < p.K; k += 8 * THREADS) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[k + u * THREADS];
#pragma unroll
            for (int u = 0; u < 8; ++u) pass1(v[u]);
        }
        for (; k < p.K; k += THREADS) pass1(row[k]);
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
        for (int w = 1; w < kWaves; ++w) r = (i == 3) ? fmin(r, part[w][i]) : r + part[w][i];
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
                if ((long long)i * THREADS + threadIdx.x < p.K) pass2(val[i]);
        } else {
            long long k = threadIdx.x;
            for (; k + 7 * THREADS < p.K; k += 8 * THREADS) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = row[k + u * THREADS];
#pragma unroll
                for (int u = 0; u < 8; ++u) pass2(v[u]);
            }
            for (; k < p.K; k += THREADS) pass2(row[k]);
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
            for (int w = 1; w < kWaves; ++w) r += part2[w][v];
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
                        for (int w = 0; w < kWaves; ++w) n += partc[w][v * NQ + j];
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
