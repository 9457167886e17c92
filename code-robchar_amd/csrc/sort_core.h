// Merge-path building blocks of the row sort (ECDF) - plain C++ so that the CPU unit tests can run the exact index
// logic (tests/test_host_core.py); the product only ever runs it inside sort_rows_merge_kernel.
//
// A row of n = 16 * T elements (padded with +inf) is sorted by T threads: each sorts its 16 elements in registers,
// then log2(T) merge levels double the run length L = 16, 32, ...  In a level, thread t produces the 16 outputs
// [16 t, 16 t + 16) of the merge of the run pair that contains them: a binary search along the output diagonal
// (merge path; Green, McColl, Bader 2012) gives its starting positions in the two runs, then 16 sequential steps.
// The buffer is addressed through pad(e) = e + (e >> 4) (one spare slot per 16: the 16-element writes of
// neighbouring threads fall into different LDS banks).
#pragma once

#if defined(__HIPCC__)
#define RC_SD __host__ __device__ __forceinline__
#else
#define RC_SD inline
#endif

namespace rcs {

RC_SD int pad(int e) { return e + (e >> 4); }

// Number of elements taken from run A (buf[a0 .. a0+na)) among the first `diag` outputs of merge(A, B), B =
// buf[b0 .. b0+nb); ties take A first.
RC_SD int merge_path(const double* buf, int a0, int na, int b0, int nb, int diag) {
    int lo = diag > nb ? diag - nb : 0;
    int hi = diag < na ? diag : na;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (buf[pad(a0 + mid)] <= buf[pad(b0 + diag - 1 - mid)]) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// The 16 outputs [16 t, 16 t + 16) of one merge level with run length L over n elements.
RC_SD void merge_level16(const double* buf, int n, int L, int t, double (&out)[16]) {
    const int o0 = 16 * t;
    const int start = (o0 / (2 * L)) * (2 * L);
    const int na = (n - start < L) ? (n - start) : L;
    const int b0 = start + na;
    const int nb = (n - b0 < L) ? (n - b0) : L;
    const int diag = o0 - start;
    int i = merge_path(buf, start, na, b0, nb, diag);
    int j = diag - i;
    const double kInf = __builtin_inf();
    double ka = (i < na) ? buf[pad(start + i)] : kInf;
    double kb = (j < nb) ? buf[pad(b0 + j)] : kInf;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int o = 0; o < 16; ++o) {
        const bool take_a = (j >= nb) || (i < na && ka <= kb);
        out[o] = take_a ? ka : kb;
        // one refill read per step whichever run advanced (no divergent branches on the device)
        i += take_a ? 1 : 0;
        j += take_a ? 0 : 1;
        const int idx = take_a ? (start + i) : (b0 + j);
        const bool inside = take_a ? (i < na) : (j < nb);
        const double nxt = inside ? buf[pad(inside ? idx : 0)] : kInf;
        ka = take_a ? nxt : ka;
        kb = take_a ? kb : nxt;
    }
}

}  // namespace rcs
