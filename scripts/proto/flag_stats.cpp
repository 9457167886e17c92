// Host build of the chain kernel's per-sample arithmetic with the "left the one-step path" flag PER SAMPLE (on the host a wave is
// one sample): which samples flag their tile on the device, and how they cluster (scripts/proto/flag_stats.py; DESIGN.md 8 xii).
#define RC_FLAG_STATS 1
#include "../../code-robchar_amd/csrc/tridiag_core.h"
#include <string.h>
// what the stepping path starts from, per flagged sample: the wanted-eigenvalue mask of the shipped rule, the largest first step
// and the iterates after the first step
static unsigned g_roots; static double g_maxd; static double g_lam[16]; static int g_hit;
void rc_flag_stats_hook(int n, unsigned roots, double maxd, const double* lam) { g_roots = roots; g_maxd = maxd; memcpy(g_lam, lam, n * sizeof(double)); g_hit = 1; }
static const double g_sctab[128] = {RC_SINCOS_TABLE_VALUES};
template <int N, int MODE>
static void run(const double* ctrl, const double* h0d, const double* h0o, const double* draws, long long C, long long K, int in, int out, double* fid, int* flag,
                unsigned* roots, double* maxd, double* lam) {
    for (long long c = 0; c < C; ++c) for (long long k = 0; k < K; ++k) {
        const double* g = draws + (c * K + k) * 3 * N;
        double f; int extra = 0; g_hit = 0;
        bool ok = rc::chain_fidelity_fast<N, MODE>(ctrl + c * (N + 1), h0d, h0o, [g](int j) { return g[j]; }, in, out, g_sctab, f, nullptr, &extra);
        fid[c*K+k] = f; if (g_hit) { roots[c*K+k] = g_roots; maxd[c*K+k] = g_maxd; memcpy(lam + (c*K+k)*N, g_lam, N * sizeof(double)); } else roots[c*K+k] = 0;
        flag[c*K+k] = extra;          // 0: one-step path; 1 + stepping iterations otherwise
    }
}
extern "C" int flags(int N, const double* ctrl, const double* h0d, const double* h0o, const double* draws, long long C, long long K, int in, int out, double* fid, int* flag,
                     unsigned* roots, double* maxd, double* lam) {
    if (N == 7) run<7, rc::kWeightsEnds>(ctrl, h0d, h0o, draws, C, K, in, out, fid, flag, roots, maxd, lam);
    else if (N == 10) run<10, rc::kWeightsEnds>(ctrl, h0d, h0o, draws, C, K, in, out, fid, flag, roots, maxd, lam);
    else return -1;
    return 0;
}
