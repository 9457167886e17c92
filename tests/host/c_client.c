/* A plain C99 client of the C ABI (include/robchar_hip.h) - no Python, no torch, no C++: what a maintainer of another host
 * language binds.  Reads a problem from stdin, calls the blocking fidelity entry with HOST buffers and the
 * reduction, prints the results with 17 significant digits.  tests/test_cabi_symbols.py compiles it (-std=c99 -pedantic) and
 * links it against the library on the CPU; tests/test_gpu_chain.py runs it on the GPU box and compares with the oracle.
 *   stdin:  N in out ring C K   then C*(N+1) controller values, then C*K*N*3 draws
 *   stdout: "version V devices D" / C*K fidelities / per controller: rim1 std min q(thr 0.95)                               */
#include <stdio.h>
#include <stdlib.h>
#include "robchar_hip.h"

int main(void) {
    int N, in, out, ring;
    long long C, K, i;
    if (scanf("%d %d %d %d %lld %lld", &N, &in, &out, &ring, &C, &K) != 6) return 2;
    double* ctrl = (double*)malloc(sizeof(double) * (size_t)(C * (N + 1)));
    double* draws = (double*)malloc(sizeof(double) * (size_t)(C * K * N * 3));
    double* fid = (double*)malloc(sizeof(double) * (size_t)(C * K));
    double* rim1 = (double*)malloc(sizeof(double) * (size_t)(3 * C));
    double* sd = (double*)malloc(sizeof(double) * (size_t)(3 * C));
    double* mn = (double*)malloc(sizeof(double) * (size_t)(3 * C));
    double* q = (double*)malloc(sizeof(double) * (size_t)(3 * C));
    if (!ctrl || !draws || !fid || !rim1 || !sd || !mn || !q) return 2;
    for (i = 0; i < C * (N + 1); ++i) if (scanf("%lf", &ctrl[i]) != 1) return 2;
    for (i = 0; i < C * K * N * 3; ++i) if (scanf("%lf", &draws[i]) != 1) return 2;
    printf("version %d devices %d\n", rc_version(), rc_device_count());
    int rc = rc_mc_fidelity_f64(0, N, in, out, NULL, NULL, ring, ctrl, draws, C, K, fid);
    if (rc != RC_OK) { fprintf(stderr, "rc_mc_fidelity_f64: %d %s\n", rc, rc_last_error()); return 1; }
    const double thr[1] = {0.95};
    rc = rc_reduce_f64(0, fid, C, K, thr, 1, 0.0, rim1, sd, mn, q, NULL);
    if (rc != RC_OK) { fprintf(stderr, "rc_reduce_f64: %d %s\n", rc, rc_last_error()); return 1; }
    for (i = 0; i < C * K; ++i) printf("%.17g\n", fid[i]);
    for (i = 0; i < C; ++i) printf("%.17g %.17g %.17g %.17g\n", rim1[i], sd[i], mn[i], q[i]);
    /* an argument error comes back as a code and a message, not as a crash */
    rc = rc_mc_fidelity_f64(0, N, N, out, NULL, NULL, ring, ctrl, draws, C, K, fid);
    printf("bad-argument call: %d (%s)\n", rc, rc == RC_OK ? "?" : rc_last_error());
    free(ctrl); free(draws); free(fid); free(rim1); free(sd); free(mn); free(q);
    return 0;
}
