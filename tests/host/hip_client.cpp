// A HIP host program on the enqueue-only entries of the C ABI (include/robchar_hip.h) - no Python, no torch: its own device
// buffers, its own two streams, counter-based draws generated on the device, the fidelity kernel and the reduction enqueued
// behind them, nothing synchronised until the results are copied back.  This is the shape of a one-process-per-GPU integrator
// (INTEGRATION.md C).  tests/test_gpu_chain.py builds it with hipcc and compares its output with the Python layer's / the oracle.
//   argv: N in out C K seed sigma      stdin: C*(N+1) controller values
//   stdout: C*K fidelities (stream A), then per controller rim1 std min q(0.95), then the same fidelities from stream B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "robchar_hip.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define RCCHK(x) do { int r_ = (x); if (r_ != RC_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, rc_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc != 8) return 2;
    const int N = atoi(argv[1]), in = atoi(argv[2]), out = atoi(argv[3]);
    const long long C = atoll(argv[4]), K = atoll(argv[5]);
    const unsigned long long seed = strtoull(argv[6], nullptr, 10);
    const double sigma = atof(argv[7]);
    std::vector<double> ctrl((size_t)C * (N + 1));
    for (double& v : ctrl) if (scanf("%lf", &v) != 1) return 2;
    const long long nd = C * K * N * 3;
    double *d_ctrl, *d_draw[2], *d_fid[2], *d_rim, *d_std, *d_min, *d_q;
    HIPCHK(hipMalloc(&d_ctrl, ctrl.size() * sizeof(double)));
    for (int s = 0; s < 2; ++s) {
        HIPCHK(hipMalloc(&d_draw[s], (size_t)nd * sizeof(double)));
        HIPCHK(hipMalloc(&d_fid[s], (size_t)C * K * sizeof(double)));
    }
    HIPCHK(hipMalloc(&d_rim, 3 * C * sizeof(double)));
    HIPCHK(hipMalloc(&d_std, 3 * C * sizeof(double)));
    HIPCHK(hipMalloc(&d_min, 3 * C * sizeof(double)));
    HIPCHK(hipMalloc(&d_q, 3 * C * sizeof(double)));
    hipStream_t st[2];
    HIPCHK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    HIPCHK(hipMemcpy(d_ctrl, ctrl.data(), ctrl.size() * sizeof(double), hipMemcpyHostToDevice));
    const double thr[1] = {0.95};
    // two streams, the same work on each, enqueued interleaved: draws -> fidelities (-> reduction on stream A)
    for (int s = 0; s < 2; ++s) RCCHK(rc_draws_philox_f64_async(0, st[s], seed, 0ull, nd, sigma, d_draw[s]));
    for (int s = 0; s < 2; ++s)
        RCCHK(rc_mc_fidelity_f64_async(0, st[s], RC_KERNEL_AUTO, N, in, out, nullptr, nullptr, 0, d_ctrl, d_draw[s], C, K, d_fid[s]));
    RCCHK(rc_reduce_f64_async(0, st[0], d_fid[0], C, K, thr, 1, 0.0, d_rim, d_std, d_min, d_q, nullptr));
    std::vector<double> fid[2] = {std::vector<double>((size_t)C * K), std::vector<double>((size_t)C * K)};
    std::vector<double> rim(3 * C), sd(3 * C), mn(3 * C), q(3 * C);
    for (int s = 0; s < 2; ++s) HIPCHK(hipMemcpyAsync(fid[s].data(), d_fid[s], fid[s].size() * sizeof(double), hipMemcpyDeviceToHost, st[s]));
    HIPCHK(hipMemcpyAsync(rim.data(), d_rim, rim.size() * sizeof(double), hipMemcpyDeviceToHost, st[0]));
    HIPCHK(hipMemcpyAsync(sd.data(), d_std, sd.size() * sizeof(double), hipMemcpyDeviceToHost, st[0]));
    HIPCHK(hipMemcpyAsync(mn.data(), d_min, mn.size() * sizeof(double), hipMemcpyDeviceToHost, st[0]));
    HIPCHK(hipMemcpyAsync(q.data(), d_q, q.size() * sizeof(double), hipMemcpyDeviceToHost, st[0]));
    HIPCHK(hipStreamSynchronize(st[0]));
    HIPCHK(hipStreamSynchronize(st[1]));
    for (double v : fid[0]) printf("%.17g\n", v);
    for (long long c = 0; c < C; ++c) printf("%.17g %.17g %.17g %.17g\n", rim[c], sd[c], mn[c], q[c]);
    for (double v : fid[1]) printf("%.17g\n", v);
    return 0;
}
