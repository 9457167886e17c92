// Accuracy of the v_rsq_f64 / v_rcp_f64 seeds and of the refined sqrt/rsqrt used by the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../code-robchar_amd/csrc/tridiag_core.h"
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double r, inv; rc::sqrt_rsqrt(x[i], r, inv);
    o[i * 6 + 0] = __builtin_amdgcn_rsq(x[i]); o[i * 6 + 1] = __builtin_amdgcn_rcp(x[i]);
    o[i * 6 + 2] = r; o[i * 6 + 3] = inv; o[i * 6 + 4] = rc::sqrt_fast(x[i]); o[i * 6 + 5] = rc::rcp_fast(x[i]);
}
int main() {
    const int n = 1 << 20; std::vector<double> x(n), o(n * 6);
    for (int i = 0; i < n; ++i) x[i] = std::exp((double(i) / n - 0.5) * 80.0) * (1.0 + 0.37 * std::sin(i * 1.2345));
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 48);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n); hipDeviceSynchronize();
    hipMemcpy(o.data(), dout, n * 48, hipMemcpyDeviceToHost);
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        long double xs = x[i], sq = sqrtl(xs);
        long double ref[6] = {1 / sq, 1 / xs, sq, 1 / sq, sq, 1 / xs};
        for (int j = 0; j < 6; ++j) { double r = std::fabs((double)((o[i * 6 + j] - ref[j]) / ref[j])); if (r > e[j]) e[j] = r; }
    }
    printf("max rel err: v_rsq_f64 %.3e  v_rcp_f64 %.3e | sqrt_rsqrt: root %.3e inv %.3e | sqrt_fast %.3e rcp_fast %.3e\n", e[0], e[1], e[2], e[3], e[4], e[5]);
}
