// Micro-benchmark: the QL rotation body of the fidelity kernel (eigenvalue-only and with the two eigenvector rows)
// as a tight loop, to separate the arithmetic's own issue rate from everything else in the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef STAGGER
#define STAGGER 0
#endif
#include "../../code-robchar_amd/csrc/tridiag_core.h"
constexpr int ITER = 4096;
#ifndef UNROLL
#define UNROLL 4
#endif
template <int VEC, int WPS>
__global__ __launch_bounds__(256, WPS) void bench(double* out, long long* cyc, double seed) {
    double d0 = seed + 1e-3 * threadIdx.x, d1 = seed * 0.7 - 1e-3 * threadIdx.x, e = 1.0 + 1e-4 * threadIdx.x;
    double g = 0.3 + 1e-5 * threadIdx.x, sn = 0.6, cs = 0.8, p = 0.01;
    double z0 = 0.1, z1 = 0.2, y0 = 0.3, y1 = 0.4;
    long long t0 = __builtin_amdgcn_s_memtime(); long long r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const int skew = STAGGER ? (int)((blockIdx.x * 4 + (threadIdx.x >> 6)) % 61) * 3 : 0;   // waves start at different trip counts
#pragma unroll UNROLL
    for (int it = skew; it < ITER + skew; ++it) {
        double f = sn * e; const double b = cs * e; const double gn = g + 1e-150;
        double r, rinv; rc::sqrt_rsqrt(fma(f, f, gn * gn), r, rinv);
        e = r * 0.5 + 0.5; sn = f * rinv; cs = gn * rinv; g = d1 - p;
        r = fma(d0 - g, sn, 2.0 * cs * b); p = sn * r; d1 = g + p; g = fma(cs, r, -b);
        d0 = d0 * 0.999 + 1e-3;
        if (VEC) { f = z1; z1 = fma(sn, z0, cs * f); z0 = fma(cs, z0, -sn * f); f = y1; y1 = fma(sn, y0, cs * f); y0 = fma(cs, y0, -sn * f); }
    }
    long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + e + g + sn + cs + p + z0 + z1 + y0 + y1;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int VEC, int WPS> void run(double* out, long long* cyc, int ninstr) {
    const int blocks = 256 * WPS; std::vector<long long> h(2 * blocks);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((bench<VEC, WPS>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
    double s = 0, rt = 0; for (int i = 0; i < blocks; ++i) { s += h[2 * i]; rt += h[2 * i + 1]; } s /= blocks; rt /= blocks;
    printf("rows=%d waves/SIMD %d: %.1f ticks per rotation per wave, %.2f ticks per rotation per SIMD (~%d VALU each -> %.2f ticks/instr), clock %.3f GHz, %.2f G rotations/s chip\n",
           VEC, WPS, s / ITER, s / ITER / WPS, ninstr, s / ITER / WPS / ninstr, s / (rt / 100e6) / 1e9, 1024.0 * WPS * 64 * ITER / (rt / 100e6) / 1e9);
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 256 * 8 * 256 * 8); hipMalloc(&cyc, 256 * 8 * 16);
    run<1, 5>(out, cyc, 36); run<0, 8>(out, cyc, 28);
    return 0;
}
