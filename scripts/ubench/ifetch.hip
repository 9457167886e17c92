// Micro-benchmark: does straight-line (fully unrolled) fp64 code issue as fast as a tight loop?
// Same 8 independent FMA chains; body unrolled U times inside an outer loop so that total work is equal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef WPS
#define WPS 4
#endif
constexpr int TOTAL = 1 << 14;   // FMAs per chain
template <int U>
__global__ __launch_bounds__(256) void bench(double* out, long long* cyc, double seed) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = seed + 0.001 * (threadIdx.x + 17 * j);
    const double c1 = seed * 0.5, c2 = seed * 0.25;
    long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
    for (int it = 0; it < TOTAL / U; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(a[j], c1, c2);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; const int blocks = 256 * WPS;
    hipMalloc(&out, blocks * 256 * 8); hipMalloc(&cyc, blocks * 8);
    std::vector<long long> h(blocks);
#define RUN(U) { for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(bench<U>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25); hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, blocks*8, hipMemcpyDeviceToHost); double s=0; for(auto v: h) s+=v; s/=blocks; printf("unroll %5d (body %6d B): %.2f cycles per wave-instruction per SIMD at %d waves/SIMD\n", U, U*8*8, s/(TOTAL*8.0*WPS), WPS); }
    RUN(1) RUN(8) RUN(64) RUN(256) RUN(1024) RUN(4096)
    return 0;
}
