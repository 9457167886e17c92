// Micro-benchmark: fp64 VALU issue rate vs number of VGPR source operands (4 waves/SIMD, 8 independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef WPS
#define WPS 4
#endif
constexpr int ITER = 512;
template <int OP>
__global__ __launch_bounds__(256) void bench(double* out, long long* cyc, double seed) {
    double a[8], b[8], c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = seed + 0.001 * (threadIdx.x + 17 * j); b[j] = 1.0 + 1e-9 * (threadIdx.x + j); c[j] = 1e-3 * (threadIdx.x + 3 * j); }
    const double s1 = seed * 0.5, s2 = seed * 0.25;
    long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 2
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) a[j] = fma(a[j], s1, s2);           // 1 VGPR src
            if (OP == 1) a[j] = fma(a[j], b[j], s2);         // 2 VGPR src
            if (OP == 2) a[j] = fma(a[j], b[j], c[j]);       // 3 VGPR src
            if (OP == 3) a[j] = a[j] * b[j];                 // mul 2 VGPR
            if (OP == 4) a[j] = a[j] + c[j];                 // add 2 VGPR
            if (OP == 5) a[j] = fma(b[j], c[j], a[j]);       // fmac form
            if (OP == 6) a[j] = fma(a[j], b[(j + 1) & 7], c[(j + 3) & 7]);   // 3 VGPR src, different regs
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] + b[j] + c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; const int blocks = 256 * WPS;
    hipMalloc(&out, blocks * 256 * 8); hipMalloc(&cyc, blocks * 8);
    std::vector<long long> h(blocks);
    const char* names[] = {"fma 1 VGPR src (2 SGPR)", "fma 2 VGPR src", "fma 3 VGPR src", "mul 2 VGPR src", "add 2 VGPR src", "fmac (b*c+a)", "fma 3 VGPR src (mixed regs)"};
#define RUN(OP) { for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25); hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, blocks*8, hipMemcpyDeviceToHost); double s=0; for(auto v: h) s+=v; s/=blocks; printf("%-30s %6.2f cycles per wave-instruction per SIMD (%d waves/SIMD)\n", names[OP], s/(ITER*8.0*WPS), WPS); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}
