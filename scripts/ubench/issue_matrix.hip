// Micro-benchmark: fp64 FMA throughput per SIMD as a function of waves/SIMD and independent chains per wave.
// All operands in VGPRs (a = fma(a, b, c)), runtime-launched with W waves per SIMD (blocks of 256 threads, W per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int ITER = 65536;
template <int ILP>
__global__ __launch_bounds__(256) void bench(double* out, long long* cyc, double seed) {
    double a[ILP], b[ILP], c[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) { a[j] = seed + 0.001 * (threadIdx.x + 17 * j); b[j] = 1.0 + 1e-9 * (threadIdx.x + j); c[j] = 1e-3 * (threadIdx.x + 3 * j); }
    long long t0 = __builtin_amdgcn_s_memtime();
    long long r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) a[j] = fma(a[j], b[j], c[j]);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double acc = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) acc += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int ILP> void run(int wps, double* out, long long* cyc) {
    const int blocks = 256 * wps; std::vector<long long> h(2 * blocks);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(bench<ILP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
    double s = 0, rt = 0; for (int i = 0; i < blocks; ++i) { s += h[2 * i]; rt += h[2 * i + 1]; } s /= blocks; rt /= blocks;
    const double secs = rt / 100e6;
    printf("waves/SIMD %d  chains/wave %d : %.2f ticks per FMA per SIMD | clock %.3f GHz | %.2f T lane-FMA/s chip-wide = %.1f TFLOP/s fp64\n", wps, ILP, s / (ITER * (double)ILP * wps), s / secs / 1e9, 1024.0 * wps * 64.0 * ITER * ILP / secs / 1e12, 2 * 1024.0 * wps * 64.0 * ITER * ILP / secs / 1e12);
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 256 * 8 * 256 * 8); hipMalloc(&cyc, 256 * 8 * 16);
    for (int w : {4, 5, 8}) { run<1>(w, out, cyc); run<4>(w, out, cyc); }
    return 0;
}
