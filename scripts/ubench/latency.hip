// Micro-benchmark: dependent-issue latency (cycles between two dependent wave-instructions), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int ITER = 2048;
template <int OP>
__global__ __launch_bounds__(256) void bench(double* out, long long* cyc, double seed, int m) {
    double a = seed + 0.001 * threadIdx.x, b = seed * 0.75;
    const double c1 = seed * 0.5, c2 = seed * 0.25;
    long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 8
    for (int it = 0; it < ITER; ++it) {
        if (OP == 0) a = fma(a, c1, c2);
        if (OP == 1) a = a * c1;
        if (OP == 2) a = a + c1;
        if (OP == 3) a = __builtin_amdgcn_rsq(a) + c1;          // rsq + add (subtract add latency)
        if (OP == 4) { if (a > c1) a = a * c2; a = a + c2; }       // cmp -> exec-mask branch region -> add
        if (OP == 5) { a = (a > b) ? c1 : a; a = a + c2; }         // cmp -> cndmask x2 -> add
        if (OP == 6) { float f = (float)a; f = __builtin_amdgcn_rsqf(f); a = (double)f + c1; }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; const int blocks = 256;
    hipMalloc(&out, blocks * 256 * 8); hipMalloc(&cyc, blocks * 8);
    std::vector<long long> h(blocks);
    const char* names[] = {"fma_f64 -> fma_f64", "mul_f64 -> mul_f64", "add_f64 -> add_f64", "rsq_f64 + add_f64", "cmp + branch(execz) region(mul) + add", "cmp + 2 cndmask + add", "cvt_f32 + rsq_f32 + cvt_f64 + add"};
#define RUN(OP) { for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25, 3); hipDeviceSynchronize(); hipMemcpy(h.data(), cyc, blocks*8, hipMemcpyDeviceToHost); double s=0; for(auto v: h) s+=v; s/=blocks; printf("%-40s %7.2f cycles per iteration (1 wave/SIMD, dependent chain)\n", names[OP], s/ITER); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}
