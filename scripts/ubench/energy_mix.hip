// Micro-benchmark: sustained throughput of VALU instruction mixes on CHAOTIC data at full occupancy.  The fidelity
// kernel runs at the socket power cap, so at equal issue cost the sustained rate of a mix is a proxy for its energy
// per instruction.  Base: 8 independent chaotic chains x <- 1.9 - x*x (one v_fma_f64 each, random mantissas);
// variants add ONE extra instruction per chain step (result kept alive through inline asm).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
constexpr int ITER = 40000;
template <int OP>
__global__ __launch_bounds__(256, 5) void mix(double* out, double seed) {
    double a[8], u[8];
    float fa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = seed + 1e-3 * ((threadIdx.x * 8 + j) % 977) + 1e-6 * blockIdx.x;
        u[j] = 0.0; fa[j] = (float)a[j];
    }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP != 9) a[j] = fma(-a[j], a[j], 1.9);
            if (OP == 1) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(u[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]));
            if (OP == 2) asm volatile("v_add_f64 %0, %1, %2" : "=v"(u[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]));
            if (OP == 3) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(u[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]), "v"(a[(j + 2) & 7]));
            if (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(fa[j]) : "v"(fa[j]), "v"(fa[(j + 1) & 7]), "v"(fa[(j + 2) & 7]));
            if (OP == 5) asm volatile("v_mov_b64 %0, %1" : "=v"(u[j]) : "v"(a[j]));
            if (OP == 6) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(fa[j]) : "v"(fa[j]), "v"(fa[(j + 1) & 7]));
            if (OP == 7) asm volatile("v_rsq_f64 %0, %1" : "=v"(u[j]) : "v"(a[j]));
            if (OP == 8) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" ::: "memory");
            if (OP == 9) {                                  // fp32 chaotic chain alone
                fa[j] = fmaf(-fa[j], fa[j], 1.9f);
            }
        }
    }
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] + u[j] + fa[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    double* out;
    const int blocks = 256 * 5;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    const char* names[] = {"fma64 chain alone", "+ v_mul_f64", "+ v_add_f64", "+ v_fma_f64", "+ v_fma_f32", "+ v_mov_b64",
                           "+ v_cndmask_b32", "+ v_rsq_f64", "+ 4 s_nop", "fma32 chain alone"};
    double base = 0;
#define RUN(OP) { for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mix<OP>, dim3(blocks), dim3(256), 0, 0, out, 0.3); CHK(hipDeviceSynchronize()); \
        auto t0 = std::chrono::steady_clock::now(); const int L = 12; for (int w = 0; w < L; ++w) hipLaunchKernelGGL(mix<OP>, dim3(blocks), dim3(256), 0, 0, out, 0.3); \
        CHK(hipDeviceSynchronize()); double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / L; \
        double per = dt / (ITER * 8.0 * 5) * 1e9; if (OP == 0) base = per; \
        printf("%-20s %8.3f ms/launch  %6.3f ns per chain step per SIMD  (extra %+6.3f ns = %5.2f x an fma64)\n", names[OP], dt * 1e3, per, per - base, (per - base) / base); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(0)
    return 0;
}
