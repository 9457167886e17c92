// Micro-benchmark: wall-time issue cost of v_fma_f32, v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 results per lane
// and instruction) and v_fma_f64 on gfx950 at 2 / 4 waves per SIMD, 8 independent chains per op, every CU busy.
// Question: would two samples per lane (packed fp32) halve the cost of the fp32 QL?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;
template <int OP, int WPS>
__global__ __launch_bounds__(256, WPS) void bench(float* out, float seed) {
    float a[8]; f2 v[8]; double q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = seed + 1e-3f * (threadIdx.x + 17 * j); v[j] = f2{a[j], a[j] * 0.5f}; q[j] = a[j]; }
    const float c1 = seed * 0.5f, c2 = seed * 0.25f; const f2 w1 = f2{c1, c2}, w2 = f2{c2, c1}; const double e1 = c1, e2 = c2;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(w1), "v"(w2));
            if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(w1));
            if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(w1));
            if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(q[j]) : "v"(e1), "v"(e2));
            if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
            if (OP == 6) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(q[j]) : "v"(e1), "v"(e2)); }
            if (OP == 7) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(w1), "v"(w2)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(q[j]) : "v"(e1), "v"(e2)); }
        }
    }
    float acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] + v[j].x + v[j].y + (float)q[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int OP, int WPS> void run(float* out, const char* name, int per) {
    const int blocks = 256 * WPS;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((bench<OP, WPS>), dim3(blocks), dim3(256), 0, 0, out, 1.25f);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((bench<OP, WPS>), dim3(blocks), dim3(256), 0, 0, out, 1.25f);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-28s waves/SIMD %d: %.3f ns per wave-instruction per SIMD\n", name, WPS, ms * 1e6 / (ITER * 8.0 * per * WPS));
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0, 4>(out, "v_fma_f32", 1); run<1, 4>(out, "v_pk_fma_f32", 1); run<2, 4>(out, "v_pk_mul_f32", 1); run<3, 4>(out, "v_pk_add_f32", 1);
    run<4, 4>(out, "v_fma_f64", 1); run<5, 4>(out, "v_mul_f32", 1); run<6, 4>(out, "fma_f32 + fma_f64 (per instr)", 2); run<7, 4>(out, "pk_fma_f32 + fma_f64 (per instr)", 2);
    run<0, 2>(out, "v_fma_f32", 1); run<1, 2>(out, "v_pk_fma_f32", 1); run<4, 2>(out, "v_fma_f64", 1); run<7, 2>(out, "pk_fma_f32 + fma_f64 (per instr)", 2);
    run<0, 1>(out, "v_fma_f32", 1); run<1, 1>(out, "v_pk_fma_f32", 1); run<4, 1>(out, "v_fma_f64", 1);
    return 0;
}
