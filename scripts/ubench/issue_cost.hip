// Micro-benchmark: issue cost (cycles per wave-instruction) of the VALU ops the fidelity kernel is made of.
// One wave per SIMD (256-thread blocks, one block per CU), 8 independent chains per op, s_memtime stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

constexpr int ITER = 256;
#ifndef WPS
#define WPS 4
#endif
template <int OP>
__global__ __launch_bounds__(256) void bench(double* out, long long* cyc, double seed) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = seed + 0.001 * (threadIdx.x + 17 * j);
    float fa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = (float)a[j];
    const double c1 = seed * 0.5, c2 = seed * 0.25;
    __builtin_amdgcn_s_waitcnt(0);
    long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) a[j] = fma(a[j], c1, c2);
            if (OP == 1) a[j] = a[j] * c1;
            if (OP == 2) a[j] = a[j] + c1;
            if (OP == 3) a[j] = __builtin_amdgcn_rsq(a[j]);
            if (OP == 4) a[j] = __builtin_amdgcn_rcp(a[j]);
            if (OP == 5) fa[j] = __builtin_amdgcn_rsqf(fa[j]);
            if (OP == 6) { fa[j] = (float)a[j]; a[j] = a[j] + c1; }          // cvt_f32_f64 + add (subtract add cost)
            if (OP == 7) { a[j] = (double)fa[j]; fa[j] = fa[j] + 1.0f; }      // cvt_f64_f32 + f32 add
            if (OP == 8) a[j] = (a[j] > c1) ? a[(j + 1) & 7] : a[j] + c2;      // cmp_f64 + 2 cndmask + add
            if (OP == 9) a[j] = rint(a[j] * c1);                               // rndne + mul
            if (OP == 14) a[j] = ((threadIdx.x + it) & 8) ? a[(j + 1) & 7] : a[j] + c2;   // int cmp + 2 cndmask + add
            if (OP == 15) { a[j] = fma(a[j], c1, c2); a[j] = __builtin_copysign(a[j], a[(j+1)&7]); }  // fma + bfi
            if (OP == 10) a[j] = __builtin_amdgcn_sqrt(a[j]);
            if (OP == 11) fa[j] = fmaf(fa[j], 0.5f, 0.25f);
            if (OP == 12) a[j] = __builtin_amdgcn_ldexp(a[j], 1);
            if (OP == 13) a[j] = fmin(a[j], c1);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double acc = 0; 
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] + fa[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double* out; long long* cyc;
    const int blocks = 256 * WPS;
    CHK(hipMalloc(&out, blocks * 256 * 8)); CHK(hipMalloc(&cyc, blocks * 8));
    const char* names[] = {"v_fma_f64","v_mul_f64","v_add_f64","v_rsq_f64","v_rcp_f64","v_rsq_f32","cvt_f32_f64+add_f64","cvt_f64_f32+add_f32","cmp_f64+2cndmask","v_rndne_f64","v_sqrt_f64","v_fma_f32","v_ldexp_f64","v_min_f64","icmp+2cndmask+add_f64","fma_f64+bfi"};
    std::vector<long long> h(blocks);
#define RUN(OP) { hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25); hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25); CHK(hipDeviceSynchronize()); CHK(hipMemcpy(h.data(), cyc, blocks*8, hipMemcpyDeviceToHost)); double s=0; for(auto v: h) s+=v; s/=blocks; printf("%-22s %7.2f cycles per wave-instruction (per SIMD, %d waves/SIMD)\n", names[OP], s/(ITER*8.0*WPS), WPS); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
    return 0;
}
