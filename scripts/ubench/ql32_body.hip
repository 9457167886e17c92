// Micro-benchmark: fp32 eigenvalue-only QL inner step, three formulations, as tight loops at 4 waves per SIMD on every CU:
//   0 implicit QL rotation (tridiag_ql_f32: 17 VALU + v_rsq_f32)      1 rational QL (Reinsch, EISPACK tqlrat: squares of
//   the couplings, two v_rcp_f32, no square root)                      2 Pal-Walker-Kahan (LAPACK dsterf), two v_rcp_f32
// Question: is a root-free step cheaper in WALL time than the rotation, given that v_rcp/v_rsq issue at a quarter rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int ITER = 8192;
template <int BODY, int WPS>
__global__ __launch_bounds__(256, WPS) void bench(float* out, long long* cyc, float seed) {
    float d0 = seed + 1e-3f * threadIdx.x, d1 = seed * 0.7f - 1e-3f * threadIdx.x, e = 1.0f + 1e-4f * threadIdx.x;
    float g = 0.3f + 1e-5f * threadIdx.x, sn = 0.6f, cs = 0.8f, p = 0.01f, h = 0.4f, s = 0.3f, gamma = 0.2f, sigma = 0.11f;
    long long r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 4
    for (int it = 0; it < ITER; ++it) {
        if (BODY == 0) {
            const float f = sn * e, b = cs * e, gn = g + 1e-15f;
            const float hh = fmaf(f, f, gn * gn);
            const float rinv = __builtin_amdgcn_rsqf(hh);
            e = hh * rinv * 0.5f + 0.5f;                               // (stand-in for the store of e[i+1]; keeps e ~ 1)
            sn = f * rinv; cs = gn * rinv; g = d1 - p;
            const float r = fmaf(d0 - g, sn, 2.0f * cs * b);
            p = sn * r; d1 = g + p; g = fmaf(cs, r, -b);
            d0 = d0 * 0.999f + 1e-3f;
        } else if (BODY == 1) {                                        // e = SQUARED coupling
            const float pp = g * h;
            const float r = pp + e;
            const float e_next = s * r;
            const float rinv = __builtin_amdgcn_rcpf(r);
            s = e * rinv;
            d1 = fmaf(s, h + d0, h);
            float gg = fmaf(-e, __builtin_amdgcn_rcpf(g), d0);
            gg = (gg == 0.0f) ? 1e-30f : gg;
            g = gg;
            h = g * pp * rinv;
            e = e_next * 0.5f + 0.5f;
            d0 = d0 * 0.999f + 1e-3f;
        } else {
            const float r = p + e;
            const float e_next = s * r;
            const float rinv = __builtin_amdgcn_rcpf(r);
            const float oldc = cs;
            cs = p * rinv; s = e * rinv;
            const float oldgam = gamma;
            gamma = fmaf(cs, d0 - sigma, -s * oldgam);
            d1 = oldgam + (d0 - gamma);
            const float pn = gamma * gamma * r * __builtin_amdgcn_rcpf(p);
            p = (cs != 0.0f) ? pn : oldc * e;
            e = e_next * 0.5f + 0.5f;
            d0 = d0 * 0.999f + 1e-3f;
        }
    }
    long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + e + g + sn + cs + p + h + s + gamma;
    if (threadIdx.x == 0) cyc[blockIdx.x] = r1 - r0;
}
template <int BODY, int WPS> void run(float* out, long long* cyc, const char* name) {
    const int blocks = 256 * WPS; std::vector<long long> hst(blocks);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((bench<BODY, WPS>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25f);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((bench<BODY, WPS>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25f);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-22s waves/SIMD %d: %.3f ms per launch = %.2f ns per step per SIMD-wave-slot, %.1f G steps/s chip\n", name, WPS, ms,
           ms * 1e6 / ITER / WPS, 1024.0 * WPS * 64 * ITER / (ms * 1e-3) / 1e9);
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 4>(out, cyc, "implicit QL rotation"); run<1, 4>(out, cyc, "rational QL (tqlrat)"); run<2, 4>(out, cyc, "PWK (dsterf)");
    }
    return 0;
}
