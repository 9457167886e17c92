// librobchar_hip.so, second translation unit: the LARGEST kernel instantiations - chains of 17 .. 24 spins (general adjugate mode,
// one wave per SIMD) and rings of 11 .. 16 spins (mixed route + repair kernel, all-fp64 route; folded band reduction) - compiled
// in parallel with robchar_hip.hip (`make -j`): the two together take the wall time the first one took alone until round 5.
// Same kernel templates (k_fidelity_chain.inc.h), instantiated here only for these sizes; the host side and the C ABI are in
// robchar_hip.hip, which reaches the launches below through three hidden entry points.  Device globals (the sin / cos table, the
// diagnostic tile counters) exist once per unit: rc_stats_* adds the two copies.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include "../../include/robchar_hip.h"
#include "kernel_params.h"
#include "tridiag_core.h"
#include "hermitian_core.h"

namespace {

using rckp::StaticH;
using rckp::FidParams;
using rckp::RingRepairList;
typedef __attribute__((address_space(1))) const void* rc_gptr_t;
typedef __attribute__((address_space(3))) void* rc_lptr_t;

#include "k_fidelity_chain.inc.h"

}  // namespace

extern "C" {

__attribute__((visibility("hidden"))) int rc_large_chain_launch(int N, void* stream, const rckp::FidParams* pp) {
    const FidParams& p = *pp;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)p.ntiles);
    switch (N) {
#define RC_CASE_ADJ(n) \
    case n: hipLaunchKernelGGL((mc_fid_chain_kernel<n, rc::kWeightsAdjugate>), grid, dim3(64), 0, s, p); break;
#ifndef RC_DEV_FEW_N
        RC_CASE_ADJ(17) RC_CASE_ADJ(18) RC_CASE_ADJ(19) RC_CASE_ADJ(20) RC_CASE_ADJ(21) RC_CASE_ADJ(22) RC_CASE_ADJ(23) RC_CASE_ADJ(24)
#endif
#undef RC_CASE_ADJ
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

__attribute__((visibility("hidden"))) int rc_large_ring_launch(int N, int mixed, void* stream, const rckp::FidParams* pp, double corner,
                                                               const rckp::RingRepairList* rlp, unsigned grid, unsigned rgrid) {
    const FidParams& p = *pp;
    hipStream_t s = (hipStream_t)stream;
    switch (N) {
#define RC_RING_CASE(n)                                                                                         \
    case n:                                                                                                     \
        if (mixed) {                                                                                            \
            hipLaunchKernelGGL(mc_fid_ring_mixed_kernel<n>, dim3(grid), dim3(64), 0, s, p, corner, *rlp);       \
            hipLaunchKernelGGL(mc_fid_ring_repair_kernel<n>, dim3(rgrid), dim3(64), 0, s, p, corner, *rlp);     \
        } else {                                                                                                \
            hipLaunchKernelGGL(mc_fid_ring_kernel<n>, dim3(grid), dim3(64), 0, s, p, corner);                   \
        }                                                                                                       \
        break;
#ifndef RC_DEV_FEW_N
        RC_RING_CASE(11) RC_RING_CASE(12) RC_RING_CASE(13) RC_RING_CASE(14) RC_RING_CASE(15) RC_RING_CASE(16)
#endif
#undef RC_RING_CASE
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

// device address of this unit's copy of a diagnostic counter: 0 = g_general_tiles, 1 = g_polish_tiles
__attribute__((visibility("hidden"))) int rc_large_counter_addr(int which, void** addr) {
    return which == 0 ? (int)hipGetSymbolAddress(addr, HIP_SYMBOL(g_general_tiles))
                      : (int)hipGetSymbolAddress(addr, HIP_SYMBOL(g_polish_tiles));
}

}  // extern "C"
