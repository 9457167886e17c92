// Parameter blocks of the fidelity kernels, shared by the two translation units of librobchar_hip.so (robchar_hip.hip: host side,
// C ABI and most kernels; robchar_large.hip: the largest instantiations - chains of 17 .. 24 spins, rings of 11 .. 16 - compiled in
// parallel).  Passed by value in the kernarg segment.
#pragma once
#include "../../include/robchar_hip.h"

namespace rckp {

struct StaticH {          // passed by value in the kernarg segment: no device allocation for 2N doubles
    double diag[RC_MAX_NSPIN];
    double off[RC_MAX_NSPIN];
};

struct FidParams {
    const double* ctrl;    // [C][N+1]
    const double* draws;   // [C][K][N][3]
    double* fid;           // [C][K]
    long long C, K;
    long long draw_cstride;     // elements between consecutive controllers' draw blocks (K*3N; 0 = shared set)
    long long tiles_per_ctrl;   // ceil(K / 64)
    long long ntiles;           // C * tiles_per_ctrl
    int in, out;
    int align16;                // draws base and every controller's run of K*3N doubles are 16-byte aligned
    StaticH h0;
    long long* stamps;          // diagnostic builds only (-DRC_STAMPS): [ntiles][8] s_memtime stamps
};

// repair list of the ring-topology route (mc_fid_ring_mixed_kernel -> mc_fid_ring_repair_kernel)
struct RingRepairList {
    unsigned long long* count;        // [1] number of listed samples of THIS call (zero on entry)
    unsigned long long* clear;        // [1] the counter the NEXT call on this stream will use: zeroed by this call's first wave
    long long* samples;               // [>= C * K] flat sample indices c * K + k
};

}  // namespace rckp
