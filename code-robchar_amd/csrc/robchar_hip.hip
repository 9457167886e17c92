// librobchar_hip.so - HIP kernels (gfx950 / MI355X) and the C ABI declared in include/robchar_hip.h.
//
// One translation unit: the kernels live in k_fidelity_chain.inc.h, k_fidelity_dense.inc.h, k_reduce_sort.inc.h and
// k_draws.inc.h (included below, inside this file's anonymous namespace); this file holds the shared parameter structs,
// the host side (per-device state, launch logic, the multi-device and legacy-stream drivers) and the extern "C" entries.
//
// Kernels
//   mc_fid_chain_kernel<N,M> one (controller, perturbation) sample per LANE, one wave per workgroup.  A wave
//                            owns tiles of 64 consecutive samples of ONE controller: the controller row is
//                            wave-uniform (scalar loads); a tile's 64*3N draws are one contiguous HBM run that
//                            is copied to LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, every HBM
//                            byte fetched exactly once, fully coalesced) and read back transposed (lane l
//                            takes its own 3N values).  Per lane: real symmetric tridiagonal implicit QL in
//                            registers with wave-uniform control flow (tridiag_core.h).
//   mc_fid_ring_kernel<N>    ring topology, N = 3..10: lane per sample, complex Hermitian matrix in registers ->
//                            Householder tridiagonalisation (hermitian_core.h) -> the same QL with two complex rows.
//   mc_fid_chain_anyn_kernel chains of 16 < N <= 32 spins: the general per-sample routine, work vectors in dynamic LDS.
//   mc_fid_jacobi_kernel     general complex Hermitian path (ring topology, cross-check): 8 or 4 samples per WAVE,
//                            dense matrix in LDS, round-robin cyclic Jacobi with the rotations of a round
//                            spread over the 64 lanes.
//   mc_fid_expm_kernel       dense complex, possibly non-Hermitian H (directional_perturbation): one WAVE per sample,
//                            Pade scaling-and-squaring expm in LDS - the reference's own algorithm shape on the device.
//   reduce_kernel            one workgroup per controller: RIM_1, std, min, Q(thr) for the centre / DKW-upper /
//                            DKW-lower variants in two passes over the K fidelities (fixed summation order).
//   sort_*_kernel            row sort for the ECDF: merge sort in LDS (K <= 16384), chunked bitonic network above.
//   philox_normal_kernel     counter-based Gaussian draws for sample spaces too large to draw on the host.
//
// Roofline: algorithmic HBM traffic is 24 N + 8 bytes per sample; the kernel is bound by
// fp64 VALU issue under the socket power cap, not by HBM.  See DESIGN.md.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <string>
#include <vector>

#include "../../include/robchar_hip.h"
#include "kernel_params.h"
#include "tridiag_core.h"
#include "hermitian_core.h"
#include "csym_core.h"
#include "sort_core.h"
#include "legacy_rng_core.h"
#include "mt19937_jump_poly.h"

// The largest instantiations live in a second translation unit (robchar_large.hip) that compiles in parallel with this one:
// chains of 17 .. 24 spins (general adjugate mode), rings of 11 .. 16 spins (mixed route + repair, all-fp64 route).  Each
// returns the hipError_t of its launch; the unit keeps its own copies of the diagnostic tile counters (rc_large_counter_addr).
extern "C" {
__attribute__((visibility("hidden"))) int rc_large_chain_launch(int N, void* stream, const rckp::FidParams* p);
__attribute__((visibility("hidden"))) int rc_large_ring_launch(int N, int mixed, void* stream, const rckp::FidParams* p, double corner,
                                                               const rckp::RingRepairList* rl, unsigned grid, unsigned rgrid);
__attribute__((visibility("hidden"))) int rc_large_counter_addr(int which, void** addr);
}

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define RC_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return fail(RC_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

using rckp::StaticH;
using rckp::FidParams;
using rckp::RingRepairList;

typedef __attribute__((address_space(1))) const void* rc_gptr_t;
typedef __attribute__((address_space(3))) void* rc_lptr_t;

#include "k_fidelity_chain.inc.h"
#include "k_fidelity_dense.inc.h"
#include "k_reduce_sort.inc.h"
#include "k_draws.inc.h"
#include "k_fidelity_philox.inc.h"
#include "k_directional.inc.h"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
std::mutex g_cfg_mu;                 // guards g_default_kernel only
int g_default_kernel = RC_KERNEL_AUTO;
long long* g_stamps = nullptr;      // diagnostic builds only

// Per-device state.  The C ABI takes a `device` argument everywhere, so nothing here is process-wide: the blocking
// entry points serialise PER DEVICE (two devices run concurrently, e.g. from the threads of
// rc_mc_metrics_sharded_f64), and the kernel attributes that must be raised before a launch
// (hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property) are tracked per device.
constexpr int kMaxDevices = 64;
enum { kAttrExpm = 0, kAttrAnyN, kAttrSortMerge, kAttrSortChunk, kAttrMtJump, kAttrMtJumpWide, kAttrCount };
// repair list of the ring-topology route (k_fidelity_chain.inc.h: RingRepairList), one per (device, stream): two counters
// used in turns + C*K sample slots.  Persistent - no allocation, no memset per call: every call's first wave zeroes the
// counter of the NEXT call - and per STREAM, so that launches on different streams of one device never share a list.
// Everything that touches the buffer is ordered on ITS stream (round 4): it is allocated with hipMallocAsync, zeroed with
// hipMemsetAsync and - when a larger problem arrives, or in rc_release_stream - released with hipFreeAsync on that stream,
// behind the kernels that still read it.  An `_async` entry therefore never synchronises the device (round 3: hipFree +
// hipMalloc + a null-stream hipMemset on first use).  rc_reserve_ring pre-sizes it outside a latency-critical region.
struct RingBuf {
    char* mem = nullptr;
    long long cap = 0;               // sample slots
    int turn = 0;                    // which of the two counters the next call uses
    unsigned long long last_use = 0; // ring_buf_for's clock: the least recently used entry is evicted beyond kMaxRingBufs
};
// Streams that ran a ring launch and were never handed back with rc_release_stream keep their buffer; the map is capped so that
// a caller cycling through short-lived streams cannot grow it without bound: the entry used longest ago is freed with a plain
// (device-synchronising, stream-agnostic: its stream may be gone) hipFree when a 17th stream arrives.  Leak bound per device:
// kMaxRingBufs x (256 + 10 bytes per sample of the largest launch on that stream).
constexpr size_t kMaxRingBufs = 16;
struct DeviceCtx {
    std::mutex mu;                   // blocking entry points: one at a time per device (they share `stream` and `ws`)
    hipStream_t stream = nullptr;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    std::mutex ring_mu;
    std::unordered_map<hipStream_t, RingBuf> ring_bufs;
    unsigned long long ring_clock = 0;
    std::atomic<bool> attr[kAttrCount];
    DeviceCtx() {
        for (auto& a : attr) a.store(false);
    }
};
DeviceCtx g_ctx[kMaxDevices];

int device_in_range(int device) {
    int n = 0;
    RC_HIP_CHECK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n || device >= kMaxDevices) return fail(RC_EINVAL, "device index out of range");
    return RC_OK;
}

// caller holds g_ctx[device].mu
int get_ctx(int device, DeviceCtx** out) {
    if (int rc = device_in_range(device)) return rc;
    RC_HIP_CHECK(hipSetDevice(device));
    if (!g_ctx[device].stream) RC_HIP_CHECK(hipStreamCreateWithFlags(&g_ctx[device].stream, hipStreamNonBlocking));
    *out = &g_ctx[device];
    return RC_OK;
}

// Raises a kernel's dynamic-LDS limit once per DEVICE (the current one).  Racing callers may both set it: idempotent.
int ensure_func_attr(int which, const void* func, int bytes) {
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(RC_EINVAL, "device index out of range");
    if (g_ctx[dev].attr[which].load(std::memory_order_acquire)) return RC_OK;
    RC_HIP_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_ctx[dev].attr[which].store(true, std::memory_order_release);
    return RC_OK;
}

int ensure_ws(DeviceCtx* ctx, size_t bytes) {
    if (ctx->ws_bytes >= bytes) return RC_OK;
    if (ctx->ws) RC_HIP_CHECK(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    RC_HIP_CHECK(hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return RC_OK;
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // plain host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice;
}

// stream-ordered release of a hipMallocAsync block at scope exit
struct StreamFree {
    void* p;
    hipStream_t s;
    ~StreamFree() {
        if (p) (void)hipFreeAsync(p, s);
    }
};

int check_common(int N, int in, int out, long long C, long long K) {
    if (N < 2 || N > RC_MAX_NSPIN) return fail(RC_EINVAL, "N must be in [2, 32]");
    if (in < 0 || in >= N || out < 0 || out >= N) return fail(RC_EINVAL, "in/out spin index out of range");
    if (C < 0 || K < 0) return fail(RC_EINVAL, "C and K must be non-negative");
    return RC_OK;
}

// caller holds the device's ring_mu.  Grow-only; stream-ordered (see RingBuf).
int ring_buf_reserve(RingBuf& rb, hipStream_t s, long long samples) {
    if (rb.cap >= samples && rb.mem) return RC_OK;
    if (rb.mem) RC_HIP_CHECK(hipFreeAsync(rb.mem, s));        // behind whatever on `s` still reads the old list
    rb.mem = nullptr;
    rb.cap = 0;
    const long long cap = samples + (samples >> 2);
    RC_HIP_CHECK(hipMallocAsync((void**)&rb.mem, 256 + (size_t)cap * sizeof(long long), s));
    RC_HIP_CHECK(hipMemsetAsync(rb.mem, 0, 256, s));           // both counters: ordered before the kernels that add to them
    rb.cap = cap;
    rb.turn = 0;
    return RC_OK;
}

// caller holds the device's ring_mu.  The stream's entry, created on first use; evicts the least recently used one beyond the cap.
RingBuf& ring_buf_for(DeviceCtx& ctx, hipStream_t s) {
    auto it = ctx.ring_bufs.find(s);
    if (it == ctx.ring_bufs.end()) {
        if (ctx.ring_bufs.size() >= kMaxRingBufs) {
            auto old = ctx.ring_bufs.begin();
            for (auto j = ctx.ring_bufs.begin(); j != ctx.ring_bufs.end(); ++j)
                if (j->second.last_use < old->second.last_use) old = j;
            if (old->second.mem) (void)hipFree(old->second.mem);
            ctx.ring_bufs.erase(old);
        }
        it = ctx.ring_bufs.emplace(s, RingBuf{}).first;
    }
    it->second.last_use = ++ctx.ring_clock;
    return it->second;
}

template <int N, int MODE>
int launch_chain(hipStream_t s, const FidParams& p) {
    const long long blocks = p.ntiles;
    if (blocks > 0x7fffffffLL) return fail(RC_EINVAL, "too many tiles for one launch");
    hipLaunchKernelGGL((mc_fid_chain_kernel<N, MODE>), dim3((unsigned)blocks), dim3(64), 0, s, p);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int enqueue_expm(hipStream_t s, int N, int in, int out, const double* h0_diag, const double* h0_offdiag, int ring,
                 const double* ctrl, const double* draws, long long draw_cstride, const double* diag_imag,
                 long long imag_cstride, long long C, long long K, double* fid, bool fast_complex_diagonal = false) {
    if (N > RC_MAX_NSPIN_FAST) return fail(RC_EINVAL, "the dense kernels (ring topology, expm) support N <= 16");
    ExpmParams p{};
    p.ctrl = ctrl;
    p.draws = draws;
    p.diag_imag = diag_imag;
    p.fid = fid;
    p.C = C;
    p.K = K;
    p.draw_cstride = draw_cstride;
    p.imag_cstride = imag_cstride;
    p.N = N;
    p.in = in;
    p.out = out;
    p.ring = ring ? 1 : 0;
    for (int i = 0; i < RC_MAX_NSPIN; ++i) {
        p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
        p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
    }
    const size_t lds = (size_t)kExpmWaves * kExpmBufs * N * N * sizeof(cplx);
    if (int rc = ensure_func_attr(kAttrExpm, (const void*)mc_fid_expm_kernel,
                                  kExpmWaves * kExpmBufs * RC_MAX_NSPIN_FAST * RC_MAX_NSPIN_FAST * (int)sizeof(cplx)))
        return rc;
    const long long total = C * K;
    long long blocks = (total + kExpmWaves - 1) / kExpmWaves;
    if (blocks > 256LL * 16) blocks = 256LL * 16;              // grid-stride loop inside; every wave exits
    if (fast_complex_diagonal && !ring && N <= kCsymMaxN) {
        // chain with a complex diagonal: the lane-per-sample complex symmetric QL route, then the expm kernel over the
        // samples it marked (normally none: the waves of that pass find nothing and end)
        const long long cblocks = (total + 63) / 64;
        if (cblocks > 0x7fffffffLL) return fail(RC_EINVAL, "too many samples for one launch");
        switch (N) {
#define RC_CSYM_CASE(n) \
    case n: hipLaunchKernelGGL(mc_fid_csym_kernel<n>, dim3((unsigned)cblocks), dim3(64), 0, s, p); break;
            RC_CSYM_CASE(2) RC_CSYM_CASE(3) RC_CSYM_CASE(4) RC_CSYM_CASE(5) RC_CSYM_CASE(6) RC_CSYM_CASE(7) RC_CSYM_CASE(8)
            RC_CSYM_CASE(9) RC_CSYM_CASE(10) RC_CSYM_CASE(11) RC_CSYM_CASE(12)
#undef RC_CSYM_CASE
        }
        p.only_marked = 1;
    }
    hipLaunchKernelGGL(mc_fid_expm_kernel, dim3((unsigned)blocks), dim3(64 * kExpmWaves), lds, s, p);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int enqueue_fidelity(hipStream_t s, int kernel, int N, int in, int out, const double* h0_diag,
                     const double* h0_offdiag, int ring, const double* ctrl, const double* draws,
                     long long draw_cstride, long long C, long long K, double* fid) {
    if (draw_cstride < 0) draw_cstride = K * N * 3;          // default: private draws per controller
    if (draw_cstride != 0 && draw_cstride < K * N * 3) return fail(RC_EINVAL, "draws_ctrl_stride overlaps controllers");
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (C == 0 || K == 0) return RC_OK;
    if (!ctrl || !draws || !fid) return fail(RC_EINVAL, "NULL array pointer");
    const bool ends = (in == 0 && out == N - 1) || (in == N - 1 && out == 0);
    if (ring && N == 2) ring = 0;               // the closure of a 2-ring IS the chain bond (noise_model.py:83-85 re-assigns 1)
    // ring: AUTO = the mixed-precision route (round 3; N = 11 .. 16 since round 5: folded band reduction instead of the dense
    // Householder); RC_KERNEL_RING_HH asks for the all-fp64 route explicitly
    const bool mixed_ring = (kernel == RC_KERNEL_AUTO) && ring && N <= kRingMaxN;
    if (kernel == RC_KERNEL_AUTO)
        kernel = ring ? (N <= kRingMaxN ? RC_KERNEL_RING_HH : RC_KERNEL_JACOBI) : RC_KERNEL_TRIDIAG_ADJ;
    if (kernel == RC_KERNEL_RING_HH) {
        if (!ring) return fail(RC_EINVAL, "RC_KERNEL_RING_HH is the ring-topology kernel (chains: the tridiagonal kernels)");
        if (N > kRingMaxN) return fail(RC_EINVAL, "the lane-per-sample ring kernels support N <= 16");
        FidParams p{};
        p.ctrl = ctrl;
        p.draws = draws;
        p.fid = fid;
        p.C = C;
        p.K = K;
        p.draw_cstride = draw_cstride;
        p.tiles_per_ctrl = (K + 63) / 64;
        p.ntiles = C * p.tiles_per_ctrl;
        p.in = in;
        p.out = out;
        p.align16 = (((uintptr_t)draws & 15) == 0 && (((size_t)draw_cstride * 8) & 15) == 0) ? 1 : 0;
        for (int i = 0; i < RC_MAX_NSPIN; ++i) {
            p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
            p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
        }
        if (p.ntiles > 0x7fffffffLL) return fail(RC_EINVAL, "too many tiles for one launch");
        const dim3 grid((unsigned)p.ntiles);
        if (mixed_ring) {
            // mixed-precision route + the repair launch right behind it on the same stream: the route lists the samples it
            // does not trust itself with, the repair kernel recomputes them lane per sample (a few hundred waves that end
            // at once when the list is empty).  List and counters: the stream's persistent RingBuf.
            RingRepairList rl{};
            {
                int dev = 0;
                RC_HIP_CHECK(hipGetDevice(&dev));
                if (dev < 0 || dev >= kMaxDevices) return fail(RC_EINVAL, "device index out of range");
                std::lock_guard<std::mutex> lk(g_ctx[dev].ring_mu);
                RingBuf& rb = ring_buf_for(g_ctx[dev], s);
                if (int rc = ring_buf_reserve(rb, s, C * K)) return rc;      // (first use of this stream, or a larger problem: rare)
                rl.count = (unsigned long long*)(rb.mem + 64 * rb.turn);
                rl.clear = (unsigned long long*)(rb.mem + 64 * (rb.turn ^ 1));
                rl.samples = (long long*)(rb.mem + 256);
                rb.turn ^= 1;
            }
            // the repair kernel walks the list with a grid-stride loop; its width only matters when MANY samples are listed
            // (a translation-invariant ring lists every sample).  RC_RING_REPAIR_GRID (environment, read once): A/B knob.
            static const long long kRepairGrid = [] {
                const char* e = getenv("RC_RING_REPAIR_GRID");
                const long long v = e ? atoll(e) : 0;
                return v > 0 ? v : 1024LL;
            }();
            const long long nwaves = (C * K + 63) / 64;
            const dim3 rgrid((unsigned)(nwaves < kRepairGrid ? nwaves : kRepairGrid));
            if (N > 10) {                              // 11 .. 16: instantiated in robchar_large.hip
                RC_HIP_CHECK((hipError_t)rc_large_ring_launch(N, 1, (void*)s, &p, 1.0, &rl, grid.x, rgrid.x));
                return RC_OK;
            }
            switch (N) {
#define RC_RING_CASE(n)                                                                         \
    case n:                                                                                     \
        hipLaunchKernelGGL(mc_fid_ring_mixed_kernel<n>, grid, dim3(64), 0, s, p, 1.0, rl);      \
        hipLaunchKernelGGL(mc_fid_ring_repair_kernel<n>, rgrid, dim3(64), 0, s, p, 1.0, rl);    \
        break;
                RC_RING_CASE(3) RC_RING_CASE(4) RC_RING_CASE(5) RC_RING_CASE(6) RC_RING_CASE(7) RC_RING_CASE(8) RC_RING_CASE(9)
                RC_RING_CASE(10)
#undef RC_RING_CASE
            }
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
        if (N > 10) {                                   // 11 .. 16: instantiated in robchar_large.hip
            RC_HIP_CHECK((hipError_t)rc_large_ring_launch(N, 0, (void*)s, &p, 1.0, nullptr, grid.x, 0u));
            return RC_OK;
        }
        switch (N) {
#define RC_RING_CASE(n) \
    case n: hipLaunchKernelGGL(mc_fid_ring_kernel<n>, grid, dim3(64), 0, s, p, 1.0); break;
            RC_RING_CASE(3) RC_RING_CASE(4) RC_RING_CASE(5) RC_RING_CASE(6) RC_RING_CASE(7) RC_RING_CASE(8) RC_RING_CASE(9)
            RC_RING_CASE(10)
#undef RC_RING_CASE
        }
        RC_HIP_CHECK(hipGetLastError());
        return RC_OK;
    }
    if (kernel == RC_KERNEL_TRIDIAG_QL || kernel == RC_KERNEL_TRIDIAG_ADJ) {
        // (N = 15, 16: the end-to-end instantiation needs more than 256 registers - one wave per SIMD - and is slower than the
        // general adjugate one, which covers the pair (0, N - 1) as well and fits two: 338 / 395 us against 310 / 342 us per 1e6)
        const int mode = (kernel == RC_KERNEL_TRIDIAG_QL) ? rc::kWeightsRows
                                                          : ((ends && N <= 14) ? rc::kWeightsEnds : rc::kWeightsAdjugate);
        if (ring) return fail(RC_EINVAL, "the tridiagonal QL kernel handles chain topology only");
        FidParams p{};
        p.ctrl = ctrl;
        p.draws = draws;
        p.fid = fid;
        p.C = C;
        p.K = K;
        p.draw_cstride = draw_cstride;
        p.tiles_per_ctrl = (K + 63) / 64;
        p.ntiles = C * p.tiles_per_ctrl;
        p.in = in;
        p.out = out;
        p.align16 = (((uintptr_t)draws & 15) == 0 && (((size_t)draw_cstride * 8) & 15) == 0) ? 1 : 0;
        p.stamps = g_stamps;
        for (int i = 0; i < RC_MAX_NSPIN; ++i) {
            p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
            p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
        }
        // chains of 17 .. 24 spins (round 5): the register-resident general-adjugate instantiation (one wave per SIMD) for the
        // eigenvalue-only kernels; the rows mode keeps the LDS kernel there (its 4N more doubles of state do not fit)
        if (N > RC_MAX_NSPIN_CHAIN || (N > RC_MAX_NSPIN_FAST && mode == rc::kWeightsRows)) {
            const long long blocks = p.ntiles;
            if (blocks > 0x7fffffffLL) return fail(RC_EINVAL, "too many tiles for one launch");
            const size_t lds = (size_t)4 * N * 64 * sizeof(double);
            if (int rc = ensure_func_attr(kAttrAnyN, (const void*)mc_fid_chain_anyn_kernel, 4 * RC_MAX_NSPIN * 64 * (int)sizeof(double)))
                return rc;
            hipLaunchKernelGGL(mc_fid_chain_anyn_kernel, dim3((unsigned)blocks), dim3(64), lds, s, p, N);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
        if (N > RC_MAX_NSPIN_FAST) {                   // 17 .. 24: instantiated in robchar_large.hip
            if (p.ntiles > 0x7fffffffLL) return fail(RC_EINVAL, "too many tiles for one launch");
            RC_HIP_CHECK((hipError_t)rc_large_chain_launch(N, (void*)s, &p));
            return RC_OK;
        }
        switch (N) {
#define RC_CASE(n)                                                                        \
    case n:                                                                               \
        return mode == rc::kWeightsRows ? launch_chain<n, rc::kWeightsRows>(s, p)         \
               : (mode == rc::kWeightsEnds ? launch_chain<n, rc::kWeightsEnds>(s, p)      \
                                           : launch_chain<n, rc::kWeightsAdjugate>(s, p));
/* N = 15, 16: no end-to-end instantiation (never dispatched: see above) */
#define RC_CASE_ADJ_ROWS(n)                                                               \
    case n:                                                                               \
        return mode == rc::kWeightsRows ? launch_chain<n, rc::kWeightsRows>(s, p) : launch_chain<n, rc::kWeightsAdjugate>(s, p);
#ifdef RC_DEV_FEW_N      /* kernel-tuning builds only (scripts/): the three BASELINE sizes, a third of the compile time */
            RC_CASE(5) RC_CASE(7) RC_CASE(10)
#else
            RC_CASE(2) RC_CASE(3) RC_CASE(4) RC_CASE(5) RC_CASE(6) RC_CASE(7) RC_CASE(8) RC_CASE(9)
            RC_CASE(10) RC_CASE(11) RC_CASE(12) RC_CASE(13) RC_CASE(14) RC_CASE_ADJ_ROWS(15) RC_CASE_ADJ_ROWS(16)
#endif
#undef RC_CASE
#undef RC_CASE_ADJ_ROWS
        }
        return fail(RC_EINVAL, "unsupported N");
    }
    if (N > RC_MAX_NSPIN_FAST) return fail(RC_EINVAL, "the dense kernels (ring topology, expm) support N <= 16");
    if (kernel == RC_KERNEL_EXPM)
        return enqueue_expm(s, N, in, out, h0_diag, h0_offdiag, ring, ctrl, draws, draw_cstride, nullptr, 0, C, K, fid);
    if (kernel == RC_KERNEL_JACOBI) {
        JacParams p{};
        p.ctrl = ctrl;
        p.draws = draws;
        p.fid = fid;
        p.C = C;
        p.K = K;
        p.draw_cstride = draw_cstride;
        p.N = N;
        p.in = in;
        p.out = out;
        p.ring = ring ? 1 : 0;
        for (int i = 0; i < RC_MAX_NSPIN; ++i) {
            p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
            p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
        }
        const long long total = C * K;
        const int spw = (N <= 8) ? 8 : 4;                       // samples per wave
        long long blocks = (total + kJacWaves * spw - 1) / (kJacWaves * spw);
        if (blocks > 256LL * 16) blocks = 256LL * 16;          // grid-stride loop inside; every wave exits
        if (N <= 8)
            hipLaunchKernelGGL((mc_fid_jacobi_kernel<8, 8>), dim3((unsigned)blocks), dim3(64 * kJacWaves), 0, s, p);
        else
            hipLaunchKernelGGL((mc_fid_jacobi_kernel<16, 16>), dim3((unsigned)blocks), dim3(64 * kJacWaves), 0, s, p);
        RC_HIP_CHECK(hipGetLastError());
        return RC_OK;
    }
    return fail(RC_EINVAL, "unknown kernel id");
}

int enqueue_reduce(hipStream_t s, const double* fid, long long C, long long K, const double* thr, int nq,
                   double eps, double* rim1, double* stdv, double* minf, double* q, double* sorted_out, bool standalone = false) {
    if (C < 0 || K < 0) return fail(RC_EINVAL, "C and K must be non-negative");
    if (nq < 0 || nq > kMaxQ) return fail(RC_EINVAL, "nq must be in [0, 8]");
    if (nq > 0 && !thr) return fail(RC_EINVAL, "q_thresholds is NULL");
    if (C == 0) return RC_OK;
    if (K == 0) return fail(RC_EINVAL, "K must be positive for a reduction");
    if (!fid) return fail(RC_EINVAL, "NULL fid pointer");
    if (C > 0x7fffffffLL) return fail(RC_EINVAL, "too many controllers for one launch");
    RedParams p{};
    p.fid = fid;
    p.C = C;
    p.K = K;
    p.nq = nq;
    for (int j = 0; j < nq; ++j) p.thr[j] = thr[j];
    p.eps = eps;
    p.rim1 = rim1;
    p.stdv = stdv;
    p.minf = minf;
    p.q = q;
    if (rim1 || stdv || minf || q) {
        if (K <= kWaveRowMaxK && C >= 64) {                   // many short rows: one wave per row
            const dim3 grid((unsigned)((C + 3) / 4));
            if (nq == 0)
                hipLaunchKernelGGL(reduce_rows_wave_kernel<0>, grid, dim3(256), 0, s, p);
            else if (nq <= 2)
                hipLaunchKernelGGL(reduce_rows_wave_kernel<2>, grid, dim3(256), 0, s, p);
            else
                hipLaunchKernelGGL(reduce_rows_wave_kernel<kMaxQ>, grid, dim3(256), 0, s, p);
        } else if (nq <= 2 && K > kWaveRowMaxK && K <= 32LL * 128) {      // rows of up to 4096 / 8192 values: narrower workgroups,
            if (nq == 0) hipLaunchKernelGGL((reduce_kernel<0, 128>), dim3((unsigned)C), dim3(128), 0, s, p);      // more rows in flight
            else hipLaunchKernelGGL((reduce_kernel<2, 128>), dim3((unsigned)C), dim3(128), 0, s, p);
        } else if (nq <= 2 && K > kWaveRowMaxK && K <= 32LL * 256) {
            if (nq == 0) hipLaunchKernelGGL((reduce_kernel<0, 256>), dim3((unsigned)C), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((reduce_kernel<2, 256>), dim3((unsigned)C), dim3(256), 0, s, p);
        } else if (standalone && nq <= 2 && K > 32LL * 256 && K <= 40LL * 256) {
            // Rows of 8193 .. 10 240 values (BASELINE's K = 10 000) with NOTHING running beside the reduction (RC_REDUCE_STANDALONE:
            // the product's cold calls, the blocking and multi-device entries): 256 threads x 40 cached values - five rows in flight
            // per CU instead of two: 11 000 rows 440 -> 208 us, 1 000 rows 46 -> 29.5 us (profiles/r04_reduce_sweep.txt).  A caller
            // that OVERLAPS the reduction with fidelity launches (bench.py's side stream in steady state) keeps the 512-thread
            // version below: latency-bound, it fills issue slots the fidelity kernel leaves idle, where this one displaces its
            // waves (steady-state step +1.5 %, round 4).  The hint picks the route, and with it the summation order of these rows.
            if (nq == 0) hipLaunchKernelGGL((reduce_kernel<0, 256, 40>), dim3((unsigned)C), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((reduce_kernel<2, 256, 40>), dim3((unsigned)C), dim3(256), 0, s, p);
        } else if (nq == 0)
            hipLaunchKernelGGL(reduce_kernel<0>, dim3((unsigned)C), dim3(kRedThreads), 0, s, p);
        else if (nq <= 2)
            hipLaunchKernelGGL(reduce_kernel<2>, dim3((unsigned)C), dim3(kRedThreads), 0, s, p);
        else
            hipLaunchKernelGGL(reduce_kernel<kMaxQ>, dim3((unsigned)C), dim3(kRedThreads), 0, s, p);
        RC_HIP_CHECK(hipGetLastError());
    }
    if (sorted_out) {
        long long P = 2;
        while (P < K) P <<= 1;
        if (C * P > (1LL << 34)) return fail(RC_EINVAL, "sorted_out: workspace would exceed 128 GiB");
        if (K <= kSortChunk) {                                        // one fused launch, no workspace: merge sort
            if (int rc = ensure_func_attr(kAttrSortMerge, (const void*)sort_rows_merge_kernel, (kSortChunk / 16 * 17 + 1) * 8))
                return rc;
            const int T = (int)((K + 15) / 16), n = 16 * T;
            const int threads = ((T + 63) / 64) * 64;
            hipLaunchKernelGGL(sort_rows_merge_kernel, dim3((unsigned)C), dim3(threads), (size_t)(n / 16 * 17 + 1) * sizeof(double),
                               s, fid, sorted_out, K, n);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
        // workspace of THIS call, allocated and released in stream order (two streams sorting long rows at the same time
        // each get their own; nothing is shared between calls)
        struct SortWs {
            double* work = nullptr;
            int* flags = nullptr;
        } wsv, *ws = &wsv;
        RC_HIP_CHECK(hipMallocAsync((void**)&wsv.work, (size_t)C * P * sizeof(double), s));
        if (hipError_t e = hipMallocAsync((void**)&wsv.flags, (size_t)C * sizeof(int), s); e != hipSuccess) {
            (void)hipFreeAsync(wsv.work, s);
            return fail(RC_EHIP, std::string("hipMallocAsync(sort flags): ") + hipGetErrorString(e));
        }
        struct Release {
            SortWs* w;
            hipStream_t st;
            ~Release() {
                (void)hipFreeAsync(w->work, st);
                (void)hipFreeAsync(w->flags, st);
            }
        } release{ws, s};
        RC_HIP_CHECK(hipMemsetAsync(ws->flags, 0, (size_t)C * sizeof(int), s));
        {
            // long rows (K > 16384): chunk sort -> for each larger size one fused pass over the strides >= 16384 (up to four per
            // launch) and one chunk pass over the rest; the first pass reads `fid`, the last writes `sorted_out`
            if (int rc = ensure_func_attr(kAttrSortChunk, (const void*)sort_chunk16_kernel, kSortChunk / 16 * 17 * 8)) return rc;
            const size_t lds16 = (size_t)(kSortChunk / 16 * 17) * sizeof(double);
            const dim3 cgrid((unsigned)C, (unsigned)(P / kSortChunk));
            hipLaunchKernelGGL(sort_chunk16_kernel, cgrid, dim3(kSortThreads), lds16, s, fid, ws->work, sorted_out, ws->flags,
                               K, P, 2LL, (long long)kSortChunk, 1, 0);
            for (long long size = 2LL * kSortChunk; size <= P; size <<= 1) {
                long long hi = size >> 1;
                while (hi >= kSortChunk) {
                    int nleft = 0;
                    for (long long x = hi; x >= kSortChunk; x >>= 1) ++nleft;
                    const int take = nleft >= 4 ? 4 : nleft;
                    const long long S = hi >> (take - 1);
                    const dim3 ggrid((unsigned)C, 128);
                    switch (take) {
                        case 4: hipLaunchKernelGGL(sort_global_fused_kernel<4>, ggrid, dim3(256), 0, s, ws->work, P, size, S); break;
                        case 3: hipLaunchKernelGGL(sort_global_fused_kernel<3>, ggrid, dim3(256), 0, s, ws->work, P, size, S); break;
                        case 2: hipLaunchKernelGGL(sort_global_fused_kernel<2>, ggrid, dim3(256), 0, s, ws->work, P, size, S); break;
                        default: hipLaunchKernelGGL(sort_global_fused_kernel<1>, ggrid, dim3(256), 0, s, ws->work, P, size, S); break;
                    }
                    hi = S >> 1;
                }
                hipLaunchKernelGGL(sort_chunk16_kernel, cgrid, dim3(kSortThreads), lds16, s, fid, ws->work, sorted_out,
                                   ws->flags, K, P, size, size, 0, size == P ? 1 : 0);
            }
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
        RC_HIP_CHECK(hipGetLastError());
    }
    return RC_OK;
}

// ------------------------------------------------------------------------------------------------
// single-process multi-device driver (rc_mc_fidelity_sharded_f64 / rc_mc_metrics_sharded_f64)
// ------------------------------------------------------------------------------------------------
// The (controller x perturbation) sample space is split by CONTROLLER into contiguous balanced blocks, one per device
// (per-controller vectors and reductions stay device-local; the reference's only parallel construct is
// the dead Pool at mcsim.py:451-455).  One host thread per device: its controllers are processed in chunks of at
// most kShardChunkBytes of draws through the device's workspace - H2D of the chunk's draws (or Philox generation on the
// device), fidelity kernel, reduction, D2H of fidelities / metric rows straight into the caller's host arrays.  No
// collective is needed for a host-resident result.
constexpr size_t kShardChunkBytes = (size_t)4 << 30;

struct ShardJob {
    int device, kernel, N, in, out, ring;
    const double *h0d, *h0o, *ctrl, *draws;          // host; draws may be null (Philox)
    unsigned long long seed, offset;
    double sigma;
    long long C, K, c0, c1;                          // this device owns controllers [c0, c1)
    const double* thr;
    int nq;
    double eps;
    double *rim1, *stdv, *minf, *q, *fid_out;        // host, FULL arrays ([3][C], [3][nq][C], [C][K]); may be null
    int rc = RC_OK;
    std::string err;
};

int run_shard_locked(ShardJob* j) {
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(j->device, &ctx)) return rc;
    const long long G = 3LL * j->N, K = j->K, Cl = j->c1 - j->c0;
    if (Cl <= 0 || K == 0) return RC_OK;
    const bool want_red = j->rim1 || j->stdv || j->minf || (j->q && j->nq);
    // counter-based draws on a chain of <= 16 spins with the eigenvalue-only kernels: generated inside the fidelity kernel
    // (k_fidelity_philox.inc.h) - no draw tensor, so a chunk is bounded by its fidelities only (K x 8 bytes per controller)
    // (ONE predicate for this route and the Python layer's: rc_philox_fused_pays, environment switch read per call)
    const bool fused = !j->draws && !j->ring && rc_philox_fused_pays(j->N, j->in, j->out) == 1 &&
                       (j->kernel == RC_KERNEL_AUTO || j->kernel == RC_KERNEL_TRIDIAG_ADJ);
    long long cc_max = (long long)(kShardChunkBytes / ((size_t)K * (fused ? 1 : G) * sizeof(double)));
    if (cc_max < 1) cc_max = 1;
    if (cc_max > Cl) cc_max = Cl;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nb_ctrl = up((size_t)cc_max * (j->N + 1) * sizeof(double));
    const size_t nb_draw = fused ? 0 : up((size_t)cc_max * K * G * sizeof(double));
    const size_t nb_fid = up((size_t)cc_max * K * sizeof(double));
    const size_t nb_c3 = up((size_t)3 * cc_max * sizeof(double));
    const size_t nb_q = up((size_t)3 * (j->nq > 0 ? j->nq : 1) * cc_max * sizeof(double));
    if (int rc = ensure_ws(ctx, nb_ctrl + nb_draw + nb_fid + 3 * nb_c3 + nb_q)) return rc;
    char* w = (char*)ctx->ws;
    double* d_ctrl = (double*)w; w += nb_ctrl;
    double* d_draw = (double*)w; w += nb_draw;
    double* d_fid = (double*)w;  w += nb_fid;
    double* d_rim = (double*)w;  w += nb_c3;
    double* d_std = (double*)w;  w += nb_c3;
    double* d_min = (double*)w;  w += nb_c3;
    double* d_q = (double*)w;
    hipStream_t st = ctx->stream;
    for (long long a = j->c0; a < j->c1; a += cc_max) {
        const long long cc = (j->c1 - a < cc_max) ? (j->c1 - a) : cc_max;
        RC_HIP_CHECK(hipMemcpyAsync(d_ctrl, j->ctrl + a * (j->N + 1), (size_t)cc * (j->N + 1) * sizeof(double),
                                    hipMemcpyHostToDevice, st));
        if (fused) {
            // element ((c K + k) N + i) 3 + slot of the stream: independent of how the controllers are sharded
            if (int rc = rc_mc_fidelity_philox_f64_async(j->device, st, j->kernel, j->N, j->in, j->out, j->h0d, j->h0o, d_ctrl,
                                                         j->seed, j->offset + (unsigned long long)(a * K * G), j->sigma, nullptr,
                                                         cc, K, d_fid))
                return rc;
        } else if (j->draws) {
            RC_HIP_CHECK(hipMemcpyAsync(d_draw, j->draws + a * K * G, (size_t)cc * K * G * sizeof(double),
                                        hipMemcpyHostToDevice, st));
        } else {
            // element ((c K + k) N + i) 3 + slot of the stream: independent of how the controllers are sharded
            if (int rc = rc_draws_philox_f64_async(j->device, st, j->seed, j->offset + (unsigned long long)(a * K * G),
                                                   cc * K * G, j->sigma, d_draw))
                return rc;
        }
        if (!fused)
            if (int rc = enqueue_fidelity(st, j->kernel, j->N, j->in, j->out, j->h0d, j->h0o, j->ring, d_ctrl, d_draw, -1, cc, K,
                                          d_fid))
                return rc;
        if (want_red) {
            if (int rc = enqueue_reduce(st, d_fid, cc, K, j->thr, j->nq, j->eps, j->rim1 ? d_rim : nullptr,
                                        j->stdv ? d_std : nullptr, j->minf ? d_min : nullptr,
                                        (j->q && j->nq) ? d_q : nullptr, nullptr, /*standalone=*/true))
                return rc;
            // device rows [3][cc] -> columns [a, a+cc) of the caller's [3][C]
            const size_t wbytes = (size_t)cc * sizeof(double), dpitch = (size_t)j->C * sizeof(double);
            if (j->rim1) RC_HIP_CHECK(hipMemcpy2DAsync(j->rim1 + a, dpitch, d_rim, wbytes, wbytes, 3, hipMemcpyDeviceToHost, st));
            if (j->stdv) RC_HIP_CHECK(hipMemcpy2DAsync(j->stdv + a, dpitch, d_std, wbytes, wbytes, 3, hipMemcpyDeviceToHost, st));
            if (j->minf) RC_HIP_CHECK(hipMemcpy2DAsync(j->minf + a, dpitch, d_min, wbytes, wbytes, 3, hipMemcpyDeviceToHost, st));
            if (j->q && j->nq)
                RC_HIP_CHECK(hipMemcpy2DAsync(j->q + a, dpitch, d_q, wbytes, wbytes, (size_t)3 * j->nq, hipMemcpyDeviceToHost, st));
        }
        if (j->fid_out)
            RC_HIP_CHECK(hipMemcpyAsync(j->fid_out + a * K, d_fid, (size_t)cc * K * sizeof(double), hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipStreamSynchronize(st));       // the workspace is reused by the next chunk
    }
    return RC_OK;
}

void run_shard(ShardJob* j) {
    std::lock_guard<std::mutex> lk(g_ctx[j->device].mu);
    j->rc = run_shard_locked(j);
    if (j->rc) j->err = g_last_error;            // thread-local of THIS worker: hand it to the caller
}

int run_sharded(int ndev, const int* devices, ShardJob proto) {
    if (int rc = check_common(proto.N, proto.in, proto.out, proto.C, proto.K)) return rc;
    if (ndev < 1 || ndev > kMaxDevices) return fail(RC_EINVAL, "ndev must be in [1, 64]");
    if (proto.nq < 0 || proto.nq > kMaxQ) return fail(RC_EINVAL, "nq must be in [0, 8]");
    if (proto.nq > 0 && !proto.thr) return fail(RC_EINVAL, "q_thresholds is NULL");
    if (proto.C == 0 || proto.K == 0) return RC_OK;
    if (!proto.ctrl) return fail(RC_EINVAL, "NULL controllers pointer");
    if (!proto.fid_out && !proto.rim1 && !proto.stdv && !proto.minf && !(proto.q && proto.nq))
        return fail(RC_EINVAL, "no output requested");
    for (int r = 0; r < ndev; ++r) {
        const int dev = devices ? devices[r] : r;
        if (int rc = device_in_range(dev)) return rc;
        // a device listed twice would only serialise on its lock; refused because it is never what a caller wants -
        // except for rehearsing the multi-block assembly on a one-GPU box (RC_ALLOW_DUPLICATE_DEVICES=1, tests only)
        const char* dup = getenv("RC_ALLOW_DUPLICATE_DEVICES");
        for (int r2 = 0; r2 < r && !(dup && dup[0] == '1'); ++r2)
            if ((devices ? devices[r2] : r2) == dev) return fail(RC_EINVAL, "a device is listed twice");
    }
    std::vector<ShardJob> jobs(ndev, proto);
    const long long base = proto.C / ndev, extra = proto.C % ndev;
    long long start = 0;
    for (int r = 0; r < ndev; ++r) {
        jobs[r].device = devices ? devices[r] : r;
        jobs[r].c0 = start;
        start += base + (r < extra ? 1 : 0);
        jobs[r].c1 = start;
    }
    std::vector<std::thread> th;
    for (int r = 1; r < ndev; ++r) th.emplace_back(run_shard, &jobs[r]);
    run_shard(&jobs[0]);
    for (auto& t : th) t.join();
    for (int r = 0; r < ndev; ++r)
        if (jobs[r].rc) return fail(jobs[r].rc, "device " + std::to_string(jobs[r].device) + ": " + jobs[r].err);
    return RC_OK;
}

// ------------------------------------------------------------------------------------------------
// driver of the device-side legacy stream (rc_draws_legacy_f64)
// ------------------------------------------------------------------------------------------------
// The stream is produced in SEGMENTS of at most kLegacySegWords raw words: block-aligned word range [g0, g1) of the
// generator's output sequence, the first block being known (the caller's key / the previous segment's last block).
// After every segment the accepted-attempt count comes back to the host, which decides whether more words are needed.
constexpr long long kLegacySegWords = 1LL << 27;      // 512 MiB of raw words per segment

// `words` (a multiple of 624, >= 1248) raw MT19937 state words into `raw` (device): raw[0 .. 624) = the block `carry`
// (host), the rest is the generator's output sequence behind it - P parallel sub-streams of kMtJumpWords words whose
// start windows come from jump-ahead (k_draws.inc.h).  Enqueued on `st`; `carry` must stay valid until the copy ran.
int mt_fill_raw(hipStream_t st, const unsigned int* carry, long long words, unsigned int* raw) {
    const int P = (int)((words - rcl::kMtN + kMtJumpWords - 1) / kMtJumpWords);
    unsigned int* d_seeds = nullptr;
    RC_HIP_CHECK(hipMallocAsync((void**)&d_seeds, (size_t)P * rcl::kMtN * sizeof(unsigned int), st));
    StreamFree free_seeds{d_seeds, st};
    RC_HIP_CHECK(hipMemcpyAsync(d_seeds, carry, rcl::kMtN * sizeof(unsigned int), hipMemcpyHostToDevice, st));
    if (P > 1) {
        if (int rc = ensure_func_attr(kAttrMtJump, (const void*)mt19937_jump_step_kernel<24>, kJumpLdsWords * (int)sizeof(unsigned int)))
            return rc;
        if (int rc = ensure_func_attr(kAttrMtJumpWide, (const void*)mt19937_jump_step_kernel<12>, kJumpLdsWords * (int)sizeof(unsigned int)))
            return rc;
        // windows 1..3 by jumps of B, then up to 4 at a time by 4 B, up to 16 at a time by 16 B, up to 64 at a time by 64 B
        int have = 1;
        for (int stride = 1; stride <= kJumpMaxStride && have < P; stride *= 4) {
            const int limit = (stride == kJumpMaxStride) ? P : (P < 4 * stride ? P : 4 * stride);
            while (have < limit) {
                const int cnt = (limit - have < stride) ? (limit - have) : stride;
                if (cnt <= 10)                            // few jumps: all workgroups resident at once either way - 24 per jump,
                    hipLaunchKernelGGL((mt19937_jump_step_kernel<24>), dim3(24, cnt), dim3(kJumpThreads),      // 26 words each
                                       kJumpLdsWords * sizeof(unsigned int), st, d_seeds, have, stride);
                else                                      // many: 12 per jump, 52 words each (the regeneration is per workgroup)
                    hipLaunchKernelGGL((mt19937_jump_step_kernel<12>), dim3(12, cnt), dim3(kJumpThreads),
                                       kJumpLdsWords * sizeof(unsigned int), st, d_seeds, have, stride);
                have += cnt;
            }
        }
    }
    hipLaunchKernelGGL(mt19937_raw_kernel, dim3((unsigned)P), dim3(64), 0, st, (const unsigned int*)d_seeds, raw, words);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int legacy_normal_stream(hipStream_t st, rc_mt19937_state* state, long long n_periods, long long period, long long skip,
                         const double* scales_host, double* out_dev) {
    const long long n_total = n_periods * period;
    const long long e_shift = state->has_gauss ? 1 : 0;
    const long long n_need = n_total - e_shift;
    const long long pairs = (n_need + 1) / 2;
    double* d_scales = nullptr;
    RC_HIP_CHECK(hipMallocAsync((void**)&d_scales, (size_t)n_periods * sizeof(double), st));
    StreamFree free_scales{d_scales, st};
    RC_HIP_CHECK(hipMemcpyAsync(d_scales, scales_host, (size_t)n_periods * sizeof(double), hipMemcpyHostToDevice, st));
    if (e_shift) {                                    // element 0 of the stream is the host's cached normal
        long long pi;
        const long long slot = rcl::stream_slot(0, period, skip, &pi);
        if (slot >= 0) {
            const double v = 0.0 + scales_host[pi] * state->gauss;
            RC_HIP_CHECK(hipMemcpyAsync(out_dev + slot, &v, sizeof(double), hipMemcpyHostToDevice, st));
            RC_HIP_CHECK(hipStreamSynchronize(st));   // `v` is a stack temporary
        }
        state->has_gauss = 0;
        state->gauss = 0.0;
    }
    if (pairs == 0) return RC_OK;

    long long* d_last = nullptr;
    RC_HIP_CHECK(hipMallocAsync((void**)&d_last, 5 * sizeof(long long), st));
    StreamFree free_last{d_last, st};
    RC_HIP_CHECK(hipMemsetAsync(d_last, 0xff, 5 * sizeof(long long), st));

    // global word coordinates: word 0 = key[0] of the caller's state; the next unread word is `pos`
    std::vector<unsigned int> carry(state->key, state->key + rcl::kMtN);   // block at g0
    long long g0 = 0;                                 // first global word of the current segment (multiple of 624)
    const long long w0 = state->pos;                  // first word of attempt 0
    long long t_next = 0, rank = 0;                   // attempts processed so far, accepted among them
    long long last[5] = {-1, 0, 0, 0, 0};
    unsigned int* raw = nullptr;
    long long raw_words = 0;
    StreamFree free_raw{nullptr, st};
    while (rank < pairs) {
        // attempts still expected (acceptance pi/4, eight-sigma margin), capped by the segment size
        const long long missing = pairs - rank;
        long long t_want = (long long)((double)missing / 0.78539816339744831 + 12.0 * sqrt((double)missing)) + 64;
        const long long first_word = w0 + 4 * t_next;                // global index; lies in the block at g0 or the one after
        long long words = (first_word - g0) + 4 * t_want;
        if (words > kLegacySegWords) words = kLegacySegWords;
        words = ((words + rcl::kMtN - 1) / rcl::kMtN) * rcl::kMtN;
        if (words < 2 * rcl::kMtN) words = 2 * rcl::kMtN;
        const long long t_count = (g0 + words - first_word) / 4;     // attempts wholly inside [g0, g0 + words)
        const long long cap = words;
        if (raw_words < cap) {
            if (raw) (void)hipFreeAsync(raw, st);
            raw = nullptr;
            free_raw.p = nullptr;
            RC_HIP_CHECK(hipMallocAsync((void**)&raw, (size_t)cap * sizeof(unsigned int), st));
            free_raw.p = raw;
            raw_words = cap;
        }
        if (int rc = mt_fill_raw(st, carry.data(), words, raw)) return rc;
        const long long nwg = (t_count + kLgAttempts - 1) / kLgAttempts;
        unsigned long long* d_counts = nullptr;
        RC_HIP_CHECK(hipMallocAsync((void**)&d_counts, (size_t)(nwg + 1) * sizeof(unsigned long long), st));
        StreamFree free_counts{d_counts, st};
        LegacyParams p{};
        p.raw = raw;
        p.w_first = first_word - g0;
        p.t_first = t_next;
        p.t_count = t_count;
        p.rank_base = rank;
        p.pairs_needed = pairs;
        p.e_shift = e_shift;
        p.n_total = n_total;
        p.period = period;
        p.skip = skip;
        p.scales = d_scales;
        p.out = out_dev;
        p.wg_counts = d_counts;
        p.last = d_last;
        hipLaunchKernelGGL(legacy_count_kernel, dim3((unsigned)nwg), dim3(kLgThreads), 0, st, p);
        hipLaunchKernelGGL(legacy_scan_kernel, dim3(1), dim3(1024), 0, st, d_counts, nwg);
        hipLaunchKernelGGL(legacy_emit_kernel, dim3((unsigned)nwg), dim3(kLgThreads), 0, st, p);
        RC_HIP_CHECK(hipGetLastError());
        unsigned long long accepted = 0;
        RC_HIP_CHECK(hipMemcpyAsync(&accepted, d_counts + nwg, sizeof(accepted), hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipMemcpyAsync(last, d_last, sizeof(last), hipMemcpyDeviceToHost, st));
        // the segment's last block is the next segment's first (an attempt may straddle the boundary)
        RC_HIP_CHECK(hipMemcpyAsync(carry.data(), raw + words - rcl::kMtN, rcl::kMtN * sizeof(unsigned int),
                                    hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipStreamSynchronize(st));
        rank += (long long)accepted;
        if (rank >= pairs) {
            // the generator stands right after the last needed attempt: global word wf
            const long long wf = w0 + 4 * (last[0] + 1);
            long long blk = wf / rcl::kMtN, pos = wf % rcl::kMtN;
            if (pos == 0) {                           // NumPy's representation of a block boundary: pos = 624 of the block before
                blk -= 1;
                pos = rcl::kMtN;
            }
            if (blk * rcl::kMtN < g0 || (blk + 1) * rcl::kMtN > g0 + words) return fail(RC_EHIP, "legacy stream: final block outside the segment");
            RC_HIP_CHECK(hipMemcpy(state->key, raw + (blk * rcl::kMtN - g0), rcl::kMtN * sizeof(unsigned int), hipMemcpyDeviceToHost));
            state->pos = (int)pos;
            if (n_need & 1) {
                // odd count: the second normal of the last attempt stays cached - computed HERE with the host's libm,
                // exactly as NumPy computes it (legacy-distributions.c: f = sqrt(-2 log(r2) / r2); gauss = f * x1)
                double x1, x2, r2;
                rcl::polar_attempt((unsigned)last[1], (unsigned)last[2], (unsigned)last[3], (unsigned)last[4], x1, x2, r2);
                const double f = sqrt(-2.0 * log(r2) / r2);
                state->gauss = f * x1;
                state->has_gauss = 1;
            }
            break;
        }
        t_next += t_count;
        g0 += words - rcl::kMtN;
    }
    return RC_OK;
}

// ------------------------------------------------------------------------------------------------
// driver of the device-side directional draws (rc_directional_draws_legacy_dev; kernels + scheme: k_draws.inc.h)
// ------------------------------------------------------------------------------------------------
constexpr long long kDirChunkSamples = 1LL << 23;     // samples per pass (5.7e7 raw words = 228 MB, 57 MB of lengths)

int directional_chunk(hipStream_t st, rc_mt19937_state* state, long long n, unsigned int rng, unsigned int mask, double sigma,
                      int* idx_dev, double* ab_dev) {
    // words per sample: (mask + 1) / (rng + 1) for the index + 4 / (pi / 4) for the accepted attempt; ten-sigma margin
    const double per = (rng ? (double)(mask + 1.0) / (double)(rng + 1.0) : 0.0) + 4.0 / 0.78539816339744831;
    double grow = 1.0;
    const long long ngroups = (n + kDirGroup - 1) / kDirGroup;
    std::vector<long long> starts((size_t)ngroups);
    for (int attempt = 0; attempt < 6; ++attempt, grow *= 1.5) {
        long long words = state->pos + (long long)(grow * ((double)n * per + 40.0 * sqrt((double)n) + 4096.0));
        words = ((words + rcl::kMtN - 1) / rcl::kMtN + 1) * rcl::kMtN;
        if (words < 2 * rcl::kMtN) words = 2 * rcl::kMtN;
        unsigned int* raw = nullptr;
        RC_HIP_CHECK(hipMallocAsync((void**)&raw, (size_t)words * sizeof(unsigned int), st));
        StreamFree free_raw{raw, st};
        if (int rc = mt_fill_raw(st, state->key, words, raw)) return rc;
        const long long first = state->pos;           // the first unread word (raw[0 .. 624) is the caller's block)
        const long long npos = words - first;
        unsigned char* d_len = nullptr;
        RC_HIP_CHECK(hipMallocAsync((void**)&d_len, (size_t)npos, st));
        StreamFree free_len{d_len, st};
        const dim3 pgrid((unsigned)((npos + 255) / 256));
        hipLaunchKernelGGL(dir_len_kernel, pgrid, dim3(256), 0, st, (const unsigned int*)raw, first, words, rng, mask, d_len);
        RC_HIP_CHECK(hipGetLastError());
        const int shift = state->has_gauss ? 1 : 0;
        if (shift) {                                  // a_0 = the cached normal the generator entered with
            const double a0 = 0.0 + sigma * state->gauss;
            RC_HIP_CHECK(hipMemcpyAsync(ab_dev, &a0, sizeof(double), hipMemcpyHostToDevice, st));
            RC_HIP_CHECK(hipStreamSynchronize(st));   // `a0` is a stack temporary
        }
        // ---- the walk over the sample chain ON THE DEVICE (k_draws.inc.h: dir_blk_kernel ...): nothing crosses PCIe but
        // the final state record.  RC_DIR_WALK (environment, read per call; A/B and test knob): "host" = the host walk below
        // only, "fallback" = the device pass runs and is then treated as failed (exercises the hand-over to the host walk).
        const char* walk_env = getenv("RC_DIR_WALK");
        const bool kHostWalk = walk_env && strcmp(walk_env, "host") == 0;
        const bool kForceFallback = walk_env && strcmp(walk_env, "fallback") == 0;
        const long long nblk = (npos + kDwB - 1) / kDwB;
        const int S = (nblk <= 64LL * kDwMaxSup) ? 64 : kDwS;          // blocks per superblock
        const long long nsup = (nblk + S - 1) / S;
        if (!kHostWalk && nsup <= kDwMaxSup) {
            // one allocation: blk_exit [nblk][E] u8 | blk_cnt [nblk][E] u16 | sup_exit [nsup][E] u8 | sup_cnt [nsup][E] u32 |
            // sup_base [nsup] i64 | blk_base [nblk] i64 | sup_entry [nsup] u8 | blk_entry [nblk] u8 | result record
            auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t o_bexit = 0, o_bcnt = o_bexit + up((size_t)nblk * kDwE), o_sexit = o_bcnt + up((size_t)nblk * kDwE * 2),
                         o_scnt = o_sexit + up((size_t)nsup * kDwE), o_sbase = o_scnt + up((size_t)nsup * kDwE * 4),
                         o_bbase = o_sbase + up((size_t)nsup * 8), o_sentry = o_bbase + up((size_t)nblk * 8),
                         o_bentry = o_sentry + up((size_t)nsup), o_res = o_bentry + up((size_t)nblk),
                         total = o_res + up(sizeof(DirWalkResult));
            unsigned char* ws = nullptr;
            RC_HIP_CHECK(hipMallocAsync((void**)&ws, total, st));
            StreamFree free_ws{ws, st};
            DirWalkResult* d_res = (DirWalkResult*)(ws + o_res);
            RC_HIP_CHECK(hipMemsetAsync(d_res, 0xff, sizeof(DirWalkResult), st));        // wf = -1: "not reached"
            hipLaunchKernelGGL(dir_blk_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const unsigned char*)d_len, npos,
                               ws + o_bexit, (unsigned short*)(ws + o_bcnt));
            hipLaunchKernelGGL(dir_sup_kernel, dim3((unsigned)nsup), dim3(64), 0, st, (const unsigned char*)(ws + o_bexit),
                               (const unsigned short*)(ws + o_bcnt), nblk, S, ws + o_sexit, (unsigned int*)(ws + o_scnt));
            hipLaunchKernelGGL(dir_top_kernel, dim3(1), dim3(64), 0, st, (const unsigned char*)(ws + o_sexit),
                               (const unsigned int*)(ws + o_scnt), (int)nsup, n, ws + o_sentry, (long long*)(ws + o_sbase));
            hipLaunchKernelGGL(dir_desc_kernel, dim3((unsigned)nsup), dim3(64), 0, st, (const unsigned char*)(ws + o_bexit),
                               (const unsigned short*)(ws + o_bcnt), nblk, S, n, (const unsigned char*)(ws + o_sentry),
                               (const long long*)(ws + o_sbase), ws + o_bentry, (long long*)(ws + o_bbase));
            DirEmitBlkParams bp{};
            bp.raw = raw;
            bp.len = d_len;
            bp.blk_entry = ws + o_bentry;
            bp.blk_base = (const long long*)(ws + o_bbase);
            bp.first = first;
            bp.npos = npos;
            bp.n = n;
            bp.rng = rng;
            bp.mask = mask;
            bp.shift = shift;
            bp.sigma = sigma;
            bp.idx = idx_dev;
            bp.ab = ab_dev;
            bp.res = d_res;
            hipLaunchKernelGGL(dir_emit_blk_kernel, dim3((unsigned)nblk), dim3(256), 0, st, bp);
            hipLaunchKernelGGL(dir_final_kernel, dim3(1), dim3(256), 0, st, (const unsigned int*)raw, first, words, d_res);
            RC_HIP_CHECK(hipGetLastError());
            static thread_local DirWalkResult* h_res = nullptr;       // pinned, one per host thread (never freed: see `pin` below)
            if (!h_res) RC_HIP_CHECK(hipHostMalloc((void**)&h_res, sizeof(DirWalkResult), hipHostMallocPortable));
            RC_HIP_CHECK(hipMemcpyAsync(h_res, d_res, sizeof(DirWalkResult), hipMemcpyDeviceToHost, st));
            RC_HIP_CHECK(hipStreamSynchronize(st));
            if (h_res->ok == 1 && !kForceFallback) {
                memcpy(state->key, h_res->key, rcl::kMtN * sizeof(unsigned int));
                state->pos = h_res->pos;
                if (shift) {
                    // the second normal of the last sample's attempt stays cached - computed with the host's libm, as NumPy does
                    double x1, x2, r2;
                    rcl::polar_attempt(h_res->last_words[0], h_res->last_words[1], h_res->last_words[2], h_res->last_words[3],
                                       x1, x2, r2);
                    const double f = sqrt(-2.0 * log(r2) / r2);
                    state->gauss = f * x1;
                    state->has_gauss = 1;
                }
                return RC_OK;
            }
            // not reached: the buffer was too short (next attempt: larger), or a sample longer than the walk follows across a
            // block boundary - the host walk below decides which; whatever the device pass wrote is overwritten
        }
        unsigned short* d_lenk = nullptr;
        RC_HIP_CHECK(hipMallocAsync((void**)&d_lenk, (size_t)npos * sizeof(unsigned short), st));
        StreamFree free_lenk{d_lenk, st};
        hipLaunchKernelGGL(dir_lenk_kernel, pgrid, dim3(256), 0, st, (const unsigned char*)d_len, npos, d_lenk);
        RC_HIP_CHECK(hipGetLastError());
        // group lengths to the host through a pinned buffer of this thread (grow-only; portable: the thread may serve several devices)
        // (never freed: a thread-exit destructor would call into the HIP runtime while the process may be tearing it down)
        static thread_local struct Pinned {
            void* p = nullptr;
            size_t bytes = 0;
        } pin;
        const size_t need = (size_t)npos * sizeof(unsigned short);
        if (pin.bytes < need) {
            if (pin.p) (void)hipHostFree(pin.p);
            pin.p = nullptr;
            pin.bytes = 0;
            RC_HIP_CHECK(hipHostMalloc(&pin.p, need + (need >> 2), hipHostMallocPortable));
            pin.bytes = need + (need >> 2);
        }
        const unsigned short* lenk = (const unsigned short*)pin.p;
        RC_HIP_CHECK(hipMemcpyAsync(pin.p, d_lenk, need, hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipStreamSynchronize(st));
        // the sequential part: one dependent load per group of kDirGroup samples ...
        long long p = 0, g = 0;
        const long long nfull = n / kDirGroup;
        for (; g < nfull; ++g) {
            if (p >= npos) break;
            const unsigned short l = lenk[(size_t)p];
            if (l == 0) break;                        // ran off the buffer (or a sample too long): more words needed
            starts[(size_t)g] = first + p;
            p += l;
        }
        if (g < nfull) continue;                      // (ten-sigma margin missed: larger buffer)
        // ... and the last, partial group sample by sample on a small window of the per-sample lengths
        const int rest = (int)(n - nfull * kDirGroup);
        if (rest) {
            if (p >= npos) continue;
            starts[(size_t)nfull] = first + p;
            unsigned char tail[kDirGroup * 256];
            const long long have = (npos - p < (long long)sizeof(tail)) ? (npos - p) : (long long)sizeof(tail);
            RC_HIP_CHECK(hipMemcpyAsync(tail, d_len + p, (size_t)have, hipMemcpyDeviceToHost, st));
            RC_HIP_CHECK(hipStreamSynchronize(st));
            long long o = 0;
            int j = 0;
            for (; j < rest; ++j) {
                if (o >= have || tail[o] == 0 || tail[o] == 255) break;
                o += tail[o];
            }
            if (j < rest) continue;
            p += o;
        }
        const long long wf = first + p;               // the generator stands here afterwards
        long long* d_starts = nullptr;
        RC_HIP_CHECK(hipMallocAsync((void**)&d_starts, (size_t)ngroups * sizeof(long long), st));
        StreamFree free_starts{d_starts, st};
        unsigned int* d_last = nullptr;
        RC_HIP_CHECK(hipMallocAsync((void**)&d_last, 4 * sizeof(unsigned int), st));
        StreamFree free_last{d_last, st};
        RC_HIP_CHECK(hipMemcpyAsync(d_starts, starts.data(), (size_t)ngroups * sizeof(long long), hipMemcpyHostToDevice, st));
        DirEmitParams ep{};
        ep.raw = raw;
        ep.len = d_len;
        ep.starts = d_starts;
        ep.first = first;
        ep.n = n;
        ep.rng = rng;
        ep.mask = mask;
        ep.shift = shift;
        ep.sigma = sigma;
        ep.idx = idx_dev;
        ep.ab = ab_dev;
        ep.last_words = d_last;
        hipLaunchKernelGGL(dir_emit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ep);
        RC_HIP_CHECK(hipGetLastError());
        long long blk = wf / rcl::kMtN, pos = wf % rcl::kMtN;
        if (pos == 0) {                               // NumPy's representation of a block boundary: pos = 624 of the block before
            blk -= 1;
            pos = rcl::kMtN;
        }
        if ((blk + 1) * rcl::kMtN > words) return fail(RC_EHIP, "directional draws: final block outside the buffer");
        unsigned int lastw[4] = {0, 0, 0, 0};
        RC_HIP_CHECK(hipMemcpyAsync(lastw, d_last, sizeof(lastw), hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipMemcpyAsync(state->key, raw + blk * rcl::kMtN, rcl::kMtN * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        RC_HIP_CHECK(hipStreamSynchronize(st));
        state->pos = (int)pos;
        if (shift) {
            // the second normal of the last sample's attempt stays cached - computed with the host's libm, as NumPy does
            double x1, x2, r2;
            rcl::polar_attempt(lastw[0], lastw[1], lastw[2], lastw[3], x1, x2, r2);
            const double f = sqrt(-2.0 * log(r2) / r2);
            state->gauss = f * x1;
            state->has_gauss = 1;
        }
        return RC_OK;
    }
    return fail(RC_EHIP, "directional draws: word budget exceeded six times");
}

// ------------------------------------------------------------------------------------------------
// driver of the directional model's fidelity pass (rc_mc_fidelity_directional_f64_async; kernels: k_directional.inc.h)
// ------------------------------------------------------------------------------------------------
constexpr long long kDirFidChunk = (long long)kDirPartThreads * 8192;     // samples per partition pass (see dir_class_scatter_kernel)

template <int N>
int launch_directional(hipStream_t s, const DirParams& p, bool ends) {
    const long long nwaves = (p.n + 63) / 64;
    if (nwaves > 0x7fffffffLL) return fail(RC_EINVAL, "too many samples for one launch");
    const dim3 grid((unsigned)nwaves);                        // worst case for either class; waves beyond a class's count end at once
    if (ends) hipLaunchKernelGGL((mc_fid_dir_bond_kernel<N, rc::kWeightsEnds>), grid, dim3(64), 0, s, p);
    else hipLaunchKernelGGL((mc_fid_dir_bond_kernel<N, rc::kWeightsAdjugate>), grid, dim3(64), 0, s, p);
    hipLaunchKernelGGL(mc_fid_dir_diag_kernel<N>, grid, dim3(64), 0, s, p);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int enqueue_directional(hipStream_t s, int N, int in, int out, const double* h0_diag, const double* h0_offdiag,
                        const double* ctrl, const int* idx, const double* ab, long long C, long long K, double* fid) {
    const bool ends = (in == 0 && out == N - 1) || (in == N - 1 && out == 0);
    const long long total = C * K;
    // RC_DIR_FID_CHUNK (environment, read per call; TEST knob): samples per partition pass, rounded down to whole waves - lets
    // a small problem exercise the multi-chunk path (chunk-relative idx / ab / fid pointers, p.first, the expm list's sp_first,
    // one stream-ordered workspace per chunk) that otherwise needs C K > 2^23
    long long chunk = kDirFidChunk;
    if (const char* e = getenv("RC_DIR_FID_CHUNK")) {
        const long long v = atoll(e) / 64 * 64;
        if (v >= 64 && v < chunk) chunk = v;
    }
    for (long long done = 0; done < total; done += chunk) {
        const long long n = (total - done < chunk) ? (total - done) : chunk;
        const long long nblocks = (n + kDirPartThreads - 1) / kDirPartThreads;
        // workspace of THIS chunk, allocated and released in stream order: list + marked (n ints each), block counts, counters
        char* ws = nullptr;
        const size_t nb_list = ((size_t)n * sizeof(int) + 255) & ~(size_t)255;
        const size_t nb_blk = ((size_t)nblocks * sizeof(unsigned int) + 255) & ~(size_t)255;
        RC_HIP_CHECK(hipMallocAsync((void**)&ws, 2 * nb_list + nb_blk + 256, s));
        StreamFree free_ws{ws, s};
        DirParams p{};
        // chunk-relative sample indices: pointers are offset, the controller index needs the chunk's first sample
        p.ctrl = ctrl;
        p.idx = idx + done;
        p.ab = ab + 2 * done;
        p.fid = fid + done;
        p.K = K;
        p.n = n;
        p.in = in;
        p.out = out;
        p.list = (int*)ws;
        p.marked = (int*)(ws + nb_list);
        p.blk_counts = (unsigned int*)(ws + 2 * nb_list);
        p.counts = (unsigned int*)(ws + 2 * nb_list + nb_blk);
        p.first = done;
        for (int i = 0; i < RC_MAX_NSPIN; ++i) {
            p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
            p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
        }
        RC_HIP_CHECK(hipMemsetAsync(p.counts, 0, 256, s));
        hipLaunchKernelGGL(dir_class_count_kernel, dim3((unsigned)nblocks), dim3(kDirPartThreads), 0, s, p, N);
        hipLaunchKernelGGL(dir_class_scatter_kernel, dim3((unsigned)nblocks), dim3(kDirPartThreads), 0, s, p, N);
        int rc = RC_EINVAL;
        switch (N) {
#define RC_DIR_CASE(n) case n: rc = launch_directional<n>(s, p, ends); break;
            RC_DIR_CASE(2) RC_DIR_CASE(3) RC_DIR_CASE(4) RC_DIR_CASE(5) RC_DIR_CASE(6) RC_DIR_CASE(7) RC_DIR_CASE(8) RC_DIR_CASE(9)
            RC_DIR_CASE(10) RC_DIR_CASE(11) RC_DIR_CASE(12)
#undef RC_DIR_CASE
        }
        if (rc) return rc;
        // the expm pass over the samples neither route settled (normally none: its waves read a zero count and end)
        ExpmParams e{};
        e.ctrl = ctrl;
        e.fid = fid + done;
        e.C = C;
        e.K = K;
        e.N = N;
        e.in = in;
        e.out = out;
        e.sp_idx = p.idx;
        e.sp_ab = p.ab;
        e.sp_list = p.marked;
        e.sp_count = p.counts + 1;
        e.sp_first = done;
        e.h0 = p.h0;
        if (int rc2 = ensure_func_attr(kAttrExpm, (const void*)mc_fid_expm_kernel,
                                       kExpmWaves * kExpmBufs * RC_MAX_NSPIN_FAST * RC_MAX_NSPIN_FAST * (int)sizeof(cplx)))
            return rc2;
        hipLaunchKernelGGL(mc_fid_expm_kernel, dim3(64), dim3(64 * kExpmWaves), (size_t)kExpmWaves * kExpmBufs * N * N * sizeof(cplx), s, e);
        RC_HIP_CHECK(hipGetLastError());
    }
    return RC_OK;
}

// ------------------------------------------------------------------------------------------------
// RCCL over xGMI from the C ABI (rc_comm_init / rc_mc_metrics_gathered_f64, ABI 6)
// ------------------------------------------------------------------------------------------------
// north_star's exchange step - "an RCCL all-gather over xGMI to reassemble per-controller fidelity vectors" - without torch: one
// process, one communicator per listed device (ncclCommInitAll), the collective enqueued on every device's stream inside ONE
// group call.  librccl.so is resolved at RUN time (dlsym on what the process already carries - PyTorch ships its own copy - or
// dlopen of librccl.so.1 / librccl.so): the library keeps no link-time dependency on it and every other entry works without it.
typedef int (*rccl_init_all_t)(void**, int, const int*);
typedef int (*rccl_destroy_t)(void*);
typedef int (*rccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*rccl_group_t)(void);
typedef const char* (*rccl_errstr_t)(int);
struct RcclApi {
    rccl_init_all_t init_all = nullptr;
    rccl_destroy_t destroy = nullptr;
    rccl_allgather_t allgather = nullptr;
    rccl_group_t group_start = nullptr, group_end = nullptr;
    rccl_errstr_t errstr = nullptr;
    bool ok = false;
};
constexpr int kNcclFloat64 = 8;                  // rccl.h: ncclFloat64

const RcclApi& rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        void* h = nullptr;                           // RTLD_DEFAULT first: the copy the process already loaded (PyTorch's)
        auto sym = [&h](const char* n) -> void* {
            void* p = dlsym(RTLD_DEFAULT, n);
            if (p) return p;
            if (!h) {
                // RTLD_LOCAL | RTLD_DEEPBIND, never RTLD_GLOBAL: a process that resolves RCCL here (nothing has loaded one yet)
                // and imports PyTorch LATER gets torch's own librccl.so / librocm_smi64.so as well; with the system copies in
                // the global scope the two librocm_smi64 share their global objects by symbol interposition and the process
                // ends in glibc's "double free or corruption" inside rocm_smi's static destructors (seen: pytest running the
                // RCCL test before any test had imported torch).  Kept out of the global scope, each copy binds to itself.
                for (const char* lib : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                    h = dlopen(lib, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
                    if (h) break;
                }
            }
            return h ? dlsym(h, n) : nullptr;
        };
        a.init_all = (rccl_init_all_t)sym("ncclCommInitAll");
        a.destroy = (rccl_destroy_t)sym("ncclCommDestroy");
        a.allgather = (rccl_allgather_t)sym("ncclAllGather");
        a.group_start = (rccl_group_t)sym("ncclGroupStart");
        a.group_end = (rccl_group_t)sym("ncclGroupEnd");
        a.errstr = (rccl_errstr_t)sym("ncclGetErrorString");
        a.ok = a.init_all && a.destroy && a.allgather && a.group_start && a.group_end;
        return a;
    }();
    return api;
}

}  // namespace

struct rc_comm {
    int ndev = 0;
    std::vector<int> devices;
    std::vector<void*> comms;                        // ncclComm_t per device
};

namespace {

int rccl_fail(int code, const char* what) {
    const RcclApi& a = rccl_api();
    return fail(RC_EHIP, std::string(what) + ": " + (a.errstr ? a.errstr(code) : "RCCL error " + std::to_string(code)));
}

// One device's share of rc_mc_metrics_gathered_f64: controllers [c0, c1) -> fidelities into rows [0, c1 - c0) of `d_fid` (Cmax rows,
// the rest zero) and their metric rows into `d_tab` ([NR][Cmax]).  Everything is enqueued on the device's stream; nothing waits.
struct GatherJob {
    int device, kernel, N, in, out, ring;
    const double *h0d, *h0o, *ctrl, *draws;
    unsigned long long seed, offset;
    double sigma;
    long long C, K, c0, c1, Cmax;
    const double* thr;
    int nq;
    double eps;
    double *d_ctrl, *d_draw, *d_fid, *d_tab;         // device: ctrl [Cmax][N+1], draws (host draws only), fid [Cmax][K], table [NR][Cmax]
    int rc = RC_OK;
    std::string err;
};

int run_gather_share(GatherJob* j) {
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(j->device, &ctx)) return rc;
    hipStream_t st = ctx->stream;
    const long long G = 3LL * j->N, K = j->K, Cl = j->c1 - j->c0;
    RC_HIP_CHECK(hipMemsetAsync(j->d_fid, 0, (size_t)j->Cmax * K * sizeof(double), st));
    if (Cl > 0) {
        RC_HIP_CHECK(hipMemcpyAsync(j->d_ctrl, j->ctrl + j->c0 * (j->N + 1), (size_t)Cl * (j->N + 1) * sizeof(double),
                                    hipMemcpyHostToDevice, st));
        const bool fused = !j->draws && !j->ring && rc_philox_fused_pays(j->N, j->in, j->out) == 1 &&
                           (j->kernel == RC_KERNEL_AUTO || j->kernel == RC_KERNEL_TRIDIAG_ADJ);
        if (fused) {
            if (int rc = rc_mc_fidelity_philox_f64_async(j->device, st, j->kernel, j->N, j->in, j->out, j->h0d, j->h0o, j->d_ctrl,
                                                         j->seed, j->offset + (unsigned long long)(j->c0 * K * G), j->sigma, nullptr,
                                                         Cl, K, j->d_fid))
                return rc;
        } else {
            if (j->draws) {
                RC_HIP_CHECK(hipMemcpyAsync(j->d_draw, j->draws + j->c0 * K * G, (size_t)Cl * K * G * sizeof(double),
                                            hipMemcpyHostToDevice, st));
            } else if (int rc = rc_draws_philox_f64_async(j->device, st, j->seed, j->offset + (unsigned long long)(j->c0 * K * G),
                                                          Cl * K * G, j->sigma, j->d_draw)) {
                return rc;
            }
            if (int rc = enqueue_fidelity(st, j->kernel, j->N, j->in, j->out, j->h0d, j->h0o, j->ring, j->d_ctrl, j->d_draw, -1, Cl, K,
                                          j->d_fid))
                return rc;
        }
    }
    // metric rows of all Cmax rows (the padding rows are rows of zeros: harmless, and the table keeps one shape on every device)
    double* t = j->d_tab;
    return enqueue_reduce(st, j->d_fid, j->Cmax, K, j->thr, j->nq, j->eps, t, t + 3 * j->Cmax, t + 6 * j->Cmax,
                          j->nq ? t + 9 * j->Cmax : nullptr, nullptr, /*standalone=*/true);
}

void run_gather_share_locked(GatherJob* j) {
    std::lock_guard<std::mutex> lk(g_ctx[j->device].mu);
    j->rc = run_gather_share(j);
    if (j->rc) j->err = g_last_error;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// sum of the `slots` unsigned 64-bit counters behind a device symbol (synchronises the device); optionally zeroes them
template <typename Sym>
static long long read_tile_counter(int device, int reset, const Sym& symbol, int slots) {
    if (hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EHIP, "hipSetDevice failed");
    }
    unsigned long long v[64] = {0};
    void* addr = nullptr;
    if (hipDeviceSynchronize() != hipSuccess || hipGetSymbolAddress(&addr, HIP_SYMBOL(symbol)) != hipSuccess ||
        hipMemcpy(v, addr, sizeof(unsigned long long) * slots, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EHIP, "reading the tile counter failed");
    }
    if (reset && hipMemset(addr, 0, sizeof(unsigned long long) * slots) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EHIP, "resetting the tile counter failed");
    }
    unsigned long long sum = 0;
    for (int i = 0; i < slots; ++i) sum += v[i];
    // the same counter of the second translation unit (robchar_large.hip: chains of 17 .. 24 spins, rings of 11 .. 16)
    void* addr2 = nullptr;
    if (rc_large_counter_addr(slots == 1 ? 0 : 1, &addr2) != hipSuccess || !addr2 ||
        hipMemcpy(v, addr2, sizeof(unsigned long long) * slots, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EHIP, "reading the tile counter (second unit) failed");
    }
    if (reset && hipMemset(addr2, 0, sizeof(unsigned long long) * slots) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EHIP, "resetting the tile counter (second unit) failed");
    }
    for (int i = 0; i < slots; ++i) sum += v[i];
    return (long long)sum;
}

extern "C" {

#ifdef RC_STAMPS
// diagnostic builds only: device buffer of [ntiles][4] int64 receiving per-wave s_memtime stamps
int rc_debug_set_stamps(long long* dev_buf) { g_stamps = dev_buf; return 0; }
#endif

int rc_version(void) { return RC_ABI_VERSION; }

// Which compile-time switches of this build change RESULTS or disable a safety net (0 = the product build).  The timing
// experiments of scripts/build_variant.sh knowingly return wrong fidelities for some samples; a library built with one of them
// must never be taken for the product (code-robchar_amd/_lib.py refuses it unless ROBCHAR_ALLOW_EXPERIMENT_LIB=1; bench.py
// records the value).
int rc_build_flags(void) {
    int f = 0;
#ifdef RC_EXPERIMENT_NO_STEPPING
    f |= RC_BUILD_EXPERIMENT_NO_STEPPING;
#endif
#ifdef RC_EXPERIMENT_STEP_NOT_RUN
    f |= RC_BUILD_EXPERIMENT_STEP_NOT_RUN;
#endif
#ifdef RC_EXPERIMENT_FALLBACK_NOT_RUN
    f |= RC_BUILD_EXPERIMENT_FALLBACK_NOT_RUN;
#endif
#ifdef RC_EXPERIMENT_PHILOX_NOSTORE
    f |= RC_BUILD_EXPERIMENT_PHILOX_NOSTORE;
#endif
#ifdef RC_DEV_FEW_N
    f |= RC_BUILD_DEV_FEW_N;
#endif
#ifdef RC_STAMPS
    f |= RC_BUILD_STAMPS;
#endif
    if (!rc::kSumRuleGuard) f |= RC_BUILD_NO_SUM_RULE_GUARD;
    if (!RC_KEEP_SETTLED) f |= RC_BUILD_NO_KEEP_SETTLED;
    return f;
}

// Where the fused Philox fidelity kernel is the faster of the two bit-identical routes (profiles/r04_philox_fused_sweep.txt:
// 0.70 .. 0.86 of the two-kernel route up to N = 13 and for end-to-end pairs at N = 14; beyond, the fused instantiations run
// one wave per SIMD and the two-kernel route is 7 % faster).  ROBCHAR_PHILOX_FUSED=0 in the environment (read per call)
// switches the route off everywhere - here, in the sharded entries and in the Python layer, which all ask this function.
int rc_philox_fused_pays(int N, int in, int out) {
    if (N < 2 || N > RC_MAX_NSPIN_FAST) return 0;
    const char* e = getenv("ROBCHAR_PHILOX_FUSED");
    if (e && e[0] == '0') return 0;
    const bool ends = (in == 0 && out == N - 1) || (in == N - 1 && out == 0);
    return (N <= 13 || (N == 14 && ends)) ? 1 : 0;
}

int rc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* rc_last_error(void) { return g_last_error.c_str(); }

long long rc_stats_general_tiles(int device, int reset) { return read_tile_counter(device, reset, g_general_tiles, 1); }
long long rc_stats_polish_tiles(int device, int reset) { return read_tile_counter(device, reset, g_polish_tiles, 64); }

int rc_set_fidelity_kernel(int kernel) {
    if (kernel < RC_KERNEL_AUTO || kernel > RC_KERNEL_RING_HH) return fail(RC_EINVAL, "unknown kernel id");
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    g_default_kernel = kernel;
    return RC_OK;
}

int rc_reserve_ring(int device, void* stream, long long samples) {
    if (samples < 0) return fail(RC_EINVAL, "samples must be non-negative");
    if (int rc = device_in_range(device)) return rc;
    RC_HIP_CHECK(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_ctx[device].ring_mu);
    return ring_buf_reserve(ring_buf_for(g_ctx[device], (hipStream_t)stream), (hipStream_t)stream, samples);
}

int rc_release_stream(int device, void* stream) {
    if (int rc = device_in_range(device)) return rc;
    RC_HIP_CHECK(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_ctx[device].ring_mu);
    auto it = g_ctx[device].ring_bufs.find((hipStream_t)stream);
    if (it == g_ctx[device].ring_bufs.end()) return RC_OK;
    char* mem = it->second.mem;
    g_ctx[device].ring_bufs.erase(it);
    if (mem) RC_HIP_CHECK(hipFreeAsync(mem, (hipStream_t)stream));     // behind the stream's last ring launch
    return RC_OK;
}

int rc_mc_fidelity_f64_async(int device, void* stream, int kernel, int N, int in, int out,
                             const double* h0_diag, const double* h0_offdiag, int ring,
                             const double* controllers_dev, const double* draws_dev, long long C,
                             long long K, double* fid_out_dev) {
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_fidelity((hipStream_t)stream, kernel, N, in, out, h0_diag, h0_offdiag, ring,
                            controllers_dev, draws_dev, -1, C, K, fid_out_dev);
}

int rc_mc_fidelity_ex_f64_async(int device, void* stream, int kernel, int N, int in, int out,
                                const double* h0_diag, const double* h0_offdiag, int ring,
                                const double* controllers_dev, const double* draws_dev, long long draws_ctrl_stride,
                                long long C, long long K, double* fid_out_dev) {
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_fidelity((hipStream_t)stream, kernel, N, in, out, h0_diag, h0_offdiag, ring,
                            controllers_dev, draws_dev, draws_ctrl_stride, C, K, fid_out_dev);
}

int rc_mc_fidelity_philox_f64_async(int device, void* stream, int kernel, int N, int in, int out, const double* h0_diag,
                                    const double* h0_offdiag, const double* controllers_dev, unsigned long long seed,
                                    unsigned long long offset, double sigma, const double* sigma_rows_dev, long long C,
                                    long long K, double* fid_out_dev) {
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (C == 0 || K == 0) return RC_OK;
    if (!controllers_dev || !fid_out_dev) return fail(RC_EINVAL, "NULL array pointer");
    if (N > RC_MAX_NSPIN_FAST) return fail(RC_ENOSUP, "draws generated in the fidelity kernel: N <= 16 (longer chains: rc_draws_philox_f64_async + rc_mc_fidelity_f64_async)");
    if (kernel != RC_KERNEL_AUTO && kernel != RC_KERNEL_TRIDIAG_ADJ)
        return fail(RC_ENOSUP, "draws generated in the fidelity kernel: the eigenvalue-only chain kernels only (RC_KERNEL_AUTO / RC_KERNEL_TRIDIAG_ADJ)");
    RC_HIP_CHECK(hipSetDevice(device));
    const bool ends = (in == 0 && out == N - 1) || (in == N - 1 && out == 0);
    FidParams p{};
    p.ctrl = controllers_dev;
    p.draws = nullptr;
    p.fid = fid_out_dev;
    p.C = C;
    p.K = K;
    p.draw_cstride = 0;
    p.tiles_per_ctrl = (K + 63) / 64;
    p.ntiles = C * p.tiles_per_ctrl;
    p.in = in;
    p.out = out;
    p.align16 = 0;
    for (int i = 0; i < RC_MAX_NSPIN; ++i) {
        p.h0.diag[i] = (h0_diag && i < N) ? h0_diag[i] : 0.0;
        p.h0.off[i] = (i < N - 1) ? (h0_offdiag ? h0_offdiag[i] : 1.0) : 0.0;
    }
    PhiloxDraws q{seed, offset, sigma_rows_dev, sigma};
    hipStream_t s = (hipStream_t)stream;
    switch (N) {
#define RC_CASE(n) \
    case n: return (ends && n <= 14) ? launch_chain_philox<n, rc::kWeightsEnds>(s, p, q)  /* (same choice as enqueue_fidelity: bit-identical) */ \
                                     : launch_chain_philox<n, rc::kWeightsAdjugate>(s, p, q);
#ifdef RC_DEV_FEW_N
        RC_CASE(5) RC_CASE(7) RC_CASE(10)
#else
        RC_CASE(2) RC_CASE(3) RC_CASE(4) RC_CASE(5) RC_CASE(6) RC_CASE(7) RC_CASE(8) RC_CASE(9)
        RC_CASE(10) RC_CASE(11) RC_CASE(12) RC_CASE(13) RC_CASE(14) RC_CASE(15) RC_CASE(16)
#endif
#undef RC_CASE
    }
    return fail(RC_EINVAL, "unsupported N");
}

int rc_mc_fidelity_nh_f64_async(int device, void* stream, int N, int in, int out, const double* h0_diag,
                                const double* h0_offdiag, int ring, const double* controllers_dev,
                                const double* draws_dev, const double* diag_imag_dev, long long C, long long K,
                                double* fid_out_dev) {
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (C == 0 || K == 0) return RC_OK;
    if (!controllers_dev || !draws_dev || !fid_out_dev) return fail(RC_EINVAL, "NULL array pointer");
    RC_HIP_CHECK(hipSetDevice(device));
    // (RC_NH_EXPM_ONLY=1 in the environment: the dense Pade-expm kernel for every sample - the cross-check of the
    // complex symmetric QL route)
    const char* nh_env = getenv("RC_NH_EXPM_ONLY");
    const bool expm_only = nh_env && nh_env[0] == '1';
    return enqueue_expm((hipStream_t)stream, N, in, out, h0_diag, h0_offdiag, ring, controllers_dev, draws_dev,
                        K * N * 3, diag_imag_dev, K * N, C, K, fid_out_dev, !expm_only);
}

int rc_mc_fidelity_directional_f64_async(int device, void* stream, int N, int in, int out, const double* h0_diag,
                                         const double* h0_offdiag, int ring, const double* controllers_dev,
                                         const int* idx_dev, const double* ab_dev, long long C, long long K,
                                         double* fid_out_dev) {
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (ring && N > 2) return fail(RC_ENOSUP, "the directional entry handles chain topology (rings: the dense layout + rc_mc_fidelity_nh_f64_async)");
    if (N > kDirMaxN) return fail(RC_ENOSUP, "the directional entry supports N <= 12 (above: the dense layout + rc_mc_fidelity_nh_f64_async)");
    if (C == 0 || K == 0) return RC_OK;
    if (!controllers_dev || !idx_dev || !ab_dev || !fid_out_dev) return fail(RC_EINVAL, "NULL array pointer");
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_directional((hipStream_t)stream, N, in, out, h0_diag, h0_offdiag, controllers_dev, idx_dev, ab_dev, C, K,
                               fid_out_dev);
}

int rc_mc_fidelity_f64(int device, int N, int in, int out, const double* h0_diag,
                       const double* h0_offdiag, int ring, const double* controllers, const double* draws,
                       long long C, long long K, double* fid_out) {
    int kernel;
    {
        std::lock_guard<std::mutex> lk(g_cfg_mu);
        kernel = g_default_kernel;
    }
    return rc_mc_fidelity_kernel_f64(device, kernel, N, in, out, h0_diag, h0_offdiag, ring, controllers, draws, C, K,
                                     fid_out);
}

int rc_mc_fidelity_kernel_f64(int device, int kernel, int N, int in, int out, const double* h0_diag,
                              const double* h0_offdiag, int ring, const double* controllers, const double* draws,
                              long long C, long long K, double* fid_out) {
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (C == 0 || K == 0) return RC_OK;
    if (!controllers || !draws || !fid_out) return fail(RC_EINVAL, "NULL array pointer");
    if (int rc = device_in_range(device)) return rc;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(device, &ctx)) return rc;
    const size_t nb_ctrl = (size_t)C * (N + 1) * sizeof(double);
    const size_t nb_draw = (size_t)C * K * N * 3 * sizeof(double);
    const size_t nb_fid = (size_t)C * K * sizeof(double);
    const bool dc = is_device_ptr(controllers), dd = is_device_ptr(draws), df = is_device_ptr(fid_out);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t need = (dc ? 0 : up(nb_ctrl)) + (dd ? 0 : up(nb_draw)) + (df ? 0 : up(nb_fid));
    if (need) {
        if (int rc = ensure_ws(ctx, need)) return rc;
    }
    char* w = (char*)ctx->ws;
    const double* d_ctrl = controllers;
    const double* d_draw = draws;
    double* d_fid = fid_out;
    if (!dc) {
        RC_HIP_CHECK(hipMemcpyAsync(w, controllers, nb_ctrl, hipMemcpyHostToDevice, ctx->stream));
        d_ctrl = (const double*)w;
        w += up(nb_ctrl);
    }
    if (!dd) {
        RC_HIP_CHECK(hipMemcpyAsync(w, draws, nb_draw, hipMemcpyHostToDevice, ctx->stream));
        d_draw = (const double*)w;
        w += up(nb_draw);
    }
    if (!df) d_fid = (double*)w;
    if (int rc = enqueue_fidelity(ctx->stream, kernel, N, in, out, h0_diag, h0_offdiag, ring, d_ctrl, d_draw, -1, C, K,
                                  d_fid))
        return rc;
    if (!df) RC_HIP_CHECK(hipMemcpyAsync(fid_out, d_fid, nb_fid, hipMemcpyDeviceToHost, ctx->stream));
    RC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_reduce_f64_async(int device, void* stream, const double* fid_dev, long long C, long long K,
                        const double* q_thresholds, int nq, double dkw_eps, double* rim1_dev,
                        double* std_dev, double* minf_dev, double* q_dev, double* sorted_out_dev) {
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_reduce((hipStream_t)stream, fid_dev, C, K, q_thresholds, nq, dkw_eps, rim1_dev, std_dev,
                          minf_dev, q_dev, sorted_out_dev);
}

int rc_reduce_ex_f64_async(int device, void* stream, const double* fid_dev, long long C, long long K,
                           const double* q_thresholds, int nq, double dkw_eps, double* rim1_dev, double* std_dev,
                           double* minf_dev, double* q_dev, double* sorted_out_dev, int flags) {
    if (flags & ~RC_REDUCE_STANDALONE) return fail(RC_EINVAL, "unknown reduction flag");
    RC_HIP_CHECK(hipSetDevice(device));
    return enqueue_reduce((hipStream_t)stream, fid_dev, C, K, q_thresholds, nq, dkw_eps, rim1_dev, std_dev, minf_dev, q_dev,
                          sorted_out_dev, (flags & RC_REDUCE_STANDALONE) != 0);
}

int rc_reduce_f64(int device, const double* fid, long long C, long long K, const double* q_thresholds,
                  int nq, double dkw_eps, double* rim1, double* std_, double* minf, double* q,
                  double* sorted_out) {
    if (C < 0 || K < 0) return fail(RC_EINVAL, "C and K must be non-negative");
    if (nq < 0 || nq > kMaxQ) return fail(RC_EINVAL, "nq must be in [0, 8]");
    if (C == 0) return RC_OK;
    if (!fid) return fail(RC_EINVAL, "NULL fid pointer");
    if (int rc = device_in_range(device)) return rc;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(device, &ctx)) return rc;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nb_fid = (size_t)C * K * sizeof(double);
    const size_t nb_c3 = (size_t)3 * C * sizeof(double);
    const size_t nb_q = (size_t)3 * (nq > 0 ? nq : 1) * C * sizeof(double);
    const bool df = is_device_ptr(fid);
    // outputs are always staged (they are small), the sorted tensor only when it is a host pointer
    const bool ds = sorted_out && is_device_ptr(sorted_out);
    size_t need = (df ? 0 : up(nb_fid)) + 3 * up(nb_c3) + up(nb_q) + ((sorted_out && !ds) ? up(nb_fid) : 0);
    if (int rc = ensure_ws(ctx, need)) return rc;
    char* w = (char*)ctx->ws;
    const double* d_fid = fid;
    if (!df) {
        RC_HIP_CHECK(hipMemcpyAsync(w, fid, nb_fid, hipMemcpyHostToDevice, ctx->stream));
        d_fid = (const double*)w;
        w += up(nb_fid);
    }
    double* d_rim = (double*)w; w += up(nb_c3);
    double* d_std = (double*)w; w += up(nb_c3);
    double* d_min = (double*)w; w += up(nb_c3);
    double* d_q = (double*)w;   w += up(nb_q);
    double* d_sorted = nullptr;
    if (sorted_out) d_sorted = ds ? sorted_out : (double*)w;
    if (int rc = enqueue_reduce(ctx->stream, d_fid, C, K, q_thresholds, nq, dkw_eps, rim1 ? d_rim : nullptr,
                                std_ ? d_std : nullptr, minf ? d_min : nullptr, (q && nq) ? d_q : nullptr,
                                d_sorted, /*standalone=*/true))
        return rc;
    if (rim1) RC_HIP_CHECK(hipMemcpyAsync(rim1, d_rim, nb_c3, hipMemcpyDefault, ctx->stream));
    if (std_) RC_HIP_CHECK(hipMemcpyAsync(std_, d_std, nb_c3, hipMemcpyDefault, ctx->stream));
    if (minf) RC_HIP_CHECK(hipMemcpyAsync(minf, d_min, nb_c3, hipMemcpyDefault, ctx->stream));
    if (q && nq) RC_HIP_CHECK(hipMemcpyAsync(q, d_q, (size_t)3 * nq * C * sizeof(double), hipMemcpyDefault, ctx->stream));
    if (sorted_out && !ds) RC_HIP_CHECK(hipMemcpyAsync(sorted_out, d_sorted, nb_fid, hipMemcpyDeviceToHost, ctx->stream));
    RC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_rim_p_f64_async(int device, void* stream, const double* fid_dev, long long C, long long K, double p,
                       double* out_dev) {
    if (C < 0 || K <= 0) return fail(RC_EINVAL, "C must be >= 0 and K > 0");
    if (!(p > 0.0)) return fail(RC_EINVAL, "p must be positive (RIM_0 = 1 is a host constant)");
    if (C == 0) return RC_OK;
    if (!fid_dev || !out_dev) return fail(RC_EINVAL, "NULL array pointer");
    if (C > 0x7fffffffLL) return fail(RC_EINVAL, "too many controllers for one launch");
    RC_HIP_CHECK(hipSetDevice(device));
    hipLaunchKernelGGL(rim_p_kernel, dim3((unsigned)C), dim3(kRedThreads), 0, (hipStream_t)stream, fid_dev, C, K,
                       p, out_dev);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_rim_p_f64(int device, const double* fid, long long C, long long K, double p, double* out) {
    if (C < 0 || K <= 0) return fail(RC_EINVAL, "C must be >= 0 and K > 0");
    if (!(p > 0.0)) return fail(RC_EINVAL, "p must be positive (RIM_0 = 1 is a host constant)");
    if (C == 0) return RC_OK;
    if (!fid || !out) return fail(RC_EINVAL, "NULL array pointer");
    if (int rc = device_in_range(device)) return rc;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(device, &ctx)) return rc;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nb_fid = (size_t)C * K * sizeof(double), nb_out = (size_t)C * sizeof(double);
    const bool df = is_device_ptr(fid);
    if (int rc = ensure_ws(ctx, (df ? 0 : up(nb_fid)) + up(nb_out))) return rc;
    char* w = (char*)ctx->ws;
    const double* d_fid = fid;
    if (!df) {
        RC_HIP_CHECK(hipMemcpyAsync(w, fid, nb_fid, hipMemcpyHostToDevice, ctx->stream));
        d_fid = (const double*)w;
        w += up(nb_fid);
    }
    if (int rc = rc_rim_p_f64_async(device, ctx->stream, d_fid, C, K, p, (double*)w)) return rc;
    RC_HIP_CHECK(hipMemcpyAsync(out, w, nb_out, hipMemcpyDefault, ctx->stream));
    RC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_draws_philox_f64_async(int device, void* stream, unsigned long long seed, unsigned long long offset,
                              long long n, double scale, double* out_dev) {
    if (n < 0) return fail(RC_EINVAL, "n must be non-negative");
    if (n == 0) return RC_OK;
    if (!out_dev) return fail(RC_EINVAL, "NULL output pointer");
    RC_HIP_CHECK(hipSetDevice(device));
    long long blocks = (n / 2 + 1 + 255) / 256;               // one thread per Box-Muller pair
    if (blocks > 256LL * 32) blocks = 256LL * 32;
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seed, offset,
                       n, scale, out_dev);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_draws_philox_f64(int device, unsigned long long seed, unsigned long long offset, long long n, double scale,
                        double* out) {
    if (n < 0) return fail(RC_EINVAL, "n must be non-negative");
    if (n == 0) return RC_OK;
    if (!out) return fail(RC_EINVAL, "NULL output pointer");
    if (int rc = device_in_range(device)) return rc;
    std::lock_guard<std::mutex> lk(g_ctx[device].mu);
    DeviceCtx* ctx = nullptr;
    if (int rc = get_ctx(device, &ctx)) return rc;
    const bool dev_out = is_device_ptr(out);
    double* d_out = out;
    if (!dev_out) {
        if (int rc = ensure_ws(ctx, (size_t)n * sizeof(double))) return rc;
        d_out = (double*)ctx->ws;
    }
    if (int rc = rc_draws_philox_f64_async(device, ctx->stream, seed, offset, n, scale, d_out)) return rc;
    if (!dev_out) RC_HIP_CHECK(hipMemcpyAsync(out, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RC_OK;
}

int rc_mc_fidelity_sharded_f64(int ndev, const int* devices, int kernel, int N, int in, int out, const double* h0_diag,
                               const double* h0_offdiag, int ring, const double* controllers, const double* draws,
                               long long C, long long K, double* fid_out) {
    if (C > 0 && K > 0 && (!draws || !fid_out)) return fail(RC_EINVAL, "NULL array pointer");
    ShardJob j{};
    j.kernel = kernel; j.N = N; j.in = in; j.out = out; j.ring = ring;
    j.h0d = h0_diag; j.h0o = h0_offdiag; j.ctrl = controllers; j.draws = draws;
    j.C = C; j.K = K; j.fid_out = fid_out;
    return run_sharded(ndev, devices, j);
}

int rc_mc_metrics_sharded_f64(int ndev, const int* devices, int kernel, int N, int in, int out, const double* h0_diag,
                              const double* h0_offdiag, int ring, const double* controllers, const double* draws,
                              unsigned long long philox_seed, unsigned long long philox_offset, double sigma,
                              long long C, long long K, const double* q_thresholds, int nq, double dkw_eps,
                              double* rim1, double* std_, double* minf, double* q, double* fid_out) {
    ShardJob j{};
    j.kernel = kernel; j.N = N; j.in = in; j.out = out; j.ring = ring;
    j.h0d = h0_diag; j.h0o = h0_offdiag; j.ctrl = controllers; j.draws = draws;
    j.seed = philox_seed; j.offset = philox_offset; j.sigma = sigma;
    j.C = C; j.K = K; j.thr = q_thresholds; j.nq = nq; j.eps = dkw_eps;
    j.rim1 = rim1; j.stdv = std_; j.minf = minf; j.q = q; j.fid_out = fid_out;
    return run_sharded(ndev, devices, j);
}

int rc_comm_init(int ndev, const int* devices, rc_comm** comm_out) {
    if (!comm_out) return fail(RC_EINVAL, "NULL communicator pointer");
    *comm_out = nullptr;
    if (ndev < 1 || ndev > kMaxDevices) return fail(RC_EINVAL, "ndev must be in [1, 64]");
    std::vector<int> devs(ndev);
    for (int r = 0; r < ndev; ++r) {
        devs[r] = devices ? devices[r] : r;
        if (int rc = device_in_range(devs[r])) return rc;
        for (int r2 = 0; r2 < r; ++r2)
            if (devs[r2] == devs[r]) return fail(RC_EINVAL, "a device is listed twice (RCCL needs one rank per device)");
    }
    const RcclApi& a = rccl_api();
    if (!a.ok) return fail(RC_ENOSUP, "librccl.so not found (looked in the process, then for librccl.so.1 / librccl.so)");
    auto* c = new rc_comm;
    c->ndev = ndev;
    c->devices = devs;
    c->comms.assign(ndev, nullptr);
    if (int e = a.init_all(c->comms.data(), ndev, devs.data())) {
        delete c;
        return rccl_fail(e, "ncclCommInitAll");
    }
    *comm_out = c;
    return RC_OK;
}

int rc_comm_size(const rc_comm* comm) { return comm ? comm->ndev : 0; }

int rc_comm_destroy(rc_comm* comm) {
    if (!comm) return RC_OK;
    const RcclApi& a = rccl_api();
    int bad = 0;
    for (void* c : comm->comms)
        if (c && a.destroy) bad |= a.destroy(c);
    delete comm;
    return bad ? rccl_fail(bad, "ncclCommDestroy") : RC_OK;
}

int rc_mc_metrics_gathered_f64(rc_comm* comm, int kernel, int N, int in, int out, const double* h0_diag, const double* h0_offdiag,
                               int ring, const double* controllers, const double* draws, unsigned long long philox_seed,
                               unsigned long long philox_offset, double sigma, long long C, long long K,
                               const double* q_thresholds, int nq, double dkw_eps, double* const* table_dev,
                               double* const* fid_dev, double* table_host, double* fid_host) {
    if (!comm) return fail(RC_EINVAL, "NULL communicator");
    if (int rc = check_common(N, in, out, C, K)) return rc;
    if (nq < 0 || nq > kMaxQ) return fail(RC_EINVAL, "nq must be in [0, 8]");
    if (nq > 0 && !q_thresholds) return fail(RC_EINVAL, "q_thresholds is NULL");
    if (C == 0 || K == 0) return RC_OK;
    if (!controllers) return fail(RC_EINVAL, "NULL controllers pointer");
    const RcclApi& a = rccl_api();
    const int ndev = comm->ndev;
    // (a fidelity gather needs a receive buffer on EVERY rank of the collective: asked for before any work is enqueued)
    if (fid_host && !fid_dev && ndev > 1)
        return fail(RC_EINVAL, "fid_host with more than one device needs fid_dev (a receive buffer on every device)");
    const long long Cmax = (C + ndev - 1) / ndev, G = 3LL * N;
    const int NR = 9 + 3 * nq;
    // per device: send buffers (own share) + - unless the caller provides them - the gathered buffers
    struct Bufs {
        double *ctrl = nullptr, *draw = nullptr, *fid = nullptr, *tab = nullptr, *tab_all = nullptr, *fid_all = nullptr;
        bool own_tab = false, own_fid = false;
    };
    std::vector<Bufs> b(ndev);
    std::vector<GatherJob> jobs(ndev);
    const bool want_fid = fid_dev || fid_host;
    const long long base = C / ndev, extra = C % ndev;
    long long start = 0;
    int rc_all = RC_OK;
    // (plain hipMalloc / hipFree, not the stream-ordered pool: the call is blocking anyway, and a process that had used
    // hipMallocAsync here and then created another stream - torch.cuda.Stream() in a later test - ended in glibc's "double free
    // or corruption" inside the runtime's exit handlers on ROCm 7.2)
    auto cleanup = [&]() {                          // the call is blocking: every stream is drained, then the buffers go
        for (int r = 0; r < ndev; ++r) {
            if (hipSetDevice(comm->devices[r]) != hipSuccess) continue;
            (void)hipStreamSynchronize(g_ctx[comm->devices[r]].stream);
            for (double* p : {b[r].ctrl, b[r].draw, b[r].fid, b[r].tab}) if (p) (void)hipFree(p);
            if (b[r].own_tab && b[r].tab_all) (void)hipFree(b[r].tab_all);
            if (b[r].own_fid && b[r].fid_all) (void)hipFree(b[r].fid_all);
        }
    };
    for (int r = 0; r < ndev && rc_all == RC_OK; ++r) {
        const int dev = comm->devices[r];
        DeviceCtx* ctx = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_ctx[dev].mu);
            if (int rc = get_ctx(dev, &ctx)) { rc_all = rc; break; }
        }
        hipStream_t st = ctx->stream;
        GatherJob& j = jobs[r];
        j.device = dev; j.kernel = kernel; j.N = N; j.in = in; j.out = out; j.ring = ring;
        j.h0d = h0_diag; j.h0o = h0_offdiag; j.ctrl = controllers; j.draws = draws;
        j.seed = philox_seed; j.offset = philox_offset; j.sigma = sigma;
        j.C = C; j.K = K; j.Cmax = Cmax; j.thr = q_thresholds; j.nq = nq; j.eps = dkw_eps;
        j.c0 = start;
        start += base + (r < extra ? 1 : 0);
        j.c1 = start;
        const bool fused = !draws && !ring && rc_philox_fused_pays(N, in, out) == 1 &&
                           (kernel == RC_KERNEL_AUTO || kernel == RC_KERNEL_TRIDIAG_ADJ);
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess) e = hipMalloc((void**)&b[r].ctrl, (size_t)Cmax * (N + 1) * sizeof(double));
        if (e == hipSuccess && !fused) e = hipMalloc((void**)&b[r].draw, (size_t)Cmax * K * G * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&b[r].fid, (size_t)Cmax * K * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&b[r].tab, (size_t)NR * Cmax * sizeof(double));
        b[r].tab_all = table_dev ? table_dev[r] : nullptr;
        if (e == hipSuccess && !b[r].tab_all) {
            e = hipMalloc((void**)&b[r].tab_all, (size_t)ndev * NR * Cmax * sizeof(double));
            b[r].own_tab = true;
        }
        b[r].fid_all = fid_dev ? fid_dev[r] : nullptr;
        if (e == hipSuccess && want_fid && !b[r].fid_all && (fid_dev || r == 0)) {      // a host copy only needs device 0's
            e = hipMalloc((void**)&b[r].fid_all, (size_t)ndev * Cmax * K * sizeof(double));
            b[r].own_fid = true;
        }
        if (e != hipSuccess) rc_all = fail(RC_EHIP, std::string("rc_mc_metrics_gathered_f64 (allocation): ") + hipGetErrorString(e));
        j.d_ctrl = b[r].ctrl; j.d_draw = b[r].draw; j.d_fid = b[r].fid; j.d_tab = b[r].tab;
    }
    if (rc_all) {
        cleanup();
        return rc_all;
    }
    // every device computes and reduces its share (one host thread per device, everything enqueued on the device's stream) ...
    {
        std::vector<std::thread> th;
        for (int r = 1; r < ndev; ++r) th.emplace_back(run_gather_share_locked, &jobs[r]);
        run_gather_share_locked(&jobs[0]);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < ndev; ++r)
        if (jobs[r].rc) {
            rc_all = fail(jobs[r].rc, "device " + std::to_string(jobs[r].device) + ": " + jobs[r].err);
            break;
        }
    // ... and the exchange step: ONE group of all-gathers, each on its device's stream behind that device's kernels.  A fidelity
    // gather needs a receive buffer on EVERY rank of the collective: without fid_dev only device 0 has one, so the slabs are
    // gathered only when every device has (fid_dev given), otherwise device 0's own slab + peers' come through table only.
    const bool gather_fid = want_fid && (fid_dev != nullptr || ndev == 1);
    if (rc_all == RC_OK) {
        int e = a.group_start();
        for (int r = 0; r < ndev && !e; ++r) {
            hipStream_t st = g_ctx[comm->devices[r]].stream;
            e = a.allgather(b[r].tab, b[r].tab_all, (size_t)NR * Cmax, kNcclFloat64, comm->comms[r], st);
            if (!e && gather_fid) e = a.allgather(b[r].fid, b[r].fid_all, (size_t)Cmax * K, kNcclFloat64, comm->comms[r], st);
        }
        const int e2 = a.group_end();
        if (e || e2) rc_all = rccl_fail(e ? e : e2, "ncclAllGather");
    }
    // host copies from device 0's gathered buffers, padding dropped: table_host [NR][C], fid_host [C][K]
    if (rc_all == RC_OK && (table_host || (fid_host && gather_fid))) {
        const int dev0 = comm->devices[0];
        hipStream_t st = g_ctx[dev0].stream;
        hipError_t e = hipSetDevice(dev0);
        long long c_at = 0;
        for (int r = 0; r < ndev && e == hipSuccess; ++r) {
            const long long cl = jobs[r].c1 - jobs[r].c0;
            if (cl > 0 && table_host)
                e = hipMemcpy2DAsync(table_host + c_at, (size_t)C * sizeof(double), b[0].tab_all + (size_t)r * NR * Cmax,
                                     (size_t)Cmax * sizeof(double), (size_t)cl * sizeof(double), NR, hipMemcpyDeviceToHost, st);
            if (cl > 0 && fid_host && gather_fid && e == hipSuccess)
                e = hipMemcpyAsync(fid_host + c_at * K, b[0].fid_all + (size_t)r * Cmax * K, (size_t)cl * K * sizeof(double),
                                   hipMemcpyDeviceToHost, st);
            c_at += cl;
        }
        if (e != hipSuccess) rc_all = fail(RC_EHIP, std::string("rc_mc_metrics_gathered_f64 (copy to the host): ") + hipGetErrorString(e));
    }
    cleanup();                                        // drains every stream first: results (device and host) are complete on return
    return rc_all;
}

int rc_draws_legacy_f64(int device, void* stream, rc_mt19937_state* state, long long n_periods, long long period,
                        long long skip, const double* scales, double* out_dev) {
    if (!state) return fail(RC_EINVAL, "NULL generator state");
    if (state->pos < 0 || state->pos > 624) return fail(RC_EINVAL, "generator state: pos must be in [0, 624]");
    if (n_periods < 0 || period < 0 || skip < 0 || skip > period) return fail(RC_EINVAL, "need n_periods, period >= 0 and 0 <= skip <= period");
    if (n_periods == 0 || period == 0) return RC_OK;
    if (!scales) return fail(RC_EINVAL, "NULL scales pointer");
    if (skip < period && !out_dev) return fail(RC_EINVAL, "NULL output pointer");
    if (int rc = device_in_range(device)) return rc;
    RC_HIP_CHECK(hipSetDevice(device));
    return legacy_normal_stream((hipStream_t)stream, state, n_periods, period, skip, scales, out_dev);
}

int rc_directional_draws_legacy_dev(int device, void* stream, rc_mt19937_state* state, long long n, int ndir, double sigma,
                                    int* idx_dev, double* ab_dev) {
    if (!state) return fail(RC_EINVAL, "NULL generator state");
    if (state->pos < 0 || state->pos > 624) return fail(RC_EINVAL, "generator state: pos must be in [0, 624]");
    if (n < 0 || ndir < 1) return fail(RC_EINVAL, "need n >= 0 and ndir >= 1");
    if (n == 0) return RC_OK;
    if (!idx_dev || !ab_dev) return fail(RC_EINVAL, "NULL output pointer");
    if (int rc = device_in_range(device)) return rc;
    RC_HIP_CHECK(hipSetDevice(device));
    const unsigned int rng = (unsigned int)ndir - 1u;
    unsigned int mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    for (long long done = 0; done < n; done += kDirChunkSamples) {
        const long long cnt = (n - done < kDirChunkSamples) ? (n - done) : kDirChunkSamples;
        if (int rc = directional_chunk((hipStream_t)stream, state, cnt, rng, mask, sigma, idx_dev + done, ab_dev + 2 * done))
            return rc;
    }
    return RC_OK;
}

}  // extern "C"
