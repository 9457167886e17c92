// Host-side (CPU, no GPU) emulation of NumPy's legacy RandomState for RNG consumption patterns that are inherently
// sequential AND interleave different distributions, so that neither one vectorised NumPy call nor the device-side
// stream (rc_draws_legacy_f64: normals only) can reproduce them.  The one user on the path:
// `directional_perturbation.perturbation()` (noise_model.py:183-189) draws, PER SAMPLE,
//     idx  = np.random.randint(low=0, high=len(directions))      masked rejection on 32-bit outputs
//     nval = self.rng(size=2)                                     two legacy normals (polar method, cached second value)
// In Python that is two interpreter-level calls per sample (~3 us); here it is ~30 ns per sample, bit-identical to NumPy
// in the indices, the normals (same libm) and the generator state afterwards (tests/test_host_logic.py).
#include <hip/hip_runtime.h>          // hipcc compiles every input of this library as HIP: the shared header's qualifiers

#include <math.h>
#include <stdint.h>

#pragma STDC FP_CONTRACT OFF          // NumPy's C code has no fused multiply-adds

#include "../../include/robchar_hip.h"
#include "legacy_rng_core.h"

namespace {

struct HostMt {
    rc_mt19937_state* s;
    // numpy/random/src/mt19937/mt19937.c: mt19937_gen - regenerate the whole block in place when it is used up
    uint32_t next_raw() {
        if (s->pos >= rcl::kMtN) {
            uint32_t* k = s->key;
            for (int i = 0; i < rcl::kMtN; ++i)
                k[i] = rcl::mt_next_word(k[i], k[(i + 1) % rcl::kMtN], k[(i + rcl::kMtM) % rcl::kMtN]);
            s->pos = 0;
        }
        return s->key[s->pos++];
    }
    // legacy_gauss (numpy/random/src/legacy/legacy-distributions.c)
    double gauss() {
        if (s->has_gauss) {
            const double t = s->gauss;
            s->has_gauss = 0;
            s->gauss = 0.0;
            return t;
        }
        double x1, x2, r2;
        for (;;) {
            const uint32_t a = next_raw(), b = next_raw(), c = next_raw(), d = next_raw();
            if (rcl::polar_attempt(a, b, c, d, x1, x2, r2)) break;
        }
        const double f = sqrt(-2.0 * log(r2) / r2);
        s->gauss = f * x1;
        s->has_gauss = 1;
        return f * x2;
    }
    // RandomState.randint(0, n) for n - 1 <= 0xffffffff: masked rejection on 32-bit outputs (_bounded_integers, use_masked)
    uint32_t below(uint32_t n) {
        const uint32_t rng = n - 1;
        if (rng == 0) return 0;                       // no draw consumed
        uint32_t mask = rng;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        uint32_t v;
        do {
            v = rcl::mt_temper(next_raw()) & mask;
        } while (v > rng);
        return v;
    }
};

}  // namespace

extern "C" int rc_directional_draws_legacy(rc_mt19937_state* state, long long n, int ndir, double sigma, int* idx_out,
                                           double* ab_out) {
    if (!state || state->pos < 0 || state->pos > rcl::kMtN || n < 0 || ndir < 1) return RC_EINVAL;
    if (n > 0 && (!idx_out || !ab_out)) return RC_EINVAL;
    HostMt g{state};
    for (long long i = 0; i < n; ++i) {
        idx_out[i] = (int)g.below((uint32_t)ndir);
        ab_out[2 * i] = 0.0 + sigma * g.gauss();      // legacy_normal: loc + scale * gauss
        ab_out[2 * i + 1] = 0.0 + sigma * g.gauss();
    }
    return RC_OK;
}

// (ABI 6) Is the `log` restated in legacy_rng_core.h (glibc >= 2.28's table-driven routine in its FMA build, constants read
// from the libm of the image the library was built in) the `log` of THIS host's C library - the one NumPy calls for its legacy
// normals?  Checked once, on the host, on 2^17 arguments over the polar method's range (uniform, near 1, tiny): 1 = every
// result identical bit for bit, so the normals the device continues NumPy's stream with are NumPy's own; 0 = another libm (or a
// CPU on which glibc selects a non-FMA build): the device stream then still hands back the exact generator STATE, but its
// normals may differ from NumPy's in the last bits, and the Python layer draws on the host instead (bit-identical by
// construction).
extern "C" int rc_legacy_log_is_host_exact(void) {
    static const int exact = [] {
        static const double tab[256] = {RC_GLIBC_LOG_TAB_VALUES};
        uint64_t s = 88172645463325252ull;
        for (int j = 0; j < (1 << 17); ++j) {
            s ^= s << 13;
            s ^= s >> 7;
            s ^= s << 17;
            double u = (double)(s >> 11) * 0x1.0p-53;
            if (j % 4 == 1) u = 0.9 + 0.1 * u;
            if (j % 4 == 2) u = u * u * u * 1e-9;
            if (!(u > 0.0) || u >= 1.0) continue;
            volatile double arg = u;                     // (the library call itself, not a folded constant)
            const double a = rcl::log_glibc_fma(u, tab), b = log(arg);
            if (memcmp(&a, &b, sizeof a) != 0) return 0;
        }
        return 1;
    }();
    return exact;
}

