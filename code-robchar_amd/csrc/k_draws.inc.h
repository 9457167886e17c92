// Draw generators: philox_normal_kernel (counter-based, NOT the reference's RNG) and NumPy's legacy MT19937 /
// polar Box-Muller stream on the device (jump-ahead sub-streams, attempt count / scan / emit).
//
// Part of ONE translation unit: this file is #included by robchar_hip.hip INSIDE its anonymous namespace (after the
// shared parameter structs); it is not a stand-alone header.
// ------------------------------------------------------------------------------------------------
// counter-based Gaussian draws (explicitly NOT the reference's RNG: for sample spaces too large to draw on the
// host, e.g. BASELINE config 4 = 2.1e9 draws).  Philox4x32-10 keyed by `seed`; element e of the stream comes from
// counter (e >> 1): two 53-bit uniforms -> Box-Muller pair, element parity picks cos / sin.  Any element can be
// regenerated independently (oracle/philox_host.py does, for the parity tests).
// ------------------------------------------------------------------------------------------------
// a ^ b ^ c in ONE instruction (v_bitop3_b32, truth table 0x96; gfx950): the compiler leaves the two xors of a Philox round
// as two v_xor_b32 (round 4: 167 -> 152 VALU instructions per Box-Muller pair)
__device__ __forceinline__ unsigned int xor3(unsigned int a, unsigned int b, unsigned int c) {
#if __has_builtin(__builtin_amdgcn_bitop3_b32)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

__device__ __forceinline__ void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                              unsigned int k0, unsigned int k1, unsigned int (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned int n0 = xor3((unsigned int)(p1 >> 32), c1, k0);
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n2 = xor3((unsigned int)(p0 >> 32), c3, k1);
        const unsigned int n3 = (unsigned int)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// (ln c_k, 1 / c_k), c_k = 1/2 + (k + 1) / 256, k = 0..127: ln f = ln c_k + log1p((f - c_k) / c_k) for f in [1/2, 1)
#define RC_LN_TABLE_VALUES \
    -0.6853650401178903, 1.9844961240310077, -0.6776429940239801, 1.9692307692307693, \
    -0.6699801212784109, 1.9541984732824427, -0.6623755218931916, 1.9393939393939394, \
    -0.6548283162578087, 1.9248120300751879, -0.6473376445286511, 1.9104477611940298, \
    -0.639902666041133, 1.8962962962962964, -0.6325225587435105, 1.8823529411764706, \
    -0.6251965186514375, 1.8686131386861313, -0.6179237593223578, 1.855072463768116, \
    -0.6107035113488707, 1.841726618705036, -0.6035350218702582, 1.8285714285714285, \
    -0.5964175541013942, 1.8156028368794326, -0.5893503868783018, 1.8028169014084507, \
    -0.5823328142196552, 1.7902097902097902, -0.5753641449035618, 1.7777777777777777, \
    -0.5684437020589881, 1.7655172413793103, -0.561570822771226, 1.7534246575342465, \
    -0.5547448577008262, 1.7414965986394557, -0.5479651707154474, 1.7297297297297298, \
    -0.5412311385341033, 1.7181208053691275, -0.5345421503833068, 1.7066666666666668, \
    -0.5278976076646381, 1.695364238410596, -0.5212969236332861, 1.6842105263157894, \
    -0.514739523087127, 1.673202614379085, -0.5082248420659333, 1.6623376623376624, \
    -0.5017523275603158, 1.6516129032258065, -0.4953214372300254, 1.641025641025641, \
    -0.4889316391312544, 1.6305732484076434, -0.48258241145259567, 1.620253164556962, \
    -0.47627324225933093, 1.610062893081761, -0.4700036292457356, 1.6, \
    -0.4637730794950995, 1.5900621118012421, -0.4575811092471784, 1.5802469135802468, \
    -0.4514272436728001, 1.5705521472392638, -0.44531101665536404, 1.5609756097560976, \
    -0.4392319705789819, 1.5515151515151515, -0.43318965612301924, 1.5421686746987953, \
    -0.42718363206280735, 1.532934131736527, -0.42121346507630353, 1.5238095238095237, \
    -0.415278729556489, 1.514792899408284, -0.4093790074293007, 1.5058823529411764, \
    -0.40351388797690263, 1.4970760233918128, -0.39768296766610944, 1.4883720930232558, \
    -0.39188584998178355, 1.4797687861271676, -0.38612214526503347, 1.471264367816092, \
    -0.38039147055604844, 1.4628571428571429, -0.3746934494414107, 1.4545454545454546, \
    -0.36902771190573336, 1.4463276836158192, -0.3633938941874773, 1.4382022471910112, \
    -0.3577916386388075, 1.4301675977653632, -0.3522205935893521, 1.4222222222222223, \
    -0.3466804132137367, 1.4143646408839778, -0.34117075740276714, 1.4065934065934067, \
    -0.33569129163814154, 1.3989071038251366, -0.33024168687057687, 1.391304347826087, \
    -0.32482161940123766, 1.3837837837837839, -0.3194307707663612, 1.3763440860215055, \
    -0.31406882762497584, 1.3689839572192513, -0.3087354816496133, 1.3617021276595744, \
    -0.3034304294199201, 1.3544973544973544, -0.29815337231907635, 1.3473684210526315, \
    -0.2929040164329326, 1.3403141361256545, -0.2876820724517809, 1.3333333333333333, \
    -0.2824872555746769, 1.3264248704663213, -0.27731928541623435, 1.3195876288659794, \
    -0.27217788591581565, 1.3128205128205128, -0.26706278524904525, 1.3061224489795917, \
    -0.26197371574157396, 1.299492385786802, -0.2569104137850272, 1.292929292929293, \
    -0.2518726197550701, 1.2864321608040201, -0.24686007793152578, 1.28, \
    -0.24187253642048673, 1.2736318407960199, -0.2369097470783577, 1.2673267326732673, \
    -0.23197146543777514, 1.2610837438423645, -0.22705745063534608, 1.2549019607843137, \
    -0.2221674653411543, 1.248780487804878, -0.2173012756899814, 1.2427184466019416, \
    -0.2124586512141934, 1.2367149758454106, -0.2076393647782445, 1.2307692307692308, \
    -0.20284319251475147, 1.2248803827751196, -0.1980699137620938, 1.2190476190476192, \
    -0.19331931100349597, 1.2132701421800949, -0.18859116980755003, 1.2075471698113207, \
    -0.18388527877013736, 1.2018779342723005, -0.179201429457711, 1.1962616822429906, \
    -0.17453941635189968, 1.1906976744186046, -0.16989903679539747, 1.1851851851851851, \
    -0.16528009093910292, 1.1797235023041475, -0.16068238169047347, 1.1743119266055047, \
    -0.15610571466306167, 1.1689497716894977, -0.15154989812720093, 1.1636363636363636, \
    -0.14701474296180966, 1.158371040723982, -0.14250006260728304, 1.1531531531531531, \
    -0.13800567301944372, 1.147982062780269, -0.13353139262452263, 1.1428571428571428, \
    -0.12907704227514236, 1.1377777777777778, -0.1246424452072766, 1.1327433628318584, \
    -0.1202274269981598, 1.1277533039647578, -0.1158318155251217, 1.1228070175438596, \
    -0.11145544092532282, 1.1179039301310043, -0.1070981355563671, 1.1130434782608696, \
    -0.10275973395776894, 1.1082251082251082, -0.09844007281325252, 1.103448275862069, \
    -0.09413899091386191, 1.0987124463519313, -0.08985632912186105, 1.0940170940170941, \
    -0.08559193033540351, 1.0893617021276596, -0.0813456394539524, 1.0847457627118644, \
    -0.07711730334443129, 1.080168776371308, -0.07290677080808779, 1.0756302521008403, \
    -0.06871389254805181, 1.0711297071129706, -0.06453852113757118, 1.0666666666666667, \
    -0.06038051098890748, 1.062240663900415, -0.05623971832287608, 1.0578512396694215, \
    -0.05211600113901402, 1.0534979423868314, -0.048009219186360606, 1.0491803278688525, \
    -0.04391923393483549, 1.0448979591836736, -0.039845908547199674, 1.0406504065040652, \
    -0.03578910785158528, 1.0364372469635628, -0.0317486983145803, 1.032258064516129, \
    -0.027724548014854862, 1.0281124497991967, -0.023716526617316044, 1.024, \
    -0.01972450534777859, 1.0199203187250996, -0.015748356968139168, 1.0158730158730158, \
    -0.01178795575204224, 1.0118577075098814, -0.007843177461025893, 1.0078740157480315, \
    -0.003913899321136329, 1.003921568627451, 0.0, 1.0
__device__ const double g_ln_table[256] = {RC_LN_TABLE_VALUES};
// glibc's (invc, logc) table: the LEGACY stream's normals use the C library's log operation for operation
// (legacy_rng_core.h: log_glibc_fma) so that they equal NumPy's bit for bit; the Philox stream keeps ln_table below.
__device__ const double g_glibc_log_table[256] = {RC_GLIBC_LOG_TAB_VALUES};

// ln u for u in (0, 1]: u = 2^e f, f in [1/2, 1); ln f = ln c_k + log1p((f - c_k) / c_k) with the 128-entry (ln c, 1/c)
// table above (in LDS) and a degree-7 series on |r| <= 1/128.  A few ulp from libm.
__device__ __forceinline__ double ln_table(double u, const double* lntab) {
    const double f = __builtin_amdgcn_frexp_mant(u);
    const int ex = __builtin_amdgcn_frexp_exp(u);
    const int k = (int)((__double2hiint(f) >> 13) & 127);            // top 7 fraction bits
    const double ck = 0.5 + (double)(k + 1) * 0x1.0p-8;
    const double r = (f - ck) * lntab[2 * k + 1];                    // in [-1/128, 0)
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(r, p, 0.2);
    p = fma(r, p, -0.25);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    p = fma(r * r, p, r);                                            // log1p(r)
    return fma((double)ex, 6.93147180559945286227e-01, lntab[2 * k] + p);
}

// Box-Muller pair of counter `ctr`: elements 2 ctr = amp * cs and 2 ctr + 1 = amp * sn of stream `seed` (amp = scale * radius).
// Shared by philox_normal_kernel and by the fidelity kernel that generates its own draws (k_fidelity_philox.inc.h): the two
// produce the same bits.
__device__ __forceinline__ void philox_pair(unsigned long long seed, unsigned long long ctr, double scale, const double* lntab,
                                            const double* sctab, double& amp, double& cs, double& sn) {
    unsigned int w[4];
    philox4x32_10((unsigned int)ctr, (unsigned int)(ctr >> 32), 0u, 0u, (unsigned int)seed, (unsigned int)(seed >> 32), w);
    const unsigned long long a = (((unsigned long long)w[1] << 32) | w[0]) >> 11;
    const unsigned long long b = (((unsigned long long)w[3] << 32) | w[2]) >> 11;
    const double u1 = ((double)a + 0.5) * 0x1.0p-53;               // (0, 1)
    const double u2 = ((double)b + 0.5) * 0x1.0p-53;
    const double lnu = ln_table(u1, lntab);
    double rad, rinv;
    rc::sqrt_rsqrt(-2.0 * lnu, rad, rinv);
    rc::sincos_table(64.0 * u2, sctab, sn, cs);
    amp = scale * rad;
}

// One thread per Box-Muller PAIR (counter): one Philox call, one log / sqrt, one sin/cos -> elements 2 ctr (cos) and
// 2 ctr + 1 (sin).  The three library calls are replaced by table-driven routines (LDS reads are cheap next to fp64
// VALU work): ln u through a 128-entry (ln c, 1/c) table + a degree-7 log1p series on |r| <= 1/128
// (c_127 = 1 exactly, so u -> 1 keeps full relative accuracy), sqrt through the v_rsq_f64 seed + one third-order
// step, sin/cos(2 pi u) through rc::sincos_table (64 u is exact).  Each agrees with libm to a few ulp
// (tests: |device - numpy| < 1e-15 on 0.05-scaled draws).  3.3x the throughput of the per-element libm version.
__global__ __launch_bounds__(256) void philox_normal_kernel(unsigned long long seed, unsigned long long offset,
                                                            long long n, double scale, double* out) {
    __shared__ __attribute__((aligned(16))) double sctab[128];
    __shared__ __attribute__((aligned(16))) double lntab[256];
    if (threadIdx.x < 64)
        reinterpret_cast<double2*>(sctab)[threadIdx.x] = reinterpret_cast<const double2*>(g_sincos_table)[threadIdx.x];
    if (threadIdx.x < 128)
        reinterpret_cast<double2*>(lntab)[threadIdx.x] = reinterpret_cast<const double2*>(g_ln_table)[threadIdx.x];
    __syncthreads();
    const unsigned long long first = offset >> 1;                        // first counter touched
    const unsigned long long last = (offset + (unsigned long long)n - 1) >> 1;
    const long long npairs = (long long)(last - first + 1);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < npairs; t += (long long)gridDim.x * 256) {
        const unsigned long long ctr = first + (unsigned long long)t;
        double amp, sn, cs;
        philox_pair(seed, ctr, scale, lntab, sctab, amp, cs, sn);
        const unsigned long long e0 = ctr << 1;
#ifdef RC_EXPERIMENT_PHILOX_NOSTORE
        // TIMING EXPERIMENT ONLY (scripts/build_variant.sh): the generator's arithmetic without its 16 bytes of HBM write per
        // pair - what generating the draws INSIDE the fidelity kernel would cost
        if (amp * cs == 1234.5678 && amp * sn == 8765.4321) out[e0 - offset] = amp;
#else
        if (e0 >= offset) out[e0 - offset] = amp * cs;
        if (e0 + 1 < offset + (unsigned long long)n && e0 + 1 >= offset) out[e0 + 1 - offset] = amp * sn;
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// NumPy's legacy normal stream on the device (legacy_rng_core.h): the reference's RNG without the host
// ------------------------------------------------------------------------------------------------
// Stage 1 - raw MT19937 words.  raw[0 .. 624) holds a state block (the host's key, or the carry block of the previous
// segment); this kernel appends the following blocks.  The recurrence x[i] = next(x[i-624], x[i-623], x[i-227]) makes
// 227 consecutive words independent of each other and dependent on the chunk before: ONE wave walks the chunks (4 words
// per lane), the last 2048 words in an LDS ring, wave-level fences between chunks.  Sequential by nature, ~0.3 ns per
// word - an order of magnitude faster than NumPy's scalar generator on the host, and the words are born in HBM.
// LDS-only ordering point of ONE wave: the LDS operations of a wave execute in order, so draining the LDS counter is
// all that is needed between a chunk's writes and the next chunk's reads.  (A full release/acquire fence would also
// wait for the chunk's GLOBAL stores - hundreds of cycles per chunk on a purely sequential kernel.)
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Sub-stream p (one wave per workgroup, P workgroups in parallel) starts from the 624-word window seeds[p] - the
// generator's window at global word p * kMtJumpWords (seeds[0] = the caller's block) - and writes the kMtJumpWords words
// that FOLLOW its window, raw[624 + p B ... 624 + (p + 1) B): the concatenation over p is the sequential stream.
// `raw` must hold `total` words (a multiple of 624); the last sub-stream stops there.
// x[i+624] from x[i], x[i+1], x[i+397] with the three-input logic of gfx950: (xi & hi) | (xi1 & lo) and the final double xor
// are one v_bitop3_b32 each (rcl::mt_next_word stays the shared, portable definition: host unit test)
__device__ __forceinline__ unsigned int mt_next_word_dev(unsigned int xi, unsigned int xi1, unsigned int xim) {
#if __has_builtin(__builtin_amdgcn_bitop3_b32)
    const unsigned int y = __builtin_amdgcn_bitop3_b32(xi, xi1, 0x80000000u, 0xe4);      // c ? a : b, bit by bit (a = 0xf0, b = 0xcc, c = 0xaa)
    const unsigned int mag = (unsigned int)(-(int)(xi1 & 1u)) & 0x9908b0dfu;
    return __builtin_amdgcn_bitop3_b32(xim, y >> 1, mag, 0x96);                          // a ^ b ^ c
#else
    return rcl::mt_next_word(xi, xi1, xim);
#endif
}

constexpr int kMtWinChunks = 32;                                   // chunks generated before the window slides back
constexpr int kMtWinWords = rcl::kMtN + kMtWinChunks * rcl::kMtChunk + 64;       // (+64: the partial fourth row of a chunk)

__global__ __launch_bounds__(64) void mt19937_raw_kernel(const unsigned int* seeds, unsigned int* raw, long long total) {
    // Round 4: a LINEAR window instead of a ring.  The last 624 words sit at win[0 .. 624), the next kMtWinChunks chunks are
    // appended behind them, then the window's last 624 words are copied back to the front.  Every LDS access of a chunk is the
    // base address `w` (first word of the chunk - 624, + lane) plus a compile-time offset - no index arithmetic, no masking
    // (the ring cost 36 of the ~95 instructions a chunk took).
    __shared__ unsigned int win[kMtWinWords];
    const int lane = threadIdx.x;
    const long long p = blockIdx.x;
    const unsigned int* seed = seeds + p * rcl::kMtN;
    for (int i = lane; i < rcl::kMtN; i += 64) {
        win[i] = seed[i];
        if (p == 0) raw[i] = seed[i];
    }
    wave_lds_fence();
    // this sub-stream's share of the `total` words of the segment (the last one may be short)
    long long mine = total - rcl::kMtN - p * kMtJumpWords;
    mine = mine < 0 ? 0 : (mine > kMtJumpWords ? kMtJumpWords : mine);
    const long long full = mine / rcl::kMtChunk;
    const int rest = (int)(mine - full * rcl::kMtChunk);
    const bool tail = lane + 192 < rcl::kMtChunk;
    unsigned int* dst = raw + rcl::kMtN + p * kMtJumpWords + lane;
    // The chunk length (227) IS the recurrence's third distance - word k of a chunk needs word k of the chunk before - so that
    // term stays in the lane's own registers (`prev`); the two older terms (x[i - 624], x[i - 623]) lie 1.75 .. 2.75 chunks
    // back and are read for the NEXT chunk before this one is computed (LDS operations of one wave execute in program order):
    // the only chain from chunk to chunk is four register values.
    unsigned int prev[4], a[4], b[4];
    const unsigned int* w = win + lane;            // word k = 64 j + lane of the current chunk: new word at w[624 + 64 j]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        prev[j] = w[rcl::kMtN - rcl::kMtChunk + 64 * j];
        a[j] = w[64 * j];
        b[j] = w[64 * j + 1];
    }
    long long n = 0;
    while (n < full) {
        const long long left = full - n;
        const int todo = left < kMtWinChunks ? (int)left : kMtWinChunks;
        for (int q = 0; q < todo; ++q) {
            unsigned int an[4], bn[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {          // the NEXT chunk's old terms: written at least 170 words ago
                an[j] = w[rcl::kMtChunk + 64 * j];
                bn[j] = w[rcl::kMtChunk + 64 * j + 1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = mt_next_word_dev(a[j], b[j], prev[j]);
            unsigned int* wr = const_cast<unsigned int*>(w);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                wr[rcl::kMtN + 64 * j] = v[j];
                dst[64 * j] = v[j];                // fire and forget: nothing in this kernel reads `raw` back
            }
            if (tail) {
                wr[rcl::kMtN + 192] = v[3];
                dst[192] = v[3];
            }
            asm volatile("" ::: "memory");         // (compiler fence only: the window writes stay ahead of the next reads)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                prev[j] = v[j];
                a[j] = an[j];
                b[j] = bn[j];
            }
            w += rcl::kMtChunk;
            dst += rcl::kMtChunk;
        }
        n += todo;
        if (n < full) {
            // slide: the last 624 words (they end where the next chunk would start: w - lane + 624) go to the front; the
            // prefetched a / b were read from exactly those words, `prev` is in registers - nothing else refers to the window
            wave_lds_fence();
            const unsigned int* src = w - lane;                       // first of the last 624 words
            unsigned int t[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) t[i] = (lane + 64 * i < rcl::kMtN) ? src[lane + 64 * i] : 0u;
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < 10; ++i)
                if (lane + 64 * i < rcl::kMtN) win[lane + 64 * i] = t[i];
            wave_lds_fence();
            w = win + lane;
        }
    }
    if (rest) {                                    // last, partial chunk of the sub-stream
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < rest) dst[64 * j] = mt_next_word_dev(a[j], b[j], prev[j]);
    }
}

// Start windows of the sub-streams by jump-ahead (scripts/mt_jump_poly.py, mt19937_jump_poly.h): with g(x) = x^B mod
// phi(x), the window at distance B is  window_B[j] = XOR over the set coefficients i of g of  x[i + j].  One launch per
// round of jumps, WGS workgroups per jump: each regenerates the 19937 + 624 words behind the old window into its LDS
// (wave 0, 88 chunks, ~10 us) and produces 624 / WGS of the new words.
// XOR phase (late round 4): LANE = WORD, WAVE = an eighth of the ~9900 terms.  A term index is then uniform over the wave:
// the index table is read with SCALAR loads, two 16-bit indices per dword, eight dwords per batch, and a term costs one
// address add, one LDS read (consecutive lanes, consecutive words: conflict-free) and one xor.  (Until then a thread group
// per term slice, every lane fetching its own copy of the index through the vector memory path: ~3300 dependent
// global-load + LDS-read pairs per thread at four waves per CU - 95 of the 108 us of a launch.)  The LDS read rate of the
// CU is now the limit: 9900 terms x 64 lanes x 4 B at 128 B / clock = 8 us.
// Word 0 of a jumped window is exact only in its top bit - the only bit of it the recurrence uses; as an OUTPUT that word
// belongs to the sub-stream before.
// Sub-stream length B = 512 state blocks (2048 until late round 3): the raw-word kernel walks a sub-stream with ONE
// wave, so B sets its duration - against more jumps, whose total work grows with the number of sub-streams (a
// paper-scale algorithm: 136 sub-streams, 11 dependent launches).
constexpr int kJumpSeq = 19937 + rcl::kMtN;        // words of the stream a jump needs
constexpr int kJumpSeqPad = 20736;                 // >= kJumpSeq + 256 (whole chunks; lanes beyond a workgroup's words read on), LDS words
constexpr int kJumpThreads = 512;                 // two waves per SIMD
constexpr int kJumpWaves = kJumpThreads / 64;
constexpr int kJumpPartWords = kJumpWaves * 128;   // partial sums: [wave][up to 128 words]
constexpr int kJumpLdsWords = kJumpSeqPad + kJumpPartWords;
constexpr int kJumpMaxStride = 64;
// jump polynomials for distances B, 4 B, 16 B and 64 B: window q + m comes from window q, so the P start windows are
// built in O(log P)-ish rounds (up to m jumps of a round run side by side, blockIdx.y) instead of P - 1 jumps in sequence
__constant__ __attribute__((aligned(16))) const unsigned short g_mt_jump_idx1[kMtJumpTerms1] = {RC_MT_JUMP_IDX1_VALUES};
__constant__ __attribute__((aligned(16))) const unsigned short g_mt_jump_idx4[kMtJumpTerms4] = {RC_MT_JUMP_IDX4_VALUES};
__constant__ __attribute__((aligned(16))) const unsigned short g_mt_jump_idx16[kMtJumpTerms16] = {RC_MT_JUMP_IDX16_VALUES};
__constant__ __attribute__((aligned(16))) const unsigned short g_mt_jump_idx64[kMtJumpTerms64] = {RC_MT_JUMP_IDX64_VALUES};

// window (dst_first + y) = jump over `stride` windows from window (dst_first + y - stride), y = blockIdx.y
template <int WGS>
__global__ __launch_bounds__(kJumpThreads) void mt19937_jump_step_kernel(unsigned int* seeds, int dst_first, int stride) {
    constexpr int WORDS = rcl::kMtN / WGS;         // window words per workgroup
    constexpr int PER = (WORDS + 63) / 64;         // ... per lane
    static_assert(WORDS * WGS == rcl::kMtN && PER <= 2, "jump geometry");
    // the furthest word a lane reads - valid or not: (624 - WORDS) + 63 + 64 (PER - 1) + the largest term index - is inside the buffer
    static_assert((rcl::kMtN - WORDS) + 63 + 64 * (PER - 1) + 19936 < kJumpSeqPad, "jump geometry: LDS reads");
    extern __shared__ unsigned int xs[];           // kJumpLdsWords words
    unsigned int* part = xs + kJumpSeqPad;
    const int t = threadIdx.x;
    const int p = dst_first + (int)blockIdx.y;
    const unsigned short* jump_idx = stride == 1 ? g_mt_jump_idx1 : (stride == 4 ? g_mt_jump_idx4 : (stride == 16 ? g_mt_jump_idx16 : g_mt_jump_idx64));
    const int nterms = stride == 1 ? kMtJumpTerms1 : (stride == 4 ? kMtJumpTerms4 : (stride == 16 ? kMtJumpTerms16 : kMtJumpTerms64));
    const unsigned int* prev = seeds + (long long)(p - stride) * rcl::kMtN;
    for (int i = t; i < rcl::kMtN; i += kJumpThreads) xs[i] = prev[i];
    __syncthreads();
    if (t < 64) {                                  // wave 0: the stream after the old window
        // (same register-carried recurrence as mt19937_raw_kernel: the x[i - 227] term is the lane's own word of the chunk
        // before, the two older terms of the NEXT chunk are read before this one is computed)
        const bool tail = t + 192 < rcl::kMtChunk;
        int c = rcl::kMtN + t;
        unsigned int pv[4], a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pv[j] = xs[c + 64 * j - rcl::kMtChunk];
            a[j] = xs[c + 64 * j - rcl::kMtN];
            b[j] = xs[c + 64 * j - rcl::kMtN + 1];
        }
        for (; c - t < kJumpSeq; c += rcl::kMtChunk) {
            unsigned int an[4], bn[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                an[j] = xs[c + rcl::kMtChunk + 64 * j - rcl::kMtN];
                bn[j] = xs[c + rcl::kMtChunk + 64 * j - rcl::kMtN + 1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = mt_next_word_dev(a[j], b[j], pv[j]);
#pragma unroll
            for (int j = 0; j < 3; ++j) xs[c + 64 * j] = v[j];
            if (tail) xs[c + 192] = v[3];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pv[j] = v[j];
                a[j] = an[j];
                b[j] = bn[j];
            }
        }
        wave_lds_fence();
    }
    __syncthreads();
    // wave w: terms [2 k0, 2 k1) of the polynomial (pairs of 16-bit indices), every word of this workgroup (lane, lane + 64)
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned int* idx32 = reinterpret_cast<const unsigned int*>(jump_idx);
    const int npairs = nterms >> 1;
    const int per_wave = (npairs + kJumpWaves - 1) / kJumpWaves;
    int k = wave * per_wave;
    const int k1 = (k + per_wave < npairs) ? k + per_wave : npairs;
    const unsigned int* base = xs + blockIdx.x * WORDS + lane;
    unsigned int acc0 = 0, acc1 = 0;
    for (; k + 8 <= k1; k += 8) {                  // (the compiler makes this 16 pairs = 32 LDS reads in flight per wave;
        unsigned int pr[8];                        // the other wave of the SIMD covers the scalar loads' latency)
#pragma unroll
        for (int u = 0; u < 8; ++u) pr[u] = idx32[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned int* q0 = base + (pr[u] & 0xffffu);
            const unsigned int* q1 = base + (pr[u] >> 16);
            acc0 ^= q0[0] ^ q1[0];
            if (PER == 2) acc1 ^= q0[64] ^ q1[64];
        }
    }
    for (; k < k1; ++k) {
        const unsigned int pr = idx32[k];
        const unsigned int* q0 = base + (pr & 0xffffu);
        const unsigned int* q1 = base + (pr >> 16);
        acc0 ^= q0[0] ^ q1[0];
        if (PER == 2) acc1 ^= q0[64] ^ q1[64];
    }
    if ((nterms & 1) && wave == kJumpWaves - 1) {  // odd term count: the last index has no partner
        const unsigned int* q0 = base + jump_idx[nterms - 1];
        acc0 ^= q0[0];
        if (PER == 2) acc1 ^= q0[64];
    }
    part[wave * 128 + lane] = acc0;
    if (PER == 2) part[wave * 128 + 64 + lane] = acc1;
    __syncthreads();
    if (t < WORDS) {
        unsigned int acc = part[t];
#pragma unroll
        for (int w = 1; w < kJumpWaves; ++w) acc ^= part[w * 128 + t];
        seeds[(long long)p * rcl::kMtN + blockIdx.x * WORDS + t] = acc;
    }
}

// Stage 2 - polar-method attempts.  Attempt t reads raw words [w0 + 4t, w0 + 4t + 4) of the segment; a workgroup owns
// kLgAttempts consecutive attempts, thread x the attempts x, x + 256, ... of them (consecutive lanes read consecutive
// attempts, 16 B apart: until late round 4 a thread owned eight CONSECUTIVE attempts and every load instruction of a wave
// touched 64 cache lines).  Pass A counts the accepted attempts per workgroup, a one-block scan turns the counts into
// ranks, pass B recomputes the attempts and writes the two normals of accepted attempt number r (counted over the WHOLE
// stream) to stream elements e_shift + 2r (f x2) and e_shift + 2r + 1 (f x1), mapped through the period / skip / scale
// pattern of rcl::stream_slot - consecutive lanes hold consecutive ranks, so the stores are dense too.
constexpr int kLgThreads = 256;
constexpr int kLgPerThread = 8;
constexpr int kLgAttempts = kLgThreads * kLgPerThread;

struct LegacyParams {
    const unsigned int* raw;          // segment words; raw[0] is global word g0
    long long w_first;                // index INTO raw of the first word of attempt t_first
    long long t_first, t_count;       // attempts [t_first, t_first + t_count) are processed by this launch
    long long rank_base;              // accepted attempts before t_first
    long long pairs_needed;           // accepted attempts to emit in all
    long long e_shift, n_total;       // stream elements in front of the first generated one (0 | 1); total wanted
    long long period, skip;
    const double* scales;             // [n_periods] device
    double* out;
    unsigned long long* wg_counts;    // [nwg + 1]
    long long* last;                  // [0] attempt index of the last needed pair, [1..4] its raw words
};

__device__ __forceinline__ bool legacy_attempt(const LegacyParams& p, long long t, double& x1, double& x2, double& r2,
                                               unsigned int (&w)[4]) {
    const unsigned int* src = p.raw + p.w_first + 4 * (t - p.t_first);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = src[i];
    return rcl::polar_attempt(w[0], w[1], w[2], w[3], x1, x2, r2);
}

__global__ __launch_bounds__(kLgThreads) void legacy_count_kernel(const LegacyParams p) {
    __shared__ unsigned int wsum[kLgThreads / 64];
    const long long base = p.t_first + (long long)blockIdx.x * kLgAttempts + threadIdx.x;
    const long long t_end = p.t_first + p.t_count;
    unsigned int n = 0;
#pragma unroll
    for (int j = 0; j < kLgPerThread; ++j) {
        const long long t = base + (long long)j * kLgThreads;
        if (t < t_end) {
            double x1, x2, r2;
            unsigned int w[4];
            n += legacy_attempt(p, t, x1, x2, r2, w) ? 1u : 0u;
        }
    }
    n = wave_allsum(n);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) p.wg_counts[blockIdx.x] = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive prefix sum of counts[0 .. n) in place, total to counts[n]; one workgroup (n is a few 10^4 at most)
__global__ __launch_bounds__(1024) void legacy_scan_kernel(unsigned long long* counts, long long n) {
    __shared__ unsigned long long part[1024];
    const long long per = (n + 1023) / 1024;
    const long long lo = (long long)threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    unsigned long long s = 0;
    for (long long i = lo; i < hi; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) {
            const unsigned long long v = part[i];
            part[i] = run;
            run += v;
        }
        counts[n] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (long long i = lo; i < hi; ++i) {
        const unsigned long long v = counts[i];
        counts[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(kLgThreads) void legacy_emit_kernel(const LegacyParams p) {
    constexpr int kWaves = kLgThreads / 64, kCells = kLgPerThread * kWaves;       // (row, wave) cells of 64 attempts, in attempt order
    static_assert(kCells <= 64, "one wave scans the cells");
    __shared__ unsigned int cell[kCells];
    __shared__ __attribute__((aligned(16))) double lntab[256];
    const long long wg_rank = p.rank_base + (long long)p.wg_counts[blockIdx.x];   // accepted attempts before this workgroup
    if (wg_rank >= p.pairs_needed) return;                                         // (uniform) nothing of it is wanted
    if (threadIdx.x < 128)
        reinterpret_cast<double2*>(lntab)[threadIdx.x] = reinterpret_cast<const double2*>(g_glibc_log_table)[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long base = p.t_first + (long long)blockIdx.x * kLgAttempts + threadIdx.x;
    const long long t_end = p.t_first + p.t_count;
    // row j: attempts base + 256 j.  Accept bit, the lane's rank inside its cell, the cell's count
    double x1[kLgPerThread], x2[kLgPerThread], r2[kLgPerThread];
    unsigned int before[kLgPerThread];
    unsigned int mask = 0;
#pragma unroll
    for (int j = 0; j < kLgPerThread; ++j) {
        const long long t = base + (long long)j * kLgThreads;
        unsigned int w[4];
        bool acc = false;
        x1[j] = x2[j] = r2[j] = 0.0;
        if (t < t_end) acc = legacy_attempt(p, t, x1[j], x2[j], r2[j], w);
        const unsigned long long b = __ballot(acc);
        before[j] = __builtin_amdgcn_mbcnt_hi((unsigned int)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)b, 0u));
        if (lane == 0) cell[j * kWaves + wave] = (unsigned int)__popcll(b);
        mask |= (acc ? 1u : 0u) << j;
    }
    __syncthreads();
    if (wave == 0) {                               // exclusive scan of the cell counts
        const unsigned int v = lane < kCells ? cell[lane] : 0u;
        unsigned int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane < kCells) cell[lane] = incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kLgPerThread; ++j) {
        if (!((mask >> j) & 1u)) continue;
        const long long rank = wg_rank + cell[j * kWaves + wave] + before[j];
        if (rank >= p.pairs_needed) continue;
        const double f = __dsqrt_rn(__ddiv_rn(rcl::mul_rn(-2.0, rcl::log_glibc_fma(r2[j], lntab)), r2[j]));
        const double val[2] = {rcl::mul_rn(f, x2[j]), rcl::mul_rn(f, x1[j])};       // returned first, cached second
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long e = p.e_shift + 2 * rank + h;
            if (e < p.n_total) {
                long long pi;
                const long long slot = rcl::stream_slot(e, p.period, p.skip, &pi);
                if (slot >= 0) p.out[slot] = rcl::add_rn(0.0, rcl::mul_rn(p.scales[pi], val[h]));   // loc + scale * g
            }
        }
        if (rank == p.pairs_needed - 1) {
            const long long t = base + (long long)j * kLgThreads;
            const unsigned int* src = p.raw + p.w_first + 4 * (t - p.t_first);
            p.last[0] = t;
#pragma unroll
            for (int i = 0; i < 4; ++i) p.last[1 + i] = (long long)src[i];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// `directional_perturbation` on the legacy stream (noise_model.py:183-189), device side.  Per sample the reference
// consumes np.random.randint(0, ndir) - masked rejection on 32-bit outputs: a VARIABLE number of words - and then two
// legacy normals = one accepted polar attempt (a variable number of 4-word attempts).  So a sample's first word depends
// on everything before it: a sequential parse.  It is cut in three:
//   1. dir_len_kernel (parallel over ALL word positions p): how many words would a sample STARTING at p consume?  One
//      byte per position.
//   2. the host walks from the generator's position and collects the sample starts - one dependent load per GROUP of
//      eight samples (dir_lenk_kernel adds up the lengths of eight consecutive samples for every position; the array
//      comes over in pinned memory): ~0.3 ms per 1e6 samples where one load per sample took 2 ms;
//   3. dir_emit_kernel (parallel over samples): index, and the two normals of the accepted attempt at each start.
// The uint32 stream, the accept / reject decisions and therefore indices and generator state are bit-identical to
// NumPy's; the normals are the device's (ln from the table: a few ulp from libm), as for rc_draws_legacy_f64.
// ------------------------------------------------------------------------------------------------
// len[p - first] for p in [first, W): rcl::dir_sample_len (legacy_rng_core.h, shared with the host unit test)
__global__ __launch_bounds__(256) void dir_len_kernel(const unsigned int* raw, long long first, long long W, unsigned int rng,
                                                      unsigned int mask, unsigned char* len) {
    const long long p = first + (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= W) return;
    len[p - first] = rcl::dir_sample_len(raw, p, W, rng, mask);
}

// words consumed by kDirGroup CONSECUTIVE samples starting at position p (0: one of them is invalid or runs off the buffer):
// the host then walks one dependent load per GROUP instead of one per sample
using rcl::kDirGroup;
__global__ __launch_bounds__(256) void dir_lenk_kernel(const unsigned char* len, long long npos, unsigned short* lenk) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npos) return;
    lenk[p] = rcl::dir_group_len(len, p, npos);
}

struct DirEmitParams {
    const unsigned int* raw;
    const unsigned char* len;         // per-position sample lengths (dir_len_kernel), index = word position - first
    const long long* starts;          // [ceil(n / kDirGroup)] word index of the first word of sample kDirGroup * g
    long long first;
    long long n;
    unsigned int rng, mask;
    int shift;                        // 1: the generator entered with a cached normal (a_i = second normal of sample i-1)
    double sigma;
    int* idx;                         // [n]
    double* ab;                       // [n][2]
    unsigned int* last_words;         // [4] raw words of the LAST sample's accepted attempt
};

__global__ __launch_bounds__(256) void dir_emit_kernel(const DirEmitParams p) {
    __shared__ __attribute__((aligned(16))) double lntab[256];
    if (threadIdx.x < 128)
        reinterpret_cast<double2*>(lntab)[threadIdx.x] = reinterpret_cast<const double2*>(g_glibc_log_table)[threadIdx.x];
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    long long q = p.starts[i / kDirGroup];
    for (int j = 0; j < (int)(i % kDirGroup); ++j) q += p.len[q - p.first];      // (the host walk proved every step valid)
    unsigned int v = 0;
    if (p.rng != 0)
        while (!rcl::dir_int_accept(p.raw[q++], p.mask, p.rng, v)) {}
    double x1, x2, r2;
    unsigned int w[4];
    for (;;) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = p.raw[q + j];
        q += 4;
        if (rcl::polar_attempt(w[0], w[1], w[2], w[3], x1, x2, r2)) break;
    }
    const double f = __dsqrt_rn(__ddiv_rn(rcl::mul_rn(-2.0, rcl::log_glibc_fma(r2, lntab)), r2));
    const double first = rcl::add_rn(0.0, rcl::mul_rn(p.sigma, rcl::mul_rn(f, x2)));    // loc + scale * gauss: returned first
    const double second = rcl::add_rn(0.0, rcl::mul_rn(p.sigma, rcl::mul_rn(f, x1)));   // the cached one
    p.idx[i] = (int)v;
    if (!p.shift) {
        p.ab[2 * i] = first;
        p.ab[2 * i + 1] = second;
    } else {                                       // a_0 is the host's cached normal (written by the host)
        p.ab[2 * i + 1] = first;
        if (i + 1 < p.n) p.ab[2 * (i + 1)] = second;
    }
    if (i == p.n - 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p.last_words[j] = w[j];
    }
}

// ------------------------------------------------------------------------------------------------
// The sequential walk of step 2 ON THE DEVICE (round 4).  A sample chain is a linked list through the position array
// (next(p) = p + len[p]); what the host walk needs - 16 MB of group lengths over PCIe, one dependent load per eight samples,
// the starts back over PCIe: ~1 ms per 1e6 samples - becomes a three-level composition of "entry offset -> exit offset"
// maps, every level a few hundred DEPENDENT LDS reads (50 ns each) instead of ~1e5 dependent host loads:
//   dir_blk_kernel   the positions are cut into blocks of kDwB; a chain enters a block at most kDirMaxLen words in.  One
//                    wave per block, lane o walks the chain entering at offset o < kDwE (the block's lengths in LDS):
//                    exit offset into the next block + samples counted.  Chains merge within a few samples, the lanes
//                    are redundant on purpose - it is what makes the step parallel over blocks.
//   dir_sup_kernel   the same composition over kDwS consecutive blocks (their tables in LDS): entry -> exit, count.
//   dir_top_kernel   ONE thread walks the superblocks from the generator's position (offset 0 of block 0): every
//                    superblock's entry offset and the number of samples before it;
//   dir_desc_kernel  one thread per superblock walks its blocks: every block's entry offset and sample base;
//   dir_emit_blk_kernel  one workgroup per block: thread 0 walks the block's chain into an LDS list of sample starts,
//                    then the workgroup emits them (index + the accepted attempt's two normals, as dir_emit_kernel).
//   dir_final_kernel the generator's position after the last sample, its state block and the last attempt's words in ONE
//                    small record for the host.
// A table entry of 255 means "not followed": an invalid length (the chain leaves the buffer, or a sample longer than
// kDirMaxLen) or an exit offset >= kDwE (a sample of more than 64 words across a block boundary: probability ~1e-9 per
// boundary).  The walks stop there; if the n-th sample lies behind such a point no block reports the final position and the
// host falls back to its own walk (kept: directional_chunk) - the result is exact either way.
// ------------------------------------------------------------------------------------------------
constexpr int kDwB = 2048;            // positions per block
constexpr int kDwE = 64;              // entry offsets followed per block
constexpr int kDwS = 256;             // blocks per superblock, at most (the driver takes 64 while that leaves <= kDwMaxSup superblocks:
                                      // the superblock walks are serial, 50-130 ns per block)
constexpr int kDwMaxSup = 128;        // superblocks per pass (dir_top_kernel's tables in LDS)
constexpr int kDwMaxSamples = kDwB / 4 + 8;     // samples starting in one block (a sample is at least 4 words long)

// Inside a block the chain is followed eight samples at a time: for EVERY position q of the block h8[q] = words consumed by
// up to eight consecutive samples starting at q (stopping early once the chain has left the block), c8[q] = how many that
// were - computed by all threads side by side (eight dependent LDS reads each) - so that the serial walk of a chain through
// the block takes ~B / (8 x 6.3) steps instead of B / 6.3.  h8 = 0xffff: an invalid length on the way.
__device__ __forceinline__ void dw_stage_block(const unsigned char* len, long long npos, long long base, unsigned char* l,
                                               unsigned short* h8, unsigned char* c8, int nthreads) {
    for (int i = threadIdx.x * 16; i < kDwB; i += nthreads * 16) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (base + i + 16 <= npos) v = *reinterpret_cast<const uint4*>(len + base + i);
        else {
            unsigned char t[16];
            for (int j = 0; j < 16; ++j) t[j] = (base + i + j < npos) ? len[base + i + j] : (unsigned char)0;
            v = *reinterpret_cast<const uint4*>(t);
        }
        *reinterpret_cast<uint4*>(l + i) = v;
    }
    __syncthreads();
    for (int q0 = threadIdx.x; q0 < kDwB; q0 += nthreads) {
        int q = q0, c = 0;
        bool bad = false;
#pragma unroll 1
        for (int j = 0; j < 8 && q < kDwB; ++j) {
            const int v = l[q];
            if (v == 0 || v == 255) {
                bad = true;
                break;
            }
            q += v;
            ++c;
        }
        h8[q0] = bad ? (unsigned short)0xffff : (unsigned short)(q - q0);
        c8[q0] = (unsigned char)c;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void dir_blk_kernel(const unsigned char* len, long long npos, unsigned char* blk_exit,
                                                      unsigned short* blk_cnt) {
    __shared__ __attribute__((aligned(16))) unsigned char l[kDwB];
    __shared__ unsigned short h8[kDwB];
    __shared__ unsigned char c8[kDwB];
    const long long b = blockIdx.x;
    dw_stage_block(len, npos, b * kDwB, l, h8, c8, 256);
    if (threadIdx.x >= kDwE) return;
    int q = threadIdx.x, c = 0;
    bool bad = false;
    while (q < kDwB) {
        const int h = h8[q];
        if (h == 0xffff) {
            bad = true;
            break;
        }
        c += c8[q];
        q += h;
    }
    const int ex = q - kDwB;
    blk_exit[b * kDwE + threadIdx.x] = (bad || ex >= kDwE) ? (unsigned char)255 : (unsigned char)ex;
    blk_cnt[b * kDwE + threadIdx.x] = (unsigned short)c;
}

// tables of the blocks [b0, b0 + nb) of one superblock into LDS
__device__ __forceinline__ void dw_stage_tables(const unsigned char* blk_exit, const unsigned short* blk_cnt, long long b0, int nb,
                                                unsigned char* ex, unsigned short* cn, int nthreads) {
    const uint4* se = reinterpret_cast<const uint4*>(blk_exit + b0 * kDwE);
    const uint4* sc = reinterpret_cast<const uint4*>(blk_cnt + b0 * kDwE);
    for (int i = threadIdx.x; i < nb * kDwE / 16; i += nthreads) reinterpret_cast<uint4*>(ex)[i] = se[i];
    for (int i = threadIdx.x; i < nb * kDwE / 8; i += nthreads) reinterpret_cast<uint4*>(cn)[i] = sc[i];
    __syncthreads();
}

__global__ __launch_bounds__(64) void dir_sup_kernel(const unsigned char* blk_exit, const unsigned short* blk_cnt, long long nblk,
                                                     int S, unsigned char* sup_exit, unsigned int* sup_cnt) {
    __shared__ __attribute__((aligned(16))) unsigned char ex[kDwS * kDwE];
    __shared__ __attribute__((aligned(16))) unsigned short cn[kDwS * kDwE];
    const long long sb = blockIdx.x;
    const long long b0 = sb * S;
    const int nb = (int)((nblk - b0 < S) ? (nblk - b0) : S);
    dw_stage_tables(blk_exit, blk_cnt, b0, nb, ex, cn, 64);
    int e = threadIdx.x;
    unsigned int c = 0;
    bool bad = false;
    for (int j = 0; j < nb; ++j) {
        const int x = ex[j * kDwE + e];
        c += cn[j * kDwE + e];
        if (x == 255) {
            bad = true;
            break;
        }
        e = x;
    }
    sup_exit[sb * kDwE + threadIdx.x] = bad ? (unsigned char)255 : (unsigned char)e;
    sup_cnt[sb * kDwE + threadIdx.x] = c;
}

// entry offset (255 = the walk does not get there / not needed) and samples before, per superblock
__global__ __launch_bounds__(64) void dir_top_kernel(const unsigned char* sup_exit, const unsigned int* sup_cnt, int nsup, long long n,
                                                     unsigned char* sup_entry, long long* sup_base) {
    __shared__ __attribute__((aligned(16))) unsigned char ex[kDwMaxSup * kDwE];
    __shared__ __attribute__((aligned(16))) unsigned int cn[kDwMaxSup * kDwE];
    for (int i = threadIdx.x; i < nsup * kDwE / 16; i += 64) reinterpret_cast<uint4*>(ex)[i] = reinterpret_cast<const uint4*>(sup_exit)[i];
    for (int i = threadIdx.x; i < nsup * kDwE / 4; i += 64) reinterpret_cast<uint4*>(cn)[i] = reinterpret_cast<const uint4*>(sup_cnt)[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    int e = 0;
    long long base = 0;
    bool live = true;
    for (int sb = 0; sb < nsup; ++sb) {
        if (!live || base >= n) {
            sup_entry[sb] = 255;
            sup_base[sb] = base;
            live = false;
            continue;
        }
        sup_entry[sb] = (unsigned char)e;
        sup_base[sb] = base;
        const int x = ex[sb * kDwE + e];
        base += cn[sb * kDwE + e];
        if (x == 255) live = false;
        else e = x;
    }
}

__global__ __launch_bounds__(64) void dir_desc_kernel(const unsigned char* blk_exit, const unsigned short* blk_cnt, long long nblk,
                                                      int S, long long n, const unsigned char* sup_entry, const long long* sup_base,
                                                      unsigned char* blk_entry, long long* blk_base) {
    __shared__ __attribute__((aligned(16))) unsigned char ex[kDwS * kDwE];
    __shared__ __attribute__((aligned(16))) unsigned short cn[kDwS * kDwE];
    const long long sb = blockIdx.x;
    const long long b0 = sb * S;
    const int nb = (int)((nblk - b0 < S) ? (nblk - b0) : S);
    const int e0 = sup_entry[sb];
    if (e0 == 255) {                                 // wave-uniform: nothing of this superblock is needed
        for (int j = threadIdx.x; j < nb; j += 64) blk_entry[b0 + j] = 255;
        return;
    }
    dw_stage_tables(blk_exit, blk_cnt, b0, nb, ex, cn, 64);
    if (threadIdx.x != 0) return;
    int e = e0;
    long long base = sup_base[sb];
    bool live = true;
    for (int j = 0; j < nb; ++j) {
        if (!live || base >= n) {
            blk_entry[b0 + j] = 255;
            live = false;
            continue;
        }
        blk_entry[b0 + j] = (unsigned char)e;
        blk_base[b0 + j] = base;
        const int x = ex[j * kDwE + e];
        base += cn[j * kDwE + e];
        if (x == 255) live = false;
        else e = x;
    }
}

struct DirWalkResult {
    long long wf;                     // position (relative to `first`) right after the last sample; -1: not reached
    unsigned int last_words[4];       // raw words of the last sample's accepted attempt
    unsigned int key[624];            // the state block the generator stands in afterwards (dir_final_kernel)
    int pos;                          // ... and its position in it (NumPy's convention: 1 .. 624)
    int ok;                           // 1: everything above is valid
};

struct DirEmitBlkParams {
    const unsigned int* raw;
    const unsigned char* len;         // [npos] per-position sample lengths, index = word position - first
    const unsigned char* blk_entry;   // [nblk]
    const long long* blk_base;        // [nblk]
    long long first, npos, n;
    unsigned int rng, mask;
    int shift;
    double sigma;
    int* idx;
    double* ab;
    DirWalkResult* res;
};

__global__ __launch_bounds__(256) void dir_emit_blk_kernel(const DirEmitBlkParams p) {
    __shared__ __attribute__((aligned(16))) double lntab[256];
    __shared__ __attribute__((aligned(16))) unsigned char l[kDwB];
    __shared__ unsigned short h8[kDwB];
    __shared__ unsigned char c8[kDwB];
    __shared__ unsigned short st8[kDwMaxSamples / 8 + 2];      // start of every group of eight samples
    __shared__ int s_cnt;
    const long long b = blockIdx.x;
    const int e0 = p.blk_entry[b];
    if (e0 == 255) return;                           // workgroup-uniform
    const long long pbase = b * kDwB;
    if (threadIdx.x < 128)
        reinterpret_cast<double2*>(lntab)[threadIdx.x] = reinterpret_cast<const double2*>(g_glibc_log_table)[threadIdx.x];
    dw_stage_block(p.len, p.npos, pbase, l, h8, c8, 256);
    const long long sbase = p.blk_base[b];
    if (threadIdx.x == 0) {
        // groups of eight samples from the block's entry (only the LAST group of a block can be shorter: h8 stops where the
        // chain leaves the block); the samples wanted from this block: up to the n-th of the call
        int q = e0, c = 0, g = 0;
        while (q < kDwB && sbase + c < p.n) {
            const int h = h8[q];
            if (h == 0xffff) break;                  // the chain leaves what was followed: no final position from this block
            st8[g++] = (unsigned short)q;
            c += c8[q];
            q += h;
        }
        const long long want = p.n - sbase;
        s_cnt = (c < want) ? c : (int)want;
    }
    __syncthreads();
    const int cnt = s_cnt;
    for (int j = threadIdx.x; j < cnt; j += 256) {
        const long long i = sbase + j;
        int qs = st8[j >> 3];
        for (int k = 0; k < (j & 7); ++k) qs += l[qs];       // (the group walk proved every step valid)
        if (i == p.n - 1) p.res->wf = pbase + qs + l[qs];    // the last sample of the call: the generator stands behind it
        long long q = p.first + pbase + qs;
        unsigned int v = 0;
        if (p.rng != 0)
            while (!rcl::dir_int_accept(p.raw[q++], p.mask, p.rng, v)) {}
        double x1, x2, r2;
        unsigned int w[4];
        for (;;) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = p.raw[q + k];
            q += 4;
            if (rcl::polar_attempt(w[0], w[1], w[2], w[3], x1, x2, r2)) break;
        }
        const double f = __dsqrt_rn(__ddiv_rn(rcl::mul_rn(-2.0, rcl::log_glibc_fma(r2, lntab)), r2));
        const double first = rcl::add_rn(0.0, rcl::mul_rn(p.sigma, rcl::mul_rn(f, x2)));    // loc + scale * gauss: returned first
        const double second = rcl::add_rn(0.0, rcl::mul_rn(p.sigma, rcl::mul_rn(f, x1)));   // the cached one
        p.idx[i] = (int)v;
        if (!p.shift) {
            p.ab[2 * i] = first;
            p.ab[2 * i + 1] = second;
        } else {                                     // a_0 is the host's cached normal (written by the host)
            p.ab[2 * i + 1] = first;
            if (i + 1 < p.n) p.ab[2 * (i + 1)] = second;
        }
        if (i == p.n - 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p.res->last_words[k] = w[k];
        }
    }
}

// the generator's state after the walk: block and position of global word first + wf (NumPy's convention at a block
// boundary: pos = 624 of the block before), copied into the result record
__global__ __launch_bounds__(256) void dir_final_kernel(const unsigned int* raw, long long first, long long words, DirWalkResult* res) {
    const long long wrel = res->wf;
    if (wrel < 0) {
        if (threadIdx.x == 0) res->ok = 0;
        return;
    }
    const long long wf = first + wrel;
    long long blk = wf / rcl::kMtN, pos = wf % rcl::kMtN;
    if (pos == 0) {
        blk -= 1;
        pos = rcl::kMtN;
    }
    if (blk < 0 || (blk + 1) * rcl::kMtN > words) {
        if (threadIdx.x == 0) res->ok = 0;
        return;
    }
    for (int i = threadIdx.x; i < rcl::kMtN; i += 256) res->key[i] = raw[blk * rcl::kMtN + i];
    if (threadIdx.x == 0) {
        res->pos = (int)pos;
        res->ok = 1;
    }
}
