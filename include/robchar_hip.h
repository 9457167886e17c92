/* robchar_hip.h - C ABI of librobchar_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the Monte-Carlo robustness-characterisation hot path of RobChar.  The
 * reference is pure Python and has no FFI layer; each entry point below replaces a Python loop nest of the
 * reference and is what a ctypes binding on the reference side would call:
 *
 *   rc_mc_fidelity_f64*   replaces LOOP 2 x LOOP 3 of MCDataSim.get_algo_fid_dist (mcsim.py:434-456), i.e.
 *                         C x K calls of noise_model_base.evaluate_noisy_fidelity(x, ham_noisy=True)
 *                         (noise_model.py:98-109) with structured_perturbation.perturbation
 *                         (noise_model.py:122-147), for ONE sigma_sim level.
 *   rc_reduce_f64*        replaces the per-controller metric maps of mcsim.py:144-183 as applied in
 *                         get_metric_dict_from_scratch (mcsim.py:480-500): RIM_1 = wd_from_ideal
 *                         (wd_sortof_fast_implementation.py:82-116), np.std, min, Q(threshold), each for
 *                         the centre / DKW-upper / DKW-lower tensors, plus the sorted sample (ECDF).
 *                         rim1 (= 1 - mean_k fidelity) is also the reduction of NStochOpt.get_rims
 *                         (gen_fig_8_arim_fcall_scaling.py:121-132).
 *   rc_rim_p_f64*         replaces RIM_p (wd_sortof_fast_implementation.py:147-174).
 *   rc_mc_*_sharded_f64   the same loops over all GPUs of a node from one process (the role of the reference's dead
 *                         multiprocessing.Pool, mcsim.py:451-455): controller blocks, host arrays in / out.
 *   rc_draws_legacy_f64   the reference's RNG itself - NumPy's legacy RandomState normal stream (noise_model.py:114-115,
 *                         :137-146, the burned draw of mcsim.py:425) - continued on the device, state handed back.
 *   rc_directional_draws_legacy   the interleaved randint / normal(size=2) consumption of directional_perturbation
 *                         (noise_model.py:183-189) on the same stream, on the host; rc_directional_draws_legacy_dev: the
 *                         same with the word stream, the per-position parse and the normals on the device.
 *   rc_draws_philox_f64*  counter-based draws for sample spaces too large for a sequential stream (not the reference's RNG).
 *   rc_json_*             json.dump of the fidelity / metric tensors into the .mc / .mcm caches (mcsim.py:457-459, :501).
 *
 * Conventions: every function returns 0 on success and a negative RC_E* code on failure, with a
 * human-readable message available from rc_last_error() (thread-local).  The caller owns every buffer.
 * All inputs, outputs and results are IEEE fp64, accurate to the 1e-10 the reference's complex128 path is matched to (in
 * practice ~1e-15).  Internally the chain kernels compute their eigenvalue STARTING VALUES in fp32 (N = 3..13) and finish
 * them in fp64; nothing of fp32 accuracy reaches a result.  No Python / torch types cross this boundary.
 *
 * Data layout (row-major, fp64):
 *   controllers [C][N+1]      x[0..N-1] = biases, x[N] = time (abs() is taken, noise_model.py:99).
 *                             A row containing NaN marks a padded controller (mcsim.py:442-443):
 *                             its K outputs are NaN and its draws are not read.
 *   draws       [C][K][N][3]  (g0_i, g1_i, g2_i) per site i in the order the reference consumes its RNG
 *                             (noise_model.py:137-146), ALREADY scaled by sigma_sim.  g1_0, g2_0 are ignored.
 *   fid_out     [C][K]
 *   h0_diag     [N] or NULL   static diagonal added to every sample (NULL = 0; the XXZ term of
 *                             qnewton.py:148-150 goes here).  HOST pointer.
 *   h0_offdiag  [N-1] or NULL static couplings (NULL = 1.0, noise_model.py:80-82).  HOST pointer.
 */
#ifndef ROBCHAR_HIP_H
#define ROBCHAR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RC_ABI_VERSION 6       /* 2: + multi-device entries, legacy-stream draws, JSON cache encoder, RC_KERNEL_RING_HH;
                                  3: + rc_stats_polish_tiles; 4: + rc_directional_draws_legacy_dev;
                                  5: + rc_reserve_ring, rc_release_stream, rc_mc_fidelity_directional_f64_async,
                                     rc_mc_fidelity_philox_f64_async;
                                  6: + rc_build_flags, rc_philox_fused_pays, rc_reduce_ex_f64_async,
                                     rc_legacy_log_is_host_exact, rc_comm_init / rc_comm_size / rc_comm_destroy /
                                     rc_mc_metrics_gathered_f64 (all additive) */
#define RC_MAX_NSPIN 32        /* chain topology: register-resident fast kernels for N <= RC_MAX_NSPIN_CHAIN, a general
                                 * LDS-resident per-sample kernel (same arithmetic, ~10x slower per site) above */
#define RC_MAX_NSPIN_FAST 16   /* limit of the dense kernels (RC_KERNEL_JACOBI, RC_KERNEL_EXPM: ring, non-Hermitian), of the
                                 * mixed-precision / fused-Philox chain kernels and of the rows-mode chain kernel */
#define RC_MAX_NSPIN_CHAIN 24  /* (round 5) chains of 17 .. 24 spins: the register-resident eigenvalue-only kernel (all-fp64 QL +
                                 * adjugate weights, one wave per SIMD) instead of the LDS kernel: N = 17 at 1.2x the N = 16
                                 * time where the LDS kernel took 6x */

#define RC_OK 0
#define RC_EINVAL (-1)   /* bad argument (N out of range, in/out out of range, NULL pointer, ...) */
#define RC_EHIP (-2)     /* a HIP runtime call failed; rc_last_error() carries hipGetErrorString */
#define RC_ENOSUP (-3)   /* valid request that this build does not implement */

/* kernels selectable through rc_set_fidelity_kernel / the `kernel` argument */
#define RC_KERNEL_AUTO 0      /* chain -> TRIDIAG_ADJ; ring -> the mixed-precision lane-per-sample route + RING_HH as its repair (N <= 16) */
#define RC_KERNEL_TRIDIAG_QL 1 /* lane-per-sample real-symmetric-tridiagonal implicit QL (chain only) */
#define RC_KERNEL_TRIDIAG_ADJ 3 /* same QL on eigenvalues only; eigenvector weights from the adjugate of (lambda I - H) (chain only) */
#define RC_KERNEL_EXPM 4       /* dense complex Pade scaling-and-squaring expm in LDS, one wavefront per sample (any topology) */
#define RC_KERNEL_JACOBI 2     /* complex Hermitian cyclic Jacobi in LDS, one wavefront per sample (chain or ring) */
#define RC_KERNEL_RING_HH 5    /* ring only: lane-per-sample reduction to tridiagonal form in registers (N <= 10: dense Householder;
                                 * N = 11 .. 16, round 5: the ring folded into a pentadiagonal band + band reduction) + the QL of the chain kernels */

int rc_version(void);
int rc_device_count(void);
const char* rc_last_error(void);

/* (ABI 6) Compile-time switches of THIS build that change results or remove a safety net; 0 = the product build.  The
 * RC_BUILD_EXPERIMENT_* builds (scripts/build_variant.sh: timing experiments) knowingly return wrong fidelities for some
 * samples - a loader must refuse them (code-robchar_amd/_lib.py does, unless ROBCHAR_ALLOW_EXPERIMENT_LIB=1). */
#define RC_BUILD_EXPERIMENT_NO_STEPPING 1
#define RC_BUILD_EXPERIMENT_STEP_NOT_RUN 2
#define RC_BUILD_EXPERIMENT_FALLBACK_NOT_RUN 4
#define RC_BUILD_EXPERIMENT_PHILOX_NOSTORE 8
#define RC_BUILD_DEV_FEW_N 16            /* chain kernels for N = 5, 7, 10 only (kernel-tuning builds) */
#define RC_BUILD_STAMPS 32               /* per-wave s_memtime stamps compiled in */
#define RC_BUILD_NO_SUM_RULE_GUARD 64    /* -DRC_SUM_RULE_GUARD=0 */
#define RC_BUILD_NO_KEEP_SETTLED 128     /* -DRC_KEEP_SETTLED=0 */
#define RC_BUILD_WRONG_RESULTS_MASK 15   /* the bits under which some results are knowingly wrong */
int rc_build_flags(void);

/* Blocking call.  `controllers`, `draws`, `fid_out` may each be a host pointer or a device pointer on
 * `device` (detected with hipPointerGetAttributes); host buffers are staged through an internal per-device
 * workspace.  Runs on an internal per-device stream and returns after the result is in `fid_out`. */
int rc_mc_fidelity_f64(int device, int N, int in, int out,
                       const double* h0_diag, const double* h0_offdiag, int ring,
                       const double* controllers, const double* draws,
                       long long C, long long K, double* fid_out);

/* Same, with an explicit kernel choice (RC_KERNEL_*) instead of the process-wide default. */
int rc_mc_fidelity_kernel_f64(int device, int kernel, int N, int in, int out,
                              const double* h0_diag, const double* h0_offdiag, int ring,
                              const double* controllers, const double* draws,
                              long long C, long long K, double* fid_out);

/* Enqueue-only variant: all three arrays are DEVICE pointers on `device`; the kernel is launched on
 * `stream` (a hipStream_t; NULL = the device's default stream) and the call returns without synchronising.
 * `kernel` is one of RC_KERNEL_*. */
int rc_mc_fidelity_f64_async(int device, void* stream, int kernel, int N, int in, int out,
                             const double* h0_diag, const double* h0_offdiag, int ring,
                             const double* controllers_dev, const double* draws_dev,
                             long long C, long long K, double* fid_out_dev);

/* Ring topology through the enqueue-only entries (ring = 1, N <= 16, RC_KERNEL_AUTO): the mixed-precision route lists the
 * samples it does not trust itself with and a repair launch behind it recomputes them; list and counters live in a buffer
 * the library keeps per (device, stream), allocated, zeroed, grown and released IN STREAM ORDER on that stream - an enqueue
 * never synchronises the device.  (ABI 5)
 *   rc_reserve_ring(device, stream, samples): size that buffer for launches of up to `samples` = C * K now, e.g. outside a
 *     latency-critical region (otherwise the first ring launch on the stream, or a larger one, enqueues the allocation).
 *   rc_release_stream(device, stream): hand the stream's buffer back (stream-ordered free behind its last launch); call
 *     before destroying a stream that ran ring launches.  Both return RC_OK when there is nothing to do. */
int rc_reserve_ring(int device, void* stream, long long samples);
int rc_release_stream(int device, void* stream);

/* Extended enqueue-only variant: `draws_ctrl_stride` is the distance, in doubles, between the draw blocks of
 * consecutive controllers: K*N*3 (or any larger pitch) = every controller has its own K draws, as above;
 * 0 = ONE set of K draws [K][N][3] shared by every controller.  The shared form is the optimiser-side noisy
 * objective of the reference (qnewton.py:383-444, `fidelity_ss_av` over the fixed Hamiltonian sets built by
 * `randHset_constructor`, qnewton.py:122-137), whose perturbation is real (2 draws per site: set g2 = 0). */
int rc_mc_fidelity_ex_f64_async(int device, void* stream, int kernel, int N, int in, int out,
                                const double* h0_diag, const double* h0_offdiag, int ring,
                                const double* controllers_dev, const double* draws_dev,
                                long long draws_ctrl_stride, long long C, long long K, double* fid_out_dev);

/* The same fidelities with the COUNTER-BASED draws generated inside the kernel (ABI 5; SURVEY.md 8(d): "in philox mode draws
 * are not read"): sample (c, k), site i, slot s is element  offset + ((c K + k) N + i) 3 + s  of stream `seed` - exactly what
 * rc_draws_philox_f64_async(seed, offset, C K N 3, sigma) would have written, by the same routine, so the result is BIT-IDENTICAL
 * to generating the [C][K][N][3] tensor and calling rc_mc_fidelity_f64_async on it; only that tensor (24 N bytes per sample:
 * 16.8 GB for BASELINE config 4) never exists.  `sigma_rows_dev` [C] (or NULL: `sigma` for every row): scale per controller row,
 * so that all sigma levels of an algorithm go through one launch with the controller rows tiled.  Chain topology, the
 * eigenvalue-only kernels (kernel = RC_KERNEL_AUTO or RC_KERNEL_TRIDIAG_ADJ), N <= RC_MAX_NSPIN_FAST; RC_ENOSUP otherwise
 * (generate the tensor instead).  It is the faster route up to N = 13 (0.70 - 0.86 of the two-kernel route's time) and for
 * end-to-end pairs at N = 14; beyond that the two-kernel route is ~7 % faster.  Enqueue-only.  This replaces LOOP 2 x LOOP 3 of mcsim.py:434-456 together with the
 * perturbation draws of noise_model.py:122-147 for callers who do not need the reference's RNG stream. */
int rc_mc_fidelity_philox_f64_async(int device, void* stream, int kernel, int N, int in, int out,
                                    const double* h0_diag, const double* h0_offdiag, const double* controllers_dev,
                                    unsigned long long seed, unsigned long long offset, double sigma,
                                    const double* sigma_rows_dev, long long C, long long K, double* fid_out_dev);

/* (ABI 6) 1 when the kernel above is the faster of the two bit-identical routes for this geometry (N <= 13, or N = 14 with
 * {in, out} = {0, N-1}), else 0; 0 everywhere when ROBCHAR_PHILOX_FUSED=0 is in the environment (read per call).  The ONE
 * copy of that rule: the sharded entries below and the Python layer (backend.philox_fused_pays) both ask it. */
int rc_philox_fused_pays(int N, int in, int out);

/* Non-Hermitian variant: `diag_imag_dev` [C][K][N] (or NULL) is added to the diagonal as an IMAGINARY part,
 * H[i][i] += 1j * diag_imag.  Chains up to N = 12: a lane-per-sample complex symmetric QL kernel (the couplings stay
 * Hermitian pairs, so the diagonal gauge makes them real and leaves a complex symmetric tridiagonal matrix), with the dense
 * Pade-expm kernel (RC_KERNEL_EXPM's) as repair pass over the samples it marks; rings and N > 12: the expm kernel for every
 * sample (also when RC_NH_EXPM_ONLY=1 is in the environment - the cross-check).  This is the draw layout of the reference's
 * `directional_perturbation` (noise_model.py:150-201): a sample perturbs ONE element pair; a bond direction maps
 * onto (g1, g2) of the ordinary draws, a diagonal direction (i,i) ends up as a - ib on the diagonal because the
 * second assignment (noise_model.py:198-199) overwrites the first: g0_i = a, diag_imag_i = -b. */
int rc_mc_fidelity_nh_f64_async(int device, void* stream, int N, int in, int out,
                                const double* h0_diag, const double* h0_offdiag, int ring,
                                const double* controllers_dev, const double* draws_dev,
                                const double* diag_imag_dev, long long C, long long K, double* fid_out_dev);

/* `directional_perturbation` (noise_model.py:150-201) evaluated straight from what its RNG consumption leaves per sample
 * (ABI 5): idx_dev [C][K] (int32) = the direction index `np.random.randint(0, len(directions))` drew, ab_dev [C][K][2] = the
 * two normals of `rng(size=2)`, already scaled by sigma - exactly what rc_directional_draws_legacy_dev writes.  The sample
 * perturbs ONE element pair: z[p,q] = a + ib, z[q,p] = a - ib with (p, q) = directions[idx] in the reference's list order
 * [(0,0), (N-1,N-1), (d,d-1), (d,d), (d,d+1) for d = 1..N-2, (0,1), (1,0), (N-2,N-1), (N-1,N-2)]; for p = q the second
 * assignment wins (H[p][p] += a - ib: non-Hermitian).  No (C, K, N, 3) draw tensor exists: the samples are partitioned by
 * class on the device and evaluated lane per sample - bond directions by the real tridiagonal routes, diagonal directions by
 * the complex symmetric QL route, whatever neither settles by the Pade-expm kernel.  Chain topology, N <= 12 (RC_ENOSUP
 * otherwise: build the dense layout and call rc_mc_fidelity_nh_f64_async).  Enqueue-only; the workspace (8 bytes per
 * sample) is allocated and released in stream order. */
int rc_mc_fidelity_directional_f64_async(int device, void* stream, int N, int in, int out,
                                         const double* h0_diag, const double* h0_offdiag, int ring,
                                         const double* controllers_dev, const int* idx_dev, const double* ab_dev,
                                         long long C, long long K, double* fid_out_dev);

/* Per-controller reductions over K.  Outputs are variant-major with 3 variants in the order
 *   0: centre  F          1: " upper"  clip(F - dkw_eps, 0, 1)        2: " lower"  clip(F + dkw_eps, 0, 1)
 * (naming of mcsim.py:484-485).  Shapes: rim1/std_/minf [3][C];  q [3][nq][C] = fraction of samples >=
 * q_thresholds[j] (the reference stores the NEGATED value; signs are applied by the host layer).
 * std_ is the population standard deviation (np.std).  NaN rows give NaN (q: 0).
 * sorted_out: NULL, or [C][K] receiving each row sorted ascending (NaN rows copied through); any K (one
 * launch for K <= 16384, a bitonic network with HBM passes above; internal grow-only workspace).
 * Any output pointer may be NULL to skip it.  nq <= 8.  q_thresholds is a HOST pointer.
 * Summation order: fixed per (K, route) - bitwise reproducible from run to run.  The kernel is chosen by K (one workgroup of
 * 128 / 256 / 512 threads per row for K <= 4096 / 8192 / above); for K <= 2048 also by C (C >= 64: one wave per row), so below
 * that length the last bits of mean / std of a row may depend on how many rows are reduced together, above it they do not;
 * for 8192 < K <= 10240 also by the overlap hint of rc_reduce_ex_f64_async (this blocking entry: RC_REDUCE_STANDALONE). */
int rc_reduce_f64(int device, const double* fid, long long C, long long K,
                  const double* q_thresholds, int nq, double dkw_eps,
                  double* rim1, double* std_, double* minf, double* q, double* sorted_out);

int rc_reduce_f64_async(int device, void* stream, const double* fid_dev, long long C, long long K,
                        const double* q_thresholds, int nq, double dkw_eps,
                        double* rim1_dev, double* std_dev, double* minf_dev, double* q_dev,
                        double* sorted_out_dev);

/* (ABI 6) The same with a hint about what runs BESIDE the reduction.  flags = 0: rc_reduce_f64_async (a caller that overlaps
 * the reduction with fidelity launches on another stream: the latency-bound 512-thread route for rows of 8193 .. 10 240 values
 * fills issue slots those kernels leave idle).  RC_REDUCE_STANDALONE: nothing overlaps it - rows of that length go through 256
 * threads x 40 register-cached values, five rows in flight per CU instead of two (11 000 rows of 10 000 values: 440 -> 208 us;
 * 1 000 rows: 46 -> 29.5 us).  The blocking rc_reduce_f64 and the multi-device entries always reduce standalone.  The flag picks
 * the route and with it the summation order of rows of THAT length: mean / std may differ in the last bits between the two. */
#define RC_REDUCE_STANDALONE 1
int rc_reduce_ex_f64_async(int device, void* stream, const double* fid_dev, long long C, long long K,
                           const double* q_thresholds, int nq, double dkw_eps,
                           double* rim1_dev, double* std_dev, double* minf_dev, double* q_dev,
                           double* sorted_out_dev, int flags);

/* p-RIM per controller: out[c] = (mean_k (1 - fid[c][k])^p)^(1/p), p > 0.  fid [C][K], out [C]. */
int rc_rim_p_f64(int device, const double* fid, long long C, long long K, double p, double* out);
int rc_rim_p_f64_async(int device, void* stream, const double* fid_dev, long long C, long long K, double p,
                       double* out_dev);

/* Counter-based Gaussian draws on the device (Philox4x32-10 + Box-Muller, fp64), for sample spaces too large
 * to draw on the host.  NOT the reference's RNG (the reference uses numpy's legacy MT19937 stream, which the
 * host layer reproduces exactly); provided for scale, with parity checked by regenerating the same elements on
 * the host (oracle/philox_host.py).  out[i] = scale * z(seed, offset + i), i < n: the stream is indexed by
 * element, so any slice - e.g. one rank's controller block - can be generated independently. */
int rc_draws_philox_f64(int device, unsigned long long seed, unsigned long long offset, long long n, double scale,
                        double* out);
int rc_draws_philox_f64_async(int device, void* stream, unsigned long long seed, unsigned long long offset,
                              long long n, double scale, double* out_dev);

/* The reference's OWN random stream on the device: NumPy's legacy `RandomState` normal generator (MT19937 + polar
 * Box-Muller, what `np.random.normal` at noise_model.py:114-115 draws from), continued from `state` - the caller's
 * `np.random.get_state()`: key[624], pos, has_gauss, cached_gaussian - and left exactly where NumPy would leave it after
 * the same number of draws (uint32 stream, attempt boundaries, cached value: bit-identical; the normals themselves are NumPy's
 * bit for bit where rc_legacy_log_is_host_exact() says so - round 5: the device evaluates glibc's log operation for operation -
 * and within a few ulp otherwise).  The stream is cut into `n_periods` periods of `period` normals; the
 * first `skip` of every period are consumed but dropped (the burned draw of `rng(scale=sigma)`, mcsim.py:425), the others
 * are written contiguously to out_dev[p * (period - skip) ...] multiplied by scales[p] (HOST pointer, [n_periods]).
 * One call = all sigma levels of an algorithm: period = 1 + C*K*3N, skip = 1, scales = the levels.
 * Blocking with respect to `state` (updated on return); the output is produced on `stream`. */
typedef struct rc_mt19937_state {
    unsigned int key[624];
    int pos;
    int has_gauss;
    double gauss;
} rc_mt19937_state;
int rc_draws_legacy_f64(int device, void* stream, rc_mt19937_state* state, long long n_periods, long long period,
                        long long skip, const double* scales, double* out_dev);

/* (ABI 6) The normals of the two device-side legacy entries (rc_draws_legacy_f64, rc_directional_draws_legacy_dev) are NumPy's
 * BIT FOR BIT when the C library's log() is the routine the device restates operation for operation: glibc >= 2.28's
 * table-driven double log in the FMA build its resolver selects on x86-64 with FMA + AVX2 (legacy_rng_core.h: log_glibc_fma;
 * division and square root are correctly rounded on both sides).  1 = verified on this host (2^17 arguments against log()
 * itself, once per process); 0 = another libm: state and attempt boundaries are still exact, the normals may differ from
 * NumPy's in the last bits - a caller that needs identical DRAWS then draws on the host (the Python layer does so by itself). */
int rc_legacy_log_is_host_exact(void);

/* Host-side (CPU) emulation of the RNG consumption of `directional_perturbation.perturbation()` (noise_model.py:183-189)
 * on NumPy's legacy stream `state` (updated): per sample `np.random.randint(0, ndir)` then two legacy normals scaled by
 * sigma.  idx_out [n], ab_out [n][2].  Bit-identical to NumPy (indices, normals, state); ~100x the Python loop. */
int rc_directional_draws_legacy(rc_mt19937_state* state, long long n, int ndir, double sigma, int* idx_out, double* ab_out);

/* The same consumption pattern with the work on the GPU (ABI 4): raw MT19937 words from jump-ahead sub-streams, one
 * kernel finds for EVERY word position how many words a sample starting there would consume, the chain of sample starts
 * through that byte array - the only sequential step - is followed ON THE DEVICE (round 4: "entry offset -> exit offset"
 * maps of 2048-position blocks composed over superblocks, a few hundred dependent LDS reads per level; the host's own walk
 * of round 3 - 16 MB over PCIe, ~2 ns per sample - remains as the fallback for what the device walk does not follow: a
 * sample longer than 64 words across a block boundary), and one kernel emits index + the two normals per sample.  idx_dev
 * [n] (int32) and ab_dev [n][2] (fp64) are DEVICE pointers, filled in stream order on `stream`; `state` is updated on
 * return (the call synchronises the stream once, for a 2.5 KB state record).  Indices and generator state bit-identical to
 * NumPy's, normals within a few ulp (the device's ln), as for rc_draws_legacy_f64.  Environment, read per call (test / A/B
 * knob): RC_DIR_WALK=host forces the host walk, RC_DIR_WALK=fallback runs the device pass and then the host walk. */
int rc_directional_draws_legacy_dev(int device, void* stream, rc_mt19937_state* state, long long n, int ndir, double sigma,
                                    int* idx_dev, double* ab_dev);

/* Diagnostic: number of 64-sample tiles of the chain kernels in which at least one sample left the wave-wide fast path
 * and was repaired per sample on `device` since the last reset; synchronises the device.  A sample is repaired when it has
 * a degenerate eigenvalue pair (closer than 1e-12 of the spectral scale for the end-to-end weights, 4e-6 for the general
 * adjugate weights), when the a-posteriori sum-rule guard of the general adjugate weights rejects it, on the
 * sweep cap, or on overflow.  Rare but not impossible on random workloads - measured on the BASELINE shapes (GPU suite,
 * rounds 3 / 4): 0 tiles for configs 2 and 3 (N launches of 15 700 tiles each, every `out`), ~150-200 for config 5's ten
 * launches, 3 of config 4's 1 563 000; the tests bound it at ~10x those counts (tests/test_gpu_fullsize.py,
 * tests/test_gpu_chain.py).  One such tile costs a 1e6-evaluation launch +0.3 %.  Ring route: repaired WAVES of 64 listed
 * samples (one per 1e6-evaluation launch at N = 7).  Negative on error. */
long long rc_stats_general_tiles(int device, int reset);

/* Diagnostic (ABI 3): tiles of the mixed-precision eigenvalue path (chain kernels, N = 3..13, eigenvalue-only weight modes)
 * in which some sample needed more than the one fp64 step (close eigenvalue pair); such a tile keeps stepping - still
 * on the fast path - and costs ~20 % more.  ~9 % of the tiles of the N = 7 benchmark workload.  Same conventions. */
long long rc_stats_polish_tiles(int device, int reset);

/* Single-process multi-device entry points (the reference's only parallel construct is the dead
 * multiprocessing.Pool of mcsim.py:451-455).  The C controllers are split into `ndev` contiguous balanced blocks (the
 * first C mod ndev devices get one extra), block r on device devices[r] (devices == NULL: 0 .. ndev-1); one host
 * thread and one stream per device, all devices concurrently; every array is a HOST pointer and results are assembled
 * in the caller's host arrays, so no collective is involved.  Blocking; thread-safe (per-device locks).
 *
 * rc_mc_fidelity_sharded_f64: rc_mc_fidelity_kernel_f64 over several devices - controllers [C][N+1], draws
 *   [C][K][N][3], fid_out [C][K].
 * rc_mc_metrics_sharded_f64: fidelity + the per-controller reductions of rc_reduce_f64 on the devices; only the metric
 *   rows come back (rim1 / std_ / minf [3][C], q [3][nq][C]; any may be NULL) plus, optionally, fid_out [C][K] (NULL =
 *   metrics only: nothing of size C x K crosses PCIe).  draws == NULL: the perturbations are generated on the devices
 *   by the counter-based generator of rc_draws_philox_f64 - sample (c, k), site i, slot s is element
 *   philox_offset + ((c K + k) N + i) 3 + s of stream philox_seed, scaled by sigma - so the result does not depend on
 *   ndev (BASELINE config 4: 2.1e9 draws per level never exist on the host).
 *   (Chain, N <= 13 - and end-to-end pairs at N = 14 -, eigenvalue-only kernels: the draws are generated INSIDE the fidelity
 *   kernel - see rc_mc_fidelity_philox_f64_async -, no draw tensor exists and a chunk is bounded by its fidelities alone; at
 *   larger N generating the tensor first is the faster route, with bit-identical results.)
 * Devices process their block in chunks of <= 4 GiB of draws through a grow-only per-device workspace.  A device may
 * be listed once (RC_ALLOW_DUPLICATE_DEVICES=1 in the environment lifts that for rehearsing the multi-block assembly on a
 * one-GPU box: the blocks then take turns on the device). */
int rc_mc_fidelity_sharded_f64(int ndev, const int* devices, int kernel, int N, int in, int out,
                               const double* h0_diag, const double* h0_offdiag, int ring,
                               const double* controllers, const double* draws, long long C, long long K,
                               double* fid_out);
int rc_mc_metrics_sharded_f64(int ndev, const int* devices, int kernel, int N, int in, int out,
                              const double* h0_diag, const double* h0_offdiag, int ring,
                              const double* controllers, const double* draws,
                              unsigned long long philox_seed, unsigned long long philox_offset, double sigma,
                              long long C, long long K, const double* q_thresholds, int nq, double dkw_eps,
                              double* rim1, double* std_, double* minf, double* q, double* fid_out);

/* (ABI 6) The exchange step over RCCL / xGMI from the C ABI (`rc_comm_init` + a communicator handle; north_star:
 * "an RCCL all-gather over xGMI to reassemble per-controller fidelity vectors") - for an integrator without torch.distributed.
 *   rc_comm_init(ndev, devices, &comm): one process, one RCCL communicator per listed device (ncclCommInitAll; devices == NULL:
 *     0 .. ndev-1; a device may be listed once).  librccl.so is resolved at run time - the copy the process already carries
 *     (PyTorch ships one), else librccl.so.1, opened RTLD_LOCAL | RTLD_DEEPBIND (a process that imports PyTorch AFTERWARDS then
 *     holds two RCCL / rocm_smi pairs; kept out of the global scope they do not share their global objects) - so the library
 *     has no link-time dependency on it; RC_ENOSUP when there is none.
 *   rc_mc_metrics_gathered_f64(comm, ...): rc_mc_metrics_sharded_f64's work - controller blocks, one host thread and one stream
 *     per device, draws from the host or (draws == NULL) from the counter-based stream - but the results stay ON THE DEVICES and
 *     are exchanged there: every device all-gathers its block's metric rows and (fid_dev != NULL) its fidelity slab, in ONE RCCL
 *     group on the devices' streams, so that EVERY device ends with everything:
 *       table_dev[r] (device pointer on devices[r], or table_dev == NULL)  [ndev][NR][Cmax],  NR = 9 + 3 nq rows per block:
 *                    rim1[3], std[3], min[3], q[3][nq] (variant order of rc_reduce_f64), Cmax = ceil(C / ndev) columns of
 *                    which block b fills the first C/ndev (+1 for b < C mod ndev); the rest is padding
 *       fid_dev[r]   (or fid_dev == NULL)  [ndev][Cmax][K], same padding
 *       table_host   (or NULL) [NR][C], fid_host (or NULL; with ndev > 1 it needs fid_dev) [C][K]: unpadded copies from devices[0].
 *     Blocking; the results do not depend on ndev (same partition, same stream elements, per-controller reductions).
 * On this pool only a ONE-device communicator can execute (one GPU per box; RCCL refuses two ranks on one device): the
 * multi-device path is correct by construction and rehearsed with ndev = 1 (tests/test_gpu_multidevice.py). */
typedef struct rc_comm rc_comm;
int rc_comm_init(int ndev, const int* devices, rc_comm** comm_out);
int rc_comm_size(const rc_comm* comm);
int rc_comm_destroy(rc_comm* comm);
int rc_mc_metrics_gathered_f64(rc_comm* comm, int kernel, int N, int in, int out,
                               const double* h0_diag, const double* h0_offdiag, int ring,
                               const double* controllers, const double* draws,
                               unsigned long long philox_seed, unsigned long long philox_offset, double sigma,
                               long long C, long long K, const double* q_thresholds, int nq, double dkw_eps,
                               double* const* table_dev, double* const* fid_dev, double* table_host, double* fid_host);

/* Cached-results layout, host side (no GPU involved).  JSON text of a row-major fp64 array of `ndim` (1..8) dimensions
 * as nested lists, exactly parseable by the reference's `json.load` cache-hit branches (mcsim.py:396-397, :504-506):
 * ", " separators, `NaN` / `Infinity` / `-Infinity` tokens like Python's `json.dump` (mcsim.py:457-459, :501), shortest
 * round-trip digits (every value reads back to the identical double), integral values written with ".0".  Formatted by
 * `nthreads` host threads (<= 0: all).  rc_json_bound_f64 returns an upper bound of the text size in bytes;
 * rc_json_encode_f64 needs `cap` >= that bound and returns the number of bytes written (no terminator), or a
 * negative RC_E* code; rc_json_write_f64 streams the same text to the open file descriptor `fd` (at its current
 * offset) in blocks, without materialising it. */
long long rc_json_bound_f64(int ndim, const long long* shape);
long long rc_json_encode_f64(const double* data, int ndim, const long long* shape, char* out, long long cap,
                             int nthreads);
long long rc_json_write_f64(int fd, const double* data, int ndim, const long long* shape, int nthreads);

/* Process-wide default kernel of rc_mc_fidelity_f64 (initially RC_KERNEL_AUTO).
 *
 * Threading: the blocking entry points serialise on an internal PER-DEVICE lock (calls on different devices run
 * concurrently) and may be called from any thread; the *_async entry points only enqueue work on the caller's stream and
 * keep no shared state (the long-row sort allocates its workspace in stream order, per call); rc_last_error() is
 * thread-local. */
int rc_set_fidelity_kernel(int kernel);

#ifdef __cplusplus
}
#endif
#endif /* ROBCHAR_HIP_H */
