"""ctypes binding of librobchar_hip.so (the C ABI of include/robchar_hip.h).

There is deliberately NO fallback: if the shared library has not been built, or no MI355X is visible,
every compute entry point raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from this package.)
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ROBCHAR_HIP_LIB selects another build of the same library (kernel-tuning experiments); default = in-tree build
LIB_PATH = os.environ.get("ROBCHAR_HIP_LIB") or os.path.join(_HERE, "csrc", "librobchar_hip.so")

# every symbol include/robchar_hip.h declares
EXPORTS = (
    "rc_version", "rc_device_count", "rc_last_error", "rc_set_fidelity_kernel", "rc_stats_general_tiles", "rc_stats_polish_tiles",
    "rc_mc_fidelity_f64", "rc_mc_fidelity_kernel_f64", "rc_mc_fidelity_f64_async",
    "rc_mc_fidelity_ex_f64_async", "rc_mc_fidelity_nh_f64_async", "rc_reduce_f64", "rc_reduce_f64_async",
    "rc_rim_p_f64", "rc_rim_p_f64_async", "rc_draws_philox_f64", "rc_draws_philox_f64_async",
    "rc_json_bound_f64", "rc_json_encode_f64", "rc_json_write_f64",
    "rc_mc_fidelity_sharded_f64", "rc_mc_metrics_sharded_f64", "rc_draws_legacy_f64", "rc_directional_draws_legacy",
    "rc_directional_draws_legacy_dev", "rc_reserve_ring", "rc_release_stream",
    "rc_mc_fidelity_directional_f64_async", "rc_mc_fidelity_philox_f64_async", "rc_build_flags", "rc_philox_fused_pays",
    "rc_reduce_ex_f64_async", "rc_legacy_log_is_host_exact",
    "rc_comm_init", "rc_comm_size", "rc_comm_destroy", "rc_mc_metrics_gathered_f64",
)

RC_KERNEL_AUTO, RC_KERNEL_TRIDIAG_QL, RC_KERNEL_JACOBI, RC_KERNEL_TRIDIAG_ADJ, RC_KERNEL_EXPM, RC_KERNEL_RING_HH = 0, 1, 2, 3, 4, 5
RC_REDUCE_STANDALONE = 1
KERNELS = {"auto": RC_KERNEL_AUTO, "tridiag_ql": RC_KERNEL_TRIDIAG_QL, "jacobi": RC_KERNEL_JACOBI,
           "tridiag_adj": RC_KERNEL_TRIDIAG_ADJ, "expm": RC_KERNEL_EXPM, "ring_hh": RC_KERNEL_RING_HH}


class Mt19937State(ctypes.Structure):
    """`rc_mt19937_state` of include/robchar_hip.h: numpy's `RandomState.get_state()` tuple in C layout."""
    _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int), ("has_gauss", ctypes.c_int),
                ("gauss", ctypes.c_double)]


class RobCharHipError(RuntimeError):
    """A call into librobchar_hip.so failed (message from rc_last_error())."""


_lib = None


def _bind_to_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (soname
    libamdhip64.so.7, same as /opt/rocm's).  If librobchar_hip.so pulled in /opt/rocm's copy first, a later
    `import torch` would load a SECOND runtime that cannot see the GPU, and stream handles / device pointers
    would not be interchangeable.  Pre-loading torch's copy (when torch is installed) makes the dynamic
    linker resolve our NEEDED libamdhip64.so.7 to it, whichever of the two is imported first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def _refuse_experiment_build(lib):
    """A variant build (ROBCHAR_HIP_LIB, scripts/build_variant.sh) whose timing-experiment switches knowingly return wrong
    fidelities must never be taken for the product: refused unless the caller says so; any other non-default switch warns."""
    if not hasattr(lib, "rc_build_flags"):
        raise RobCharHipError(f"{LIB_PATH} predates ABI 6 (no rc_build_flags): rebuild it with `make -C code-robchar_amd/csrc`")
    lib.rc_build_flags.argtypes = []
    lib.rc_build_flags.restype = ctypes.c_int
    flags = int(lib.rc_build_flags())
    if flags & BUILD_WRONG_RESULTS_MASK and os.environ.get("ROBCHAR_ALLOW_EXPERIMENT_LIB") != "1":
        raise RobCharHipError(
            f"{LIB_PATH} is a timing-experiment build (rc_build_flags() = {flags}: {', '.join(build_flag_names(flags))}); its "
            "results are knowingly wrong for some samples.  Unset ROBCHAR_HIP_LIB, or set ROBCHAR_ALLOW_EXPERIMENT_LIB=1 for a timing run")
    if flags:
        import warnings
        warnings.warn(f"librobchar_hip.so: non-default build switches {build_flag_names(flags)} ({LIB_PATH})", RuntimeWarning)


def load():
    """Load the shared library once and declare the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RobCharHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C code-robchar_amd/csrc` (there is no CPU fallback)")
    _bind_to_torch_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    _refuse_experiment_build(lib)
    dp, vp, ll, i = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    dbl = ctypes.c_double
    lib.rc_version.restype = i
    lib.rc_device_count.restype = i
    lib.rc_last_error.restype = ctypes.c_char_p
    lib.rc_set_fidelity_kernel.argtypes = [i]
    lib.rc_mc_fidelity_f64.argtypes = [i, i, i, i, dp, dp, i, dp, dp, ll, ll, dp]
    lib.rc_mc_fidelity_kernel_f64.argtypes = [i, i, i, i, i, dp, dp, i, dp, dp, ll, ll, dp]
    lib.rc_mc_fidelity_f64_async.argtypes = [i, vp, i, i, i, i, dp, dp, i, dp, dp, ll, ll, dp]
    lib.rc_mc_fidelity_ex_f64_async.argtypes = [i, vp, i, i, i, i, dp, dp, i, dp, dp, ll, ll, ll, dp]
    lib.rc_mc_fidelity_nh_f64_async.argtypes = [i, vp, i, i, i, dp, dp, i, dp, dp, dp, ll, ll, dp]
    lib.rc_reduce_f64.argtypes = [i, dp, ll, ll, dp, i, dbl, dp, dp, dp, dp, dp]
    lib.rc_reduce_f64_async.argtypes = [i, vp, dp, ll, ll, dp, i, dbl, dp, dp, dp, dp, dp]
    lib.rc_rim_p_f64.argtypes = [i, dp, ll, ll, dbl, dp]
    lib.rc_rim_p_f64_async.argtypes = [i, vp, dp, ll, ll, dbl, dp]
    ull = ctypes.c_ulonglong
    lib.rc_draws_philox_f64.argtypes = [i, ull, ull, ll, dbl, dp]
    lib.rc_draws_philox_f64_async.argtypes = [i, vp, ull, ull, ll, dbl, dp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("rc_last_error",):
            fn.restype = i
    lib.rc_stats_general_tiles.argtypes = [i, i]
    lib.rc_stats_general_tiles.restype = ll
    lib.rc_stats_polish_tiles.argtypes = [i, i]
    lib.rc_stats_polish_tiles.restype = ll
    lib.rc_mc_fidelity_sharded_f64.argtypes = [i, dp, i, i, i, i, dp, dp, i, dp, dp, ll, ll, dp]
    lib.rc_mc_metrics_sharded_f64.argtypes = [i, dp, i, i, i, i, dp, dp, i, dp, dp, ull, ull, dbl, ll, ll, dp, i, dbl,
                                              dp, dp, dp, dp, dp]
    lib.rc_draws_legacy_f64.argtypes = [i, vp, dp, ll, ll, ll, dp, dp]
    lib.rc_directional_draws_legacy.argtypes = [dp, ll, i, dbl, dp, dp]
    lib.rc_directional_draws_legacy_dev.argtypes = [i, vp, dp, ll, i, dbl, dp, dp]
    lib.rc_reserve_ring.argtypes = [i, vp, ll]
    lib.rc_release_stream.argtypes = [i, vp]
    lib.rc_mc_fidelity_directional_f64_async.argtypes = [i, vp, i, i, i, dp, dp, i, dp, dp, dp, ll, ll, dp]
    lib.rc_mc_fidelity_philox_f64_async.argtypes = [i, vp, i, i, i, i, dp, dp, dp, ull, ull, dbl, dp, ll, ll, dp]
    lib.rc_json_bound_f64.argtypes = [i, dp]
    lib.rc_json_bound_f64.restype = ll
    lib.rc_json_encode_f64.argtypes = [dp, i, dp, dp, ll, i]
    lib.rc_json_encode_f64.restype = ll
    lib.rc_json_write_f64.argtypes = [i, dp, i, dp, i]
    lib.rc_json_write_f64.restype = ll
    lib.rc_philox_fused_pays.argtypes = [i, i, i]
    lib.rc_legacy_log_is_host_exact.argtypes = []
    lib.rc_comm_init.argtypes = [i, dp, ctypes.POINTER(vp)]
    lib.rc_comm_size.argtypes = [vp]
    lib.rc_comm_destroy.argtypes = [vp]
    lib.rc_mc_metrics_gathered_f64.argtypes = [vp, i, i, i, i, dp, dp, i, dp, dp, ull, ull, dbl, ll, ll, dp, i, dbl, dp, dp, dp, dp]
    lib.rc_reduce_ex_f64_async.argtypes = [i, vp, dp, ll, ll, dp, i, dbl, dp, dp, dp, dp, dp, i]
    _lib = lib
    return lib


BUILD_FLAGS = {1: "EXPERIMENT_NO_STEPPING", 2: "EXPERIMENT_STEP_NOT_RUN", 4: "EXPERIMENT_FALLBACK_NOT_RUN",
               8: "EXPERIMENT_PHILOX_NOSTORE", 16: "DEV_FEW_N", 32: "STAMPS", 64: "NO_SUM_RULE_GUARD", 128: "NO_KEEP_SETTLED"}
BUILD_WRONG_RESULTS_MASK = 15


def build_flag_names(flags: int):
    return [name for bit, name in BUILD_FLAGS.items() if flags & bit]


def build_flags() -> int:
    """`rc_build_flags()` of the loaded library (0 = the product build)."""
    return int(load().rc_build_flags())


def check(rc: int):
    if rc != 0:
        msg = load().rc_last_error()
        raise RobCharHipError(f"librobchar_hip error {rc}: {msg.decode() if msg else '?'}")


def require_gpu() -> int:
    """Number of visible GPUs; raises when there is none (no silent CPU path)."""
    n = load().rc_device_count()
    if n <= 0:
        raise RobCharHipError("no HIP device visible: the RobChar MC path runs on MI355X only")
    return n
