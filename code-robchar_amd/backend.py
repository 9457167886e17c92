"""Host-side wrappers over the C ABI (include/robchar_hip.h): NumPy arrays or torch CUDA tensors in,
same kind out.  One call = one sigma_sim level = C x K evaluations of the reference's
`evaluate_noisy_fidelity(x, ham_noisy=True)` (noise_model.py:98-109), or the per-controller metric
reductions of mcsim.py:144-183 / :480-500.

NumPy path : blocking `rc_mc_fidelity_f64` / `rc_reduce_f64` (host buffers staged by the library).
torch path : `*_async` entry points on torch's CURRENT stream with device pointers - nothing is copied and
             nothing synchronises; torch is used only as the owner of device memory and streams.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

Q_THRESHOLDS = (0.95, 0.98)          # mcsim.py:179-180


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def compute_device():
    """torch.device of the GPU this process computes on: torch's CURRENT device (one process per GPU sets it with
    `torch.cuda.set_device(LOCAL_RANK)`); raises when no GPU is visible (there is no CPU path)."""
    _lib.require_gpu()
    import torch
    return torch.device("cuda", torch.cuda.current_device())


def device_index(device=None) -> int:
    """Device ordinal for the C ABI: an explicit int / torch.device, else torch's current device (0 before torch has
    been imported by anybody: the plain-ctypes use of the library)."""
    if device is None:
        import sys
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            return int(torch.cuda.current_device())
        return 0
    if isinstance(device, int):
        return device
    return int(device.index or 0)


# metric rows of a slab packed into one (15, R) tensor: rim1[3] std[3] min[3] q[3][2]  (variant order: centre, upper, lower)
PACKED_ROWS = 15


def packed_views(packed):
    """The (15, R) packed metric tensor as the dict of views `reduce_metrics(out=...)` fills."""
    R = packed.shape[1]
    return {"rim1": packed[0:3], "std": packed[3:6], "min": packed[6:9], "q": packed[9:15].view(3, 2, R)}


def reduce_packed(fid2d, dkw_eps: float, out=None, overlapped: bool = True):
    """`reduce_metrics` of an (R, K) device slab with the reference's two thresholds into ONE (15, R) tensor - the
    only thing a metrics-only caller has to move to the host (120 bytes per controller row).  `overlapped`: see
    `reduce_metrics`."""
    import torch
    R = int(fid2d.shape[0])
    if out is None:
        out = torch.empty((PACKED_ROWS, R), dtype=torch.float64, device=fid2d.device)
    if R:
        reduce_metrics(fid2d, dkw_eps=dkw_eps, out=packed_views(out), overlapped=overlapped)
    return out


def _np_f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {a.shape}")
    return a


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def _small(vec, n, name):
    if vec is None:
        return None
    return _np_f64(vec, (n,), name)


def _check_geometry(nspin, inspin, outspin):
    if not (2 <= int(nspin) <= 32):
        raise ValueError("Nspin must be in [2, 32]")
    if not (0 <= int(inspin) < nspin and 0 <= int(outspin) < nspin):
        raise ValueError("inspin/outspin out of range")


def mc_fidelity(controllers, draws, nspin: int, inspin: int, outspin: int, h0_diag=None, h0_offdiag=None,
                ring: bool = False, device=None, kernel: str = "auto", out=None):
    """Fidelities |<out| exp(-i T H) |in>|^2 for C controllers x K perturbations.

    controllers (C, N+1); draws (C, K, N, 3) already scaled by sigma -> (C, K).  draws of shape (1, K, N, 3)
    with C > 1 controllers = ONE set of K perturbations applied to every controller (the fixed Hamiltonian sets of
    the reference's optimiser-side objective, qnewton.py:122-137, :426-444).
    NumPy inputs -> NumPy output (blocking).  torch CUDA tensors -> torch tensor on the same device,
    enqueued on the current stream (asynchronous).
    """
    _check_geometry(nspin, inspin, outspin)
    lib = _lib.load()
    _lib.require_gpu()
    device = device_index(device)
    h0d = _small(h0_diag, nspin, "h0_diag")
    h0o = _small(h0_offdiag, nspin - 1, "h0_offdiag")
    kid = _lib.KERNELS[kernel]
    n_ctrl_rows = int(controllers.shape[0]) if hasattr(controllers, "shape") else len(controllers)
    shared = int(draws.shape[0]) == 1 and n_ctrl_rows > 1       # one draw set for every controller
    if shared and not _is_torch(draws):
        import torch                                            # the shared form exists on the enqueue path only
        dev_t = torch.device("cuda", device)
        res_t = mc_fidelity(torch.as_tensor(np.ascontiguousarray(controllers, dtype=np.float64)).to(dev_t),
                            torch.as_tensor(np.ascontiguousarray(draws, dtype=np.float64)).to(dev_t), nspin, inspin,
                            outspin, h0_diag=h0_diag, h0_offdiag=h0_offdiag, ring=ring, kernel=kernel)
        res_np = res_t.cpu().numpy()
        if out is not None:
            out[...] = res_np
            return out
        return res_np
    if _is_torch(draws):
        import torch
        if not (draws.is_cuda and draws.dtype == torch.float64 and draws.is_contiguous()):
            raise ValueError("draws must be a contiguous float64 CUDA tensor")
        C, K = (n_ctrl_rows if shared else int(draws.shape[0])), int(draws.shape[1])
        if tuple(draws.shape) != ((1 if shared else C), K, nspin, 3):
            raise ValueError(f"draws: expected (C, K, {nspin}, 3) or (1, K, {nspin}, 3), got {tuple(draws.shape)}")
        dev = draws.device
        ctrl = controllers if _is_torch(controllers) else torch.as_tensor(np.asarray(controllers, dtype=np.float64))
        ctrl = ctrl.to(device=dev, dtype=torch.float64).contiguous()
        if tuple(ctrl.shape) != (C, nspin + 1):
            raise ValueError(f"controllers: expected ({C}, {nspin + 1}), got {tuple(ctrl.shape)}")
        if out is None:
            out = torch.empty((C, K), dtype=torch.float64, device=dev)
        elif not (out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and tuple(out.shape) == (C, K)):
            raise ValueError("out must be a contiguous float64 CUDA tensor of shape (C, K)")
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.rc_mc_fidelity_ex_f64_async(
            dev.index or 0, ctypes.c_void_p(stream), kid, nspin, inspin, outspin, _ptr(h0d), _ptr(h0o),
            int(bool(ring)), ctypes.c_void_p(ctrl.data_ptr()), ctypes.c_void_p(draws.data_ptr()),
            0 if shared else K * nspin * 3, C, K, ctypes.c_void_p(out.data_ptr())))
        return out
    draws = np.ascontiguousarray(draws, dtype=np.float64)
    if draws.ndim != 4 or draws.shape[2:] != (nspin, 3):
        raise ValueError(f"draws: expected (C, K, {nspin}, 3), got {draws.shape}")
    C, K = draws.shape[:2]
    ctrl = _np_f64(controllers, (C, nspin + 1), "controllers")
    res = np.empty((C, K), dtype=np.float64) if out is None else out
    _lib.check(lib.rc_mc_fidelity_kernel_f64(device, kid, nspin, inspin, outspin, _ptr(h0d), _ptr(h0o), int(bool(ring)),
                                             _ptr(ctrl), _ptr(draws), C, K, _ptr(res)))
    return res


def reduce_metrics(fid, q_thresholds=Q_THRESHOLDS, dkw_eps: float = 0.0, want_sorted: bool = False,
                   device=None, out=None, overlapped: bool = True):
    """Per-controller reductions of a (C, K) fidelity slab on the GPU.

    Returns a dict of arrays with a leading variant axis of length 3 (0 centre, 1 " upper" = clip(F-eps),
    2 " lower" = clip(F+eps); mcsim.py:484-485):  rim1 (3,C) = W1 to delta(x-1) = mean infidelity,
    std (3,C), min (3,C), q (3,nq,C) = fraction >= threshold (positive; the reference stores -q), and
    optionally sorted (C,K).  `out` (torch path): dict of preallocated outputs to reuse across calls.
    `overlapped` (torch path; the NumPy path is blocking and always standalone): does the caller run fidelity launches on
    another stream BESIDE this reduction?  False = nothing overlaps it (`RC_REDUCE_STANDALONE`: rows of 8193 .. 10 240 values
    take the dense route, 2x faster alone; what `MCDataSim` passes); True = the latency-bound route that coexists with them
    (a pipelined caller like bench.py's steady state).  Only that row length is affected, and only in the last bits.
    """
    lib = _lib.load()
    _lib.require_gpu()
    thr = np.ascontiguousarray(q_thresholds, dtype=np.float64)
    nq = int(thr.size)
    if _is_torch(fid):
        import torch
        if not (fid.is_cuda and fid.dtype == torch.float64 and fid.is_contiguous() and fid.dim() == 2):
            raise ValueError("fid must be a contiguous float64 CUDA tensor of shape (C, K)")
        C, K = (int(v) for v in fid.shape)
        dev = fid.device
        mk = lambda *s: torch.empty(s, dtype=torch.float64, device=dev)
        if out is not None:          # caller-provided contiguous float64 CUDA buffers (torch path only)
            res = {k: out[k] for k in ("rim1", "std", "min", "q")}
            assert tuple(res["rim1"].shape) == (3, C) and tuple(res["q"].shape) == (3, max(nq, 1), C)
            assert all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float64 for t in res.values())
        else:
            res = {"rim1": mk(3, C), "std": mk(3, C), "min": mk(3, C), "q": mk(3, max(nq, 1), C)}
        srt = mk(C, K) if want_sorted else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.rc_reduce_ex_f64_async(
            dev.index or 0, ctypes.c_void_p(stream), ctypes.c_void_p(fid.data_ptr()), C, K, _ptr(thr), nq,
            float(dkw_eps), ctypes.c_void_p(res["rim1"].data_ptr()), ctypes.c_void_p(res["std"].data_ptr()),
            ctypes.c_void_p(res["min"].data_ptr()), ctypes.c_void_p(res["q"].data_ptr()),
            ctypes.c_void_p(srt.data_ptr()) if srt is not None else None, 0 if overlapped else _lib.RC_REDUCE_STANDALONE))
        res["q"] = res["q"][:, :nq]
        if srt is not None:
            res["sorted"] = srt
        return res
    fid = np.ascontiguousarray(fid, dtype=np.float64)
    if fid.ndim != 2:
        raise ValueError("fid must have shape (C, K)")
    C, K = fid.shape
    device = device_index(device)
    res = {"rim1": np.empty((3, C)), "std": np.empty((3, C)), "min": np.empty((3, C)),
           "q": np.empty((3, nq, C))}
    srt = np.empty((C, K)) if want_sorted else None
    _lib.check(lib.rc_reduce_f64(device, _ptr(fid), C, K, _ptr(thr), nq, float(dkw_eps), _ptr(res["rim1"]),
                                 _ptr(res["std"]), _ptr(res["min"]), _ptr(res["q"]) if nq else None,
                                 _ptr(srt)))
    if srt is not None:
        res["sorted"] = srt
    return res


def rim_p(fid, p: float, device=None):
    """(mean_k (1 - f)^p)^(1/p) per row of a (C, K) slab on the GPU (wd_sortof_fast_implementation.py:147-174)."""
    lib = _lib.load()
    _lib.require_gpu()
    if _is_torch(fid):
        import torch
        if not (fid.is_cuda and fid.dtype == torch.float64 and fid.is_contiguous() and fid.dim() == 2):
            raise ValueError("fid must be a contiguous float64 CUDA tensor of shape (C, K)")
        C, K = (int(v) for v in fid.shape)
        out = torch.empty((C,), dtype=torch.float64, device=fid.device)
        stream = torch.cuda.current_stream(fid.device).cuda_stream
        _lib.check(lib.rc_rim_p_f64_async(fid.device.index or 0, ctypes.c_void_p(stream),
                                          ctypes.c_void_p(fid.data_ptr()), C, K, float(p),
                                          ctypes.c_void_p(out.data_ptr())))
        return out
    fid = np.ascontiguousarray(fid, dtype=np.float64)
    C, K = fid.shape
    out = np.empty((C,))
    _lib.check(lib.rc_rim_p_f64(device_index(device), _ptr(fid), C, K, float(p), _ptr(out)))
    return out


def philox_normal(shape, seed: int, scale: float = 1.0, offset: int = 0, device=None, as_torch: bool = False, out=None):
    """sigma-scaled Gaussian draws from the device's counter-based generator (NOT the reference's RNG stream;
    see include/robchar_hip.h).  Element i of the flattened result is element `offset + i` of stream `seed`.
    Returns a NumPy array, or a torch CUDA tensor when `as_torch`."""
    lib = _lib.load()
    _lib.require_gpu()
    n = int(np.prod(shape))
    if as_torch or out is not None:
        import torch
        if out is not None:
            if not (out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() == n):
                raise ValueError("out must be a contiguous float64 CUDA tensor with prod(shape) elements")
            dev = out.device
        else:
            dev = torch.device("cuda", device_index(device)) if not hasattr(device, "type") else device
            out = torch.empty(tuple(shape), dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.rc_draws_philox_f64_async(dev.index or 0, ctypes.c_void_p(stream), int(seed), int(offset), n,
                                                 float(scale), ctypes.c_void_p(out.data_ptr())))
        return out
    out = np.empty(tuple(shape), dtype=np.float64)
    _lib.check(lib.rc_draws_philox_f64(device_index(device), int(seed), int(offset), n, float(scale), _ptr(out)))
    return out


_warned_log = []


def legacy_device_exact() -> bool:
    """Are the device-continued legacy normals NumPy's bit for bit on this host (`rc_legacy_log_is_host_exact`: the C library's
    log is the routine the device restates)?  When not, the callers below draw on the HOST instead - `legacy` mode means the
    reference's draws, not draws within a few ulp of them - and say so once."""
    ok = bool(_lib.load().rc_legacy_log_is_host_exact())
    if not ok and not _warned_log:
        _warned_log.append(1)
        import warnings
        warnings.warn("this host's libm log() is not the routine librobchar_hip.so restates on the device: legacy draws are "
                      "made by NumPy on the host (bit-identical, slower) instead of being continued on the GPU", RuntimeWarning)
    return ok


def legacy_stream_usable(rng) -> bool:
    """True when `rng` (a noise_function) draws from numpy's GLOBAL legacy stream with nothing but a scale - the
    reference's default generator (noise_model.py:114-115) - AND the device can continue that stream with NumPy's own
    normals (`legacy_device_exact`)."""
    return (getattr(rng, "generator", None) is np.random.normal and set(rng.args) <= {"scale", "loc"}
            and float(rng.args.get("loc", 0.0)) == 0.0 and legacy_device_exact())


def legacy_normal_periods(n_periods: int, period: int, skip: int, scales, out=None, device=None):
    """Continue numpy's global legacy normal stream ON THE GPU (`rc_draws_legacy_f64`): `n_periods` periods of `period`
    draws, the first `skip` of each dropped (burned), the rest scaled by scales[p] -> torch tensor
    (n_periods, period - skip) on the device.  `np.random`'s state afterwards is exactly what the same draws through
    `np.random.normal` would have left (bit for bit), and so are the values wherever `legacy_device_exact()` holds (glibc's
    log restated on the device; a few ulp otherwise)."""
    import torch
    lib = _lib.load()
    _lib.require_gpu()
    scales = np.ascontiguousarray(scales, dtype=np.float64).reshape(-1)
    if scales.size != n_periods:
        raise ValueError("scales must have n_periods entries")
    if out is None:
        dev = torch.device("cuda", device_index(device))
        out = torch.empty((n_periods, period - skip), dtype=torch.float64, device=dev)
    elif not (out.is_cuda and out.dtype == torch.float64 and out.is_contiguous()
              and out.numel() == n_periods * (period - skip)):
        raise ValueError("out must be a contiguous float64 CUDA tensor with n_periods * (period - skip) elements")
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != "MT19937":
        raise ValueError("numpy's global generator is not the legacy MT19937 stream")
    st = _lib.Mt19937State()
    ctypes.memmove(st.key, np.ascontiguousarray(key, dtype=np.uint32).ctypes.data, 624 * 4)
    st.pos, st.has_gauss, st.gauss = int(pos), int(has_gauss), float(cached)
    stream = torch.cuda.current_stream(out.device).cuda_stream
    _lib.check(lib.rc_draws_legacy_f64(out.device.index or 0, ctypes.c_void_p(stream), ctypes.byref(st), int(n_periods),
                                       int(period), int(skip), _ptr(scales), ctypes.c_void_p(out.data_ptr())))
    np.random.set_state(("MT19937", np.frombuffer(st.key, dtype=np.uint32).copy(), int(st.pos), int(st.has_gauss),
                         float(st.gauss)))
    return out


def directional_draws_device(n: int, ndir: int, sigma: float, device=None):
    """`n` samples of `directional_perturbation`'s RNG consumption (per sample `np.random.randint(0, ndir)` then two legacy
    normals scaled by sigma) continued from numpy's global legacy stream ON THE GPU (`rc_directional_draws_legacy_dev`)
    -> (idx int32 [n], ab float64 [n, 2]) torch CUDA tensors.  `np.random`'s state afterwards is what the n Python-level
    draws would have left, bit for bit; indices identical, normals identical where `legacy_device_exact()` holds."""
    import torch
    lib = _lib.load()
    _lib.require_gpu()
    dev = torch.device("cuda", device_index(device))
    idx = torch.empty((n,), dtype=torch.int32, device=dev)
    ab = torch.empty((n, 2), dtype=torch.float64, device=dev)
    if n == 0:
        return idx, ab
    name, key, pos, has_gauss, cached = np.random.get_state()
    if name != "MT19937":
        raise ValueError("numpy's global generator is not the legacy MT19937 stream")
    st = _lib.Mt19937State()
    ctypes.memmove(st.key, np.ascontiguousarray(key, dtype=np.uint32).ctypes.data, 624 * 4)
    st.pos, st.has_gauss, st.gauss = int(pos), int(has_gauss), float(cached)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.rc_directional_draws_legacy_dev(dev.index or 0, ctypes.c_void_p(stream), ctypes.byref(st), int(n), int(ndir),
                                                   float(sigma), ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(ab.data_ptr())))
    np.random.set_state(("MT19937", np.frombuffer(st.key, dtype=np.uint32).copy(), int(st.pos), int(st.has_gauss),
                         float(st.gauss)))
    return idx, ab


def philox_fused_supported(nspin: int, ring: bool = False, kernel: str = "auto") -> bool:
    """Can `mc_fidelity_philox` take this geometry (chain, N <= 16, eigenvalue-only kernels)?"""
    return (not ring) and 2 <= nspin <= 16 and kernel in ("auto", "tridiag_adj")


def philox_fused_pays(nspin: int, inspin: int, outspin: int) -> bool:
    """Is the fused kernel the faster route for this geometry?  Measured at every size (profiles/r04_philox_fused_sweep.txt):
    0.70 - 0.86 of the two-kernel route's time up to N = 13 and for end-to-end pairs at N = 14; beyond that its instantiations
    run one wave per SIMD and generating the draw tensor first is 7 % faster.  (The results are bit-identical either way.)"""
    return bool(_lib.load().rc_philox_fused_pays(int(nspin), int(inspin), int(outspin)))


def mc_fidelity_philox(controllers, n_draws: int, nspin: int, inspin: int, outspin: int, seed: int, offset: int = 0,
                       sigma=0.05, h0_diag=None, h0_offdiag=None, kernel: str = "auto", out=None):
    """Fidelities with the counter-based draws generated INSIDE the kernel (`rc_mc_fidelity_philox_f64_async`): controllers
    (C, N+1) torch CUDA tensor -> (C, K) torch tensor on the same device, enqueued on the current stream; bit-identical to
    `mc_fidelity(controllers, philox_normal((C, K, N, 3), seed, scale=sigma, offset=offset))` without that tensor.
    `sigma`: a float, or a (C,) float64 CUDA tensor (one scale per controller row)."""
    import torch
    _check_geometry(nspin, inspin, outspin)
    lib = _lib.load()
    _lib.require_gpu()
    if not (_is_torch(controllers) and controllers.is_cuda):
        raise ValueError("controllers must be a torch CUDA tensor")
    dev = controllers.device
    ctrl = controllers.to(dtype=torch.float64).contiguous()
    C, K = int(ctrl.shape[0]), int(n_draws)
    if tuple(ctrl.shape) != (C, nspin + 1):
        raise ValueError(f"controllers: expected ({C}, {nspin + 1})")
    if out is None:
        out = torch.empty((C, K), dtype=torch.float64, device=dev)
    elif not (out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and tuple(out.shape) == (C, K)):
        raise ValueError("out must be a contiguous float64 CUDA tensor of shape (C, K)")
    rows = None
    if _is_torch(sigma):
        rows = sigma.to(device=dev, dtype=torch.float64).contiguous()
        if tuple(rows.shape) != (C,):
            raise ValueError("sigma: a float or a (C,) tensor")
    h0d = _small(h0_diag, nspin, "h0_diag")
    h0o = _small(h0_offdiag, nspin - 1, "h0_offdiag")
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.rc_mc_fidelity_philox_f64_async(dev.index or 0, ctypes.c_void_p(stream), _lib.KERNELS[kernel], nspin, inspin,
                                                   outspin, _ptr(h0d), _ptr(h0o), ctypes.c_void_p(ctrl.data_ptr()),
                                                   int(seed) & (2 ** 64 - 1), int(offset), 0.0 if rows is not None else float(sigma),
                                                   ctypes.c_void_p(rows.data_ptr()) if rows is not None else None, C, K,
                                                   ctypes.c_void_p(out.data_ptr())))
    return out


def mc_fidelity_directional(controllers, idx, ab, nspin: int, inspin: int, outspin: int, n_draws: int, h0_diag=None,
                            h0_offdiag=None, out=None):
    """Fidelities of `directional_perturbation` samples straight from (direction index, two normals) per sample
    (`rc_mc_fidelity_directional_f64_async`): controllers (C, N+1) torch CUDA tensor, idx (C*K,) int32, ab (C*K, 2) float64
    - what `directional_draws_device` returns - -> (C, K) torch tensor on the same device, enqueued on the current stream.
    Chain topology, N <= 12 (the library answers RC_ENOSUP otherwise: `RobCharHipError`)."""
    import torch
    _check_geometry(nspin, inspin, outspin)
    lib = _lib.load()
    _lib.require_gpu()
    if not (_is_torch(idx) and _is_torch(ab) and _is_torch(controllers)):
        raise ValueError("controllers, idx and ab must be torch tensors (idx / ab: what directional_draws_device returns)")
    dev = idx.device
    C, K = int(controllers.shape[0]), int(n_draws)
    if not (idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == C * K):
        raise ValueError("idx must be a contiguous int32 CUDA tensor with C * K entries")
    if not (ab.is_cuda and ab.dtype == torch.float64 and ab.is_contiguous() and tuple(ab.shape) == (C * K, 2)
            and ab.device == dev):
        raise ValueError("ab must be a contiguous float64 CUDA tensor of shape (C * K, 2) on idx's device")
    if controllers.is_cuda and controllers.device != dev:
        raise ValueError("controllers live on another GPU than idx / ab")
    ctrl = controllers.to(device=dev, dtype=torch.float64).contiguous()
    if tuple(ctrl.shape) != (C, nspin + 1):
        raise ValueError(f"controllers: expected ({C}, {nspin + 1})")
    if out is None:
        out = torch.empty((C, K), dtype=torch.float64, device=dev)
    elif not (_is_torch(out) and out.is_cuda and out.dtype == torch.float64 and out.is_contiguous()
              and tuple(out.shape) == (C, K) and out.device == dev):
        raise ValueError("out must be a contiguous float64 CUDA tensor of shape (C, K) on idx's device")
    h0d = _small(h0_diag, nspin, "h0_diag")
    h0o = _small(h0_offdiag, nspin - 1, "h0_offdiag")
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.rc_mc_fidelity_directional_f64_async(
        dev.index or 0, ctypes.c_void_p(stream), nspin, inspin, outspin, _ptr(h0d), _ptr(h0o), 0,
        ctypes.c_void_p(ctrl.data_ptr()), ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(ab.data_ptr()), C, K,
        ctypes.c_void_p(out.data_ptr())))
    return out


def mc_fidelity_nonhermitian(controllers, draws, diag_imag, nspin: int, inspin: int, outspin: int, h0_diag=None,
                             h0_offdiag=None, ring: bool = False, device=None):
    """Fidelities for a Hamiltonian with an IMAGINARY diagonal perturbation: H = HH + Z(draws) + diag(x) +
    1j*diag(diag_imag) - the dense Pade-expm kernel (`rc_mc_fidelity_nh_f64_async`).  controllers (C, N+1), draws
    (C, K, N, 3), diag_imag (C, K, N) or None -> (C, K).  NumPy in / NumPy out, torch CUDA in / torch out."""
    _check_geometry(nspin, inspin, outspin)
    lib = _lib.load()
    _lib.require_gpu()
    import torch
    as_numpy = not _is_torch(draws)
    dev = torch.device("cuda", device_index(device)) if as_numpy else draws.device
    to_dev = lambda a: None if a is None else (a if _is_torch(a) else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))).to(device=dev, dtype=torch.float64).contiguous()
    d, c, g = to_dev(draws), to_dev(controllers), to_dev(diag_imag)
    C, K = int(d.shape[0]), int(d.shape[1])
    if tuple(d.shape) != (C, K, nspin, 3) or tuple(c.shape) != (C, nspin + 1) or (g is not None and tuple(g.shape) != (C, K, nspin)):
        raise ValueError("shapes: controllers (C, N+1), draws (C, K, N, 3), diag_imag (C, K, N)")
    out = torch.empty((C, K), dtype=torch.float64, device=dev)
    h0d = _small(h0_diag, nspin, "h0_diag")
    h0o = _small(h0_offdiag, nspin - 1, "h0_offdiag")
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.rc_mc_fidelity_nh_f64_async(dev.index or 0, ctypes.c_void_p(stream), nspin, inspin, outspin, _ptr(h0d),
                                               _ptr(h0o), int(bool(ring)), ctypes.c_void_p(c.data_ptr()),
                                               ctypes.c_void_p(d.data_ptr()),
                                               ctypes.c_void_p(g.data_ptr()) if g is not None else None, C, K,
                                               ctypes.c_void_p(out.data_ptr())))
    return out.cpu().numpy() if as_numpy else out


def release_stream(stream=None, device=None) -> None:
    """Hand back what the library keeps for `stream` (a torch.cuda.Stream; default: the current one) - the repair list of
    the ring-topology route, 10 bytes per sample of the largest ring launch the stream ran (`rc_release_stream`; freed in
    stream order behind the stream's last launch).  Call it before retiring a side stream that ran ring launches; the
    library also caps what unreleased streams can hold (16 buffers per device, least recently used evicted)."""
    import torch
    lib = _lib.load()
    _lib.require_gpu()
    if stream is None:
        stream = torch.cuda.current_stream(torch.device("cuda", device_index(device)))
    _lib.check(lib.rc_release_stream(int(stream.device.index or 0), ctypes.c_void_p(stream.cuda_stream)))


class ring_stream:
    """Context manager: a side stream for ring-topology launches whose library-side buffer is released on exit.

        with backend.ring_stream(device) as st:            # torch.cuda.Stream, current inside the block
            backend.mc_fidelity(ctrl, draws, N, a, b, ring=True, out=fid)
    """

    def __init__(self, device=None, priority: int = 0):
        import torch
        self._torch = torch
        self.stream = torch.cuda.Stream(torch.device("cuda", device_index(device)), priority=priority)
        self._ctx = None

    def __enter__(self):
        self._ctx = self._torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        return self.stream

    def __exit__(self, *exc):
        try:
            release_stream(self.stream)
        finally:
            self._ctx.__exit__(*exc)
        return False


def general_path_tiles(device=None, reset: bool = False) -> int:
    """Diagnostic: 64-sample tiles that left the chain kernels' fast path since the last reset (0 on healthy
    workloads; each such tile is recomputed by the much slower general per-sample routine)."""
    lib = _lib.load()
    _lib.require_gpu()
    v = lib.rc_stats_general_tiles(device_index(device), int(bool(reset)))
    if v < 0:
        _lib.check(int(v))
    return int(v)


def polish_tiles(device=None, reset: bool = False) -> int:
    """Diagnostic: 64-sample tiles of the mixed-precision eigenvalue path that needed more than its one fp64 step since
    the last reset (a close eigenvalue pair somewhere in the tile; the tile keeps stepping, still on the fast path)."""
    lib = _lib.load()
    _lib.require_gpu()
    v = lib.rc_stats_polish_tiles(device_index(device), int(bool(reset)))
    if v < 0:
        _lib.check(int(v))
    return int(v)


def _devices_arg(devices):
    if devices is None:
        n = _lib.require_gpu()
        devices = list(range(n))
    devices = [int(d) for d in devices]
    return (ctypes.c_int * len(devices))(*devices), len(devices)


def mc_fidelity_sharded(controllers, draws, nspin: int, inspin: int, outspin: int, devices=None, h0_diag=None,
                        h0_offdiag=None, ring: bool = False, kernel: str = "auto"):
    """`mc_fidelity` over several GPUs of this process (`rc_mc_fidelity_sharded_f64`): host arrays in, the full
    (C, K) host array out; controllers are split into contiguous blocks, one per device."""
    _check_geometry(nspin, inspin, outspin)
    lib = _lib.load()
    dev_arr, ndev = _devices_arg(devices)
    draws = np.ascontiguousarray(draws, dtype=np.float64)
    if draws.ndim != 4 or draws.shape[2:] != (nspin, 3):
        raise ValueError(f"draws: expected (C, K, {nspin}, 3), got {draws.shape}")
    C, K = draws.shape[:2]
    ctrl = _np_f64(controllers, (C, nspin + 1), "controllers")
    res = np.empty((C, K))
    _lib.check(lib.rc_mc_fidelity_sharded_f64(ndev, dev_arr, _lib.KERNELS[kernel], nspin, inspin, outspin,
                                              _ptr(_small(h0_diag, nspin, "h0_diag")),
                                              _ptr(_small(h0_offdiag, nspin - 1, "h0_offdiag")), int(bool(ring)),
                                              _ptr(ctrl), _ptr(draws), C, K, _ptr(res)))
    return res


def mc_metrics_sharded(controllers, n_draws: int, nspin: int, inspin: int, outspin: int, draws=None, seed: int = 0,
                       offset: int = 0, sigma: float = 0.0, devices=None, h0_diag=None, h0_offdiag=None,
                       ring: bool = False, kernel: str = "auto", q_thresholds=Q_THRESHOLDS, dkw_eps: float = 0.0,
                       want_fid: bool = False):
    """Fidelity + per-controller metrics over several GPUs of this process (`rc_mc_metrics_sharded_f64`); only the
    metric rows (and the fidelities when `want_fid`) come back to the host.  `draws=None`: counter-based draws
    generated on the devices (stream `seed`, first element `offset`, scaled by `sigma`)."""
    _check_geometry(nspin, inspin, outspin)
    lib = _lib.load()
    dev_arr, ndev = _devices_arg(devices)
    ctrl = np.ascontiguousarray(controllers, dtype=np.float64)
    C, K = ctrl.shape[0], int(n_draws)
    if ctrl.shape != (C, nspin + 1):
        raise ValueError(f"controllers: expected (C, {nspin + 1})")
    if draws is not None:
        draws = _np_f64(draws, (C, K, nspin, 3), "draws")
    thr = np.ascontiguousarray(q_thresholds, dtype=np.float64)
    nq = int(thr.size)
    res = {"rim1": np.empty((3, C)), "std": np.empty((3, C)), "min": np.empty((3, C)), "q": np.empty((3, nq, C))}
    fid = np.empty((C, K)) if want_fid else None
    _lib.check(lib.rc_mc_metrics_sharded_f64(ndev, dev_arr, _lib.KERNELS[kernel], nspin, inspin, outspin,
                                             _ptr(_small(h0_diag, nspin, "h0_diag")),
                                             _ptr(_small(h0_offdiag, nspin - 1, "h0_offdiag")), int(bool(ring)),
                                             _ptr(ctrl), _ptr(draws), int(seed), int(offset), float(sigma), C, K,
                                             _ptr(thr), nq, float(dkw_eps), _ptr(res["rim1"]), _ptr(res["std"]),
                                             _ptr(res["min"]), _ptr(res["q"]) if nq else None, _ptr(fid)))
    if want_fid:
        res["fid"] = fid
    return res


class RcclComm:
    """`rc_comm_init` / `rc_comm_destroy` as a context manager: ONE process, one RCCL communicator per listed GPU - the
    exchange step of the path (an all-gather over xGMI) from the C ABI, without torch.distributed.

        with backend.RcclComm(devices=[0, 1, 2, 3]) as comm:
            res = backend.mc_metrics_gathered(comm, controllers, K, N, a, b, seed=7, sigma=0.05)
    """

    def __init__(self, devices=None):
        # PyTorch first: the library resolves RCCL from what the process already carries (torch ships its own librccl.so /
        # librocm_smi64.so) and only otherwise opens the system's; a process that ends up with BOTH pairs is one ROCm
        # does not expect (robchar_hip.hip: rccl_api)
        import torch  # noqa: F401
        self.lib = _lib.load()
        dev_arr, ndev = _devices_arg(devices)
        self.devices = [int(d) for d in dev_arr]           # rank r of the communicator = GPU devices[r]
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.rc_comm_init(ndev, dev_arr, ctypes.byref(self.handle)))
        self.ndev = int(self.lib.rc_comm_size(self.handle))

    def close(self):
        if self.handle:
            _lib.check(self.lib.rc_comm_destroy(self.handle))
            self.handle = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def mc_metrics_gathered(comm: "RcclComm", controllers, n_draws: int, nspin: int, inspin: int, outspin: int, draws=None,
                        seed: int = 0, offset: int = 0, sigma: float = 0.0, h0_diag=None, h0_offdiag=None, ring: bool = False,
                        kernel: str = "auto", q_thresholds=Q_THRESHOLDS, dkw_eps: float = 0.0, want_fid: bool = False):
    """Fidelity + per-controller metrics over the communicator's GPUs with the exchange step ON the devices
    (`rc_mc_metrics_gathered_f64`: every device all-gathers the metric rows - and the fidelity slabs when `want_fid` - over
    RCCL); returns the host copies (from the first device) in `mc_metrics_sharded`'s format.  The device-resident gathered
    buffers are the library's own here (a C caller can pass its own and keep them)."""
    _check_geometry(nspin, inspin, outspin)
    lib = _lib.load()
    ctrl = np.ascontiguousarray(controllers, dtype=np.float64)
    C, K = ctrl.shape[0], int(n_draws)
    if ctrl.shape != (C, nspin + 1):
        raise ValueError(f"controllers: expected (C, {nspin + 1})")
    if draws is not None:
        draws = _np_f64(draws, (C, K, nspin, 3), "draws")
    thr = np.ascontiguousarray(q_thresholds, dtype=np.float64)
    nq = int(thr.size)
    table = np.empty((9 + 3 * nq, C))
    fid = np.empty((C, K)) if want_fid else None
    fid_dev = None
    keep = []
    if want_fid and comm.ndev > 1:                    # a fidelity gather needs a receive buffer on every rank
        import torch
        cmax = -(-C // comm.ndev)
        for d in comm.devices:                        # rank r's receive buffer lives on GPU devices[r]
            keep.append(torch.empty((comm.ndev * cmax * K,), dtype=torch.float64, device=torch.device("cuda", d)))
        fid_dev = (ctypes.c_void_p * comm.ndev)(*[t.data_ptr() for t in keep])
    _lib.check(lib.rc_mc_metrics_gathered_f64(comm.handle, _lib.KERNELS[kernel], nspin, inspin, outspin,
                                              _ptr(_small(h0_diag, nspin, "h0_diag")), _ptr(_small(h0_offdiag, nspin - 1, "h0_offdiag")),
                                              int(bool(ring)), _ptr(ctrl), _ptr(draws), int(seed), int(offset), float(sigma), C, K,
                                              _ptr(thr), nq, float(dkw_eps), None, fid_dev, _ptr(table), _ptr(fid)))
    res = {"rim1": table[0:3].copy(), "std": table[3:6].copy(), "min": table[6:9].copy(),
           "q": table[9:].reshape(3, nq, C).copy()}
    if want_fid:
        res["fid"] = fid
    return res
