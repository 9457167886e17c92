"""Robustness metrics of a fidelity sample - API mirror of wd_sortof_fast_implementation.py, evaluated by
the GPU reduction kernel (`rc_reduce_f64`, include/robchar_hip.h).

    wd_from_ideal(fids)      RIM_1 = W_1(F, delta(x-1))       wd...py:82-116 (sorts `fids` IN PLACE, as there)
    wd_from_ideal_zero       1 - RIM_1                        wd...py:119-142
    RIM_p(fids, p)           (mean (1-f)^p)^(1/p)             wd...py:147-174
    compute_dkw_error        sqrt(ln(2/alpha) / 2n)           wd...py:38-39  (scalar host arithmetic)
    dkw_ecdf_bounds          clip(cdf -/+ eps, 0, 1)          wd...py:41-79
plus the row-wise metric table of mcsim.py:144-183 (`metric_table`).
"""
from __future__ import annotations

import math

import numpy as np

from . import backend

METRIC_NAMES = (r'$W(.,\delta(x-1))$', "Q th. 0.95", "Q th. 0.98", "std", "worst case fid")   # mcsim.py:178-183


def _as_fids(fids) -> np.ndarray:
    """Argument coercion + range guard of `check_fidtype` (wd...py:13-30): lists become arrays, scalars
    become 1-vectors, AssertionError if any |f - 1e-8| > 1 (NaN passes)."""
    if not isinstance(fids, np.ndarray):
        fids = np.array(fids) if isinstance(fids, list) else np.array([fids])
    if (np.abs(fids - 1e-8) > 1).any():
        raise AssertionError("illegal fids values - must be in [0,1]")
    return fids


def compute_dkw_error(alpha, nobs):
    return math.sqrt(math.log(2 / alpha) / (2 * nobs))


def dkw_ecdf_bounds(cdf, conf_level: float, visualize: bool = False):
    cdf = _as_fids(cdf)
    eps = compute_dkw_error(1 - conf_level, cdf.shape[-1])
    lower, upper = np.clip(cdf - eps, 0, 1), np.clip(cdf + eps, 0, 1)
    if visualize:
        import matplotlib.pyplot as plt
        plt.figure()
        grid = np.arange(len(cdf)) / len(cdf)
        for data, lab, col in ((cdf, "ecdf", "b"), (lower, "lower", "r"), (upper, "upper", "r")):
            plt.step(data, grid, label=lab, c=col)
        plt.ylabel(r"$Q_F$")
        plt.xlabel(r"$F$")
        plt.legend()
    return lower, upper


def wd_from_ideal(fids, sort_fids: bool = True):
    """1-Wasserstein distance of the sample from delta(x-1) on the GPU.  Like the reference, the caller's
    array is sorted in place when `sort_fids` (a side effect figure scripts rely on)."""
    arr = _as_fids(fids)
    flat = np.ascontiguousarray(arr, dtype=np.float64).reshape(1, -1)
    want_sorted = bool(sort_fids) and arr.ndim == 1 and arr.flags.writeable
    red = backend.reduce_metrics(flat, q_thresholds=(), want_sorted=want_sorted)       # the device row sort takes any K
    if want_sorted:
        arr[...] = red["sorted"][0].astype(arr.dtype, copy=False)
    return float(red["rim1"][0, 0])


def wd_from_ideal_zero(fids, sort_fids: bool = True):
    return 1 - wd_from_ideal(fids, sort_fids)


def RIM_p(fids, p=2):
    if p == 0:
        return 1
    arr = _as_fids(fids)
    if p == 1:
        return wd_from_ideal(np.array(arr, dtype=np.float64), sort_fids=False)
    return float(backend.rim_p(np.asarray(arr, dtype=np.float64).reshape(1, -1), p)[0])


def metric_table(level_tensor, dkw_eps: float = 0.0):
    """The five `.mcm` metrics (mcsim.py:178-183) of a (C, K) slab for the three DKW variants, on the GPU.

    Returns {suffix: {metric_name: list[C]}} with suffix in ("", " upper", " lower") and the reference's sign
    conventions (Q and worst-case negated, mcsim.py:148-176)."""
    red = backend.reduce_metrics(level_tensor, dkw_eps=dkw_eps)
    get = (lambda t: t.cpu().numpy()) if backend._is_torch(red["rim1"]) else (lambda t: t)
    rim, std, mn, q = get(red["rim1"]), get(red["std"]), get(red["min"]), get(red["q"])
    out = {}
    for v, suffix in enumerate(("", " upper", " lower")):
        out[suffix] = {
            METRIC_NAMES[0]: rim[v].tolist(),
            METRIC_NAMES[1]: (-q[v, 0]).tolist(),
            METRIC_NAMES[2]: (-q[v, 1]).tolist(),
            METRIC_NAMES[3]: std[v].tolist(),
            METRIC_NAMES[4]: (-mn[v]).tolist(),
        }
    return out
