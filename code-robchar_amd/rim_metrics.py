"""Robustness metrics of a fidelity sample - API mirror of wd_sortof_fast_implementation.py, evaluated by
the GPU reduction kernel (`rc_reduce_f64`, include/robchar_hip.h).

    wd_from_ideal(fids)      RIM_1 = W_1(F, delta(x-1))       wd...py:82-116 (sorts `fids` IN PLACE, as there)
    wd_from_ideal_zero       1 - RIM_1                        wd...py:119-142
    RIM_p(fids, p)           (mean (1-f)^p)^(1/p)             wd...py:147-174
    compute_dkw_error        sqrt(ln(2/alpha) / 2n)           wd...py:38-39  (scalar host arithmetic)
    dkw_ecdf_bounds          clip(cdf -/+ eps, 0, 1)          wd...py:41-79
    get_cdf(arrays)          (cumsum(sorted)/sum, sorted)     mcsim.py:42-47
plus the row-wise metric table of mcsim.py:144-183 (`metric_table`) and that module's metric callables under their own
names and signatures - `Q`, `wc_fids`, `std_fids`, `Q_fids`, `wd_from_ideal_fids`, `Q_partial`,
`__metric_name_to_metric__` - lazy like the reference's `map` objects, ONE reduction launch behind each.
"""
from __future__ import annotations

import math

import numpy as np

from dataclasses import dataclass

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


# ---- mcsim.py:144-183 under the reference's names ---------------------------------------------------------------------
# The reference builds `map` objects: nothing is evaluated until somebody iterates (get_metric_dict_from_scratch wraps them
# in list(), mcsim.py:490-499).  Same here - a generator that, on first use, sends ALL rows through one reduction launch
# (rows of equal length; ragged input: one launch per distinct length) and then yields Python floats row by row.

def _rows_of(fids):
    if isinstance(fids, np.ndarray) and fids.ndim == 2:
        return [fids[i] for i in range(fids.shape[0])]             # views: wd_from_ideal_fids sorts them in place
    return [r if isinstance(r, np.ndarray) else np.asarray(r, dtype=np.float64) for r in fids]


def _lazy_row_metric(fids, pick, q_thresholds=(), sort_in_place=False):
    rows = _rows_of(fids)
    out = [None] * len(rows)
    by_len = {}
    for i, r in enumerate(rows):
        by_len.setdefault(int(np.size(r)), []).append(i)
    for K, idx in by_len.items():
        if K == 0:
            raise ValueError("empty fidelity row")
        slab = np.ascontiguousarray(np.stack([np.asarray(rows[i], dtype=np.float64).reshape(-1) for i in idx]))
        if sort_in_place:
            _as_fids(slab)                                         # check_fidtype's range guard (wd...py:23-25)
        red = backend.reduce_metrics(slab, q_thresholds=q_thresholds, want_sorted=sort_in_place)
        vals = np.asarray(pick(red), dtype=np.float64)
        for j, i in enumerate(idx):
            out[i] = float(vals[j])
            r = rows[i]
            if sort_in_place and isinstance(r, np.ndarray) and r.ndim == 1 and r.flags.writeable:
                r[...] = np.asarray(red["sorted"][j]).astype(r.dtype, copy=False)      # wd_from_ideal's fids.sort() (wd...py:105)
    yield from out


def Q(fid_array, threshold):
    """Fraction of the sample at or above `threshold` (mcsim.py:144-146; 1-D numpy arrays only, as its decorator insists)."""
    if not (type(fid_array) == np.ndarray and len(fid_array.shape) == 1):
        raise TypeError("make sure arg is a numpy array")
    red = backend.reduce_metrics(np.ascontiguousarray(fid_array, dtype=np.float64).reshape(1, -1), q_thresholds=(float(threshold),))
    return float(np.asarray(red["q"])[0, 0, 0])


def get_cdf(arrays):
    """`(sorted.cumsum() / sorted.sum(), sorted)` of a 1-D sample (mcsim.py:42-47, behind the same 1-D-numpy-array guard as
    `Q`): the row sort runs on the GPU (`rc_reduce_f64`'s `sorted_out`, any length), the O(K) running sum on the host in
    NumPy's own order, so both returned arrays equal the reference's bit for bit."""
    if not (type(arrays) == np.ndarray and len(arrays.shape) == 1):
        raise TypeError("make sure arg is a numpy array")
    red = backend.reduce_metrics(np.ascontiguousarray(arrays, dtype=np.float64).reshape(1, -1), q_thresholds=(), want_sorted=True)
    sarrays = np.asarray(red["sorted"][0]).astype(arrays.dtype, copy=False)
    return sarrays.cumsum() / sarrays.sum(), sarrays


def wc_fids(fids):
    """-min per row (mcsim.py:148-149)."""
    return _lazy_row_metric(fids, lambda red: -np.asarray(red["min"])[0])


def std_fids(fids):
    """np.std per row (mcsim.py:151-152)."""
    return _lazy_row_metric(fids, lambda red: np.asarray(red["std"])[0])


def Q_fids(fids, threshold=0.95):
    """-Q(row, threshold) per row (mcsim.py:154-157)."""
    return _lazy_row_metric(fids, lambda red: -np.asarray(red["q"])[0, 0], q_thresholds=(float(threshold),))


def wd_from_ideal_fids(fids):
    """wd_from_ideal per row (mcsim.py:158-159) - rows that are numpy arrays end up sorted, as `wd_from_ideal` leaves them."""
    return _lazy_row_metric(fids, lambda red: np.asarray(red["rim1"])[0], sort_in_place=True)


@dataclass
class Q_partial:
    """mcsim.py:169-176: Q_fids with the threshold bound at construction."""
    qthres: float = 0.95

    def Q_fids(self, fids):
        return Q_fids(fids, threshold=self.qthres)


__metric_name_to_metric__ = {METRIC_NAMES[0]: wd_from_ideal_fids,
                             METRIC_NAMES[1]: Q_partial(qthres=0.95).Q_fids,
                             METRIC_NAMES[2]: Q_partial(qthres=0.98).Q_fids,
                             METRIC_NAMES[3]: std_fids,
                             METRIC_NAMES[4]: wc_fids,
                             }
