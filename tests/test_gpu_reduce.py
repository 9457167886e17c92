"""Per-controller reductions and the row sort (rc_reduce_f64*, rc_rim_p_f64*): every kernel route at its boundaries, the
overlap hint, threshold counts, NaN rows, the sorted ECDF at any K, the metric API of wd_sortof_fast_implementation.py
/ mcsim.py on the GPU."""
import ctypes
import importlib
import json
import os
import pickle

import numpy as np
import pytest

from conftest import highfid_workload, load_json
from oracle import philox_host
from oracle import robchar_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_reduce_vs_oracle(be):
    rng = np.random.default_rng(3)
    C, K = 9, 1000
    F = rng.beta(8, 1.0, size=(C, K))
    F[3] = np.nan
    F[5, :10] = 1.0
    eps = orc.compute_dkw_error(0.05, K)
    red = be.reduce_metrics(F, dkw_eps=eps, want_sorted=True)
    variants = [F, np.clip(F - eps, 0, 1), np.clip(F + eps, 0, 1)]
    for v, data in enumerate(variants):
        rows = orc.metric_rows(data)
        assert np.allclose(red["rim1"][v], rows[orc.METRIC_NAMES[0]], atol=TOL, rtol=0, equal_nan=True)
        assert np.array_equal(-red["q"][v, 0], rows[orc.METRIC_NAMES[1]])
        assert np.array_equal(-red["q"][v, 1], rows[orc.METRIC_NAMES[2]])
        assert np.allclose(red["std"][v], rows[orc.METRIC_NAMES[3]], atol=TOL, rtol=0, equal_nan=True)
        assert np.allclose(-red["min"][v], rows[orc.METRIC_NAMES[4]], atol=0, rtol=0, equal_nan=True)
    ok = ~np.isnan(F[:, 0])
    assert np.array_equal(red["sorted"][ok], np.sort(F[ok], axis=1))
    assert np.isnan(red["sorted"][3]).all()


@pytest.mark.parametrize("nq", [0, 1, 2, 5, 8])
def test_reduce_threshold_counts(be, nq):
    rng = np.random.default_rng(nq)
    C, K = 6, 3001
    F = rng.beta(5, 1.0, size=(C, K))
    thr = np.linspace(0.5, 0.99, nq)
    eps = 0.02
    red = be.reduce_metrics(F, q_thresholds=thr, dkw_eps=eps)
    variants = [F, np.clip(F - eps, 0, 1), np.clip(F + eps, 0, 1)]
    assert red["q"].shape == (3, nq, C)
    for v, data in enumerate(variants):
        for j, t in enumerate(thr):
            assert np.array_equal(red["q"][v, j], (data >= t).mean(axis=1))
        assert np.allclose(red["std"][v], data.std(axis=1), atol=1e-14, rtol=0)
        assert np.allclose(red["rim1"][v], 1 - data.mean(axis=1), atol=1e-14, rtol=0)
        assert np.array_equal(red["min"][v], data.min(axis=1))
    lib = importlib.import_module("code-robchar_amd._lib")
    with pytest.raises(lib.RobCharHipError):
        be.reduce_metrics(F, q_thresholds=np.linspace(0, 1, 9))


@pytest.mark.parametrize("K", [1, 2, 3, 100, 4096, 10000, 16384])
def test_sorted_rows(be, K):
    rng = np.random.default_rng(K)
    F = rng.random((3, K))
    red = be.reduce_metrics(F, want_sorted=True)
    assert np.array_equal(red["sorted"], np.sort(F, axis=1))
    assert np.allclose(red["rim1"][0], 1 - F.mean(axis=1), atol=1e-13, rtol=0)


@pytest.mark.parametrize("K", [16385, 20000, 100000])
def test_sorted_rows_large_K(be, K):
    """ECDF sort beyond one LDS chunk (BASELINE config 4 has K = 1e5): bitonic network with HBM passes."""
    rng = np.random.default_rng(K)
    F = rng.random((3, K))
    F[1] = np.nan
    red = be.reduce_metrics(F, want_sorted=True)
    assert np.array_equal(red["sorted"][[0, 2]], np.sort(F[[0, 2]], axis=1))
    assert np.isnan(red["sorted"][1]).all()
    assert np.allclose(red["rim1"][0, [0, 2]], 1 - F[[0, 2]].mean(axis=1), atol=1e-13, rtol=0)
    assert np.allclose(red["std"][0, [0, 2]], F[[0, 2]].std(axis=1), atol=1e-13, rtol=0)


def test_wave_per_row_reduction(be):
    """Many short rows (paper layout: L x C rows of 100 draws) go through the wave-per-row reduction kernel; same
    outputs as the workgroup-per-row kernel and as the oracle, NaN rows included, every K up to its 2048 limit."""
    rng = np.random.default_rng(21)
    for (C, K) in ((64, 1), (100, 100), (11000, 100), (257, 2048), (300, 777)):
        F = rng.beta(5, 1.2, size=(C, K))
        F[C // 2] = np.nan
        F[3, : K // 2] = 1.0
        eps = orc.compute_dkw_error(0.05, K)
        got = be.reduce_metrics(F, dkw_eps=eps)
        small = be.reduce_metrics(F[:63], dkw_eps=eps)                     # < 64 rows: workgroup-per-row kernel
        for v, data in enumerate((F, np.clip(F - eps, 0, 1), np.clip(F + eps, 0, 1))):
            ok = ~np.isnan(F).any(axis=1)
            assert np.abs(got["rim1"][v][ok] - (1 - data[ok]).mean(axis=1)).max() < 1e-13
            assert np.abs(got["std"][v][ok] - data[ok].std(axis=1)).max() < 1e-13
            assert np.array_equal(got["min"][v][ok], data[ok].min(axis=1))
            for j, thr in enumerate((0.95, 0.98)):
                assert np.array_equal(got["q"][v, j][ok], (data[ok] >= thr).mean(axis=1))
            assert np.isnan(got["rim1"][v][~ok]).all() and (got["q"][v][:, ~ok] == 0).all()
        for k in ("rim1", "std", "min", "q"):
            assert np.allclose(got[k][..., :63], small[k], atol=1e-14, rtol=0, equal_nan=True)


@pytest.mark.parametrize("K", [2047, 2048, 2049, 4096, 4097, 8192, 8193, 10000, 10240, 10241])
@pytest.mark.parametrize("C", [5, 70])
def test_reduction_routes_at_their_boundaries(be, C, K):
    """The reduction picks its kernel by row length (and, for K <= 2048, row count): one wave per row, 128- / 256- / 512-thread
    workgroups with the row cached in registers.  Every route at both sides of every boundary, against NumPy: thresholds counted
    exactly, minimum exact, mean / std to 1e-14; a NaN row; and a row's result must not depend on the rows reduced with it."""
    rng = np.random.default_rng(K + C)
    F = rng.beta(6, 1.0, size=(C, K))
    F[1] = np.nan
    F[2, ::7] = 1.0
    thr, eps = np.array([0.9, 0.99]), 0.013
    red = be.reduce_metrics(F, q_thresholds=thr, dkw_eps=eps)
    ok = np.arange(C) != 1
    for v, data in enumerate([F, np.clip(F - eps, 0, 1), np.clip(F + eps, 0, 1)]):
        for j, t in enumerate(thr):
            assert np.array_equal(red["q"][v, j][ok], (data[ok] >= t).mean(axis=1))
        assert np.allclose(red["std"][v][ok], data[ok].std(axis=1), atol=1e-14, rtol=0)
        assert np.allclose(red["rim1"][v][ok], 1 - data[ok].mean(axis=1), atol=1e-14, rtol=0)
        assert np.array_equal(red["min"][v][ok], data[ok].min(axis=1))
        assert np.isnan(red["rim1"][v][1]) and np.isnan(red["std"][v][1]) and np.isnan(red["min"][v][1])
    if K > 2048:                       # (up to 2048 the route also depends on the row count: include/robchar_hip.h)
        alone = be.reduce_metrics(F[3:4], q_thresholds=thr, dkw_eps=eps)
        for name in ("rim1", "std", "min"):
            assert np.array_equal(alone[name][:, 0], red[name][:, 3]), name
    # round 5: the overlap hint (rc_reduce_ex_f64_async).  The NumPy path above is the blocking entry = standalone; through the
    # enqueue entry both hints, on device tensors: same exact counts / minimum, mean / std to 1e-14 - and the hint only changes
    # anything (the route, hence possibly the last bits) for rows of 8193 .. 10 240 values
    import torch
    Ft = torch.from_numpy(F).cuda()
    by_hint = {ov: be.reduce_metrics(Ft, q_thresholds=thr, dkw_eps=eps, overlapped=ov) for ov in (True, False)}
    for ov, r in by_hint.items():
        for name in ("rim1", "std", "min", "q"):
            got = r[name].cpu().numpy()
            if name in ("min", "q") or not ov:
                assert np.array_equal(got, red[name], equal_nan=True), (name, ov)          # the standalone route IS the blocking entry's
            else:
                assert np.allclose(got, red[name], atol=1e-14, rtol=0, equal_nan=True), (name, ov)
    if not (8192 < K <= 10240):
        for name in ("rim1", "std"):
            assert torch.equal(by_hint[True][name].nan_to_num(), by_hint[False][name].nan_to_num()), name


def test_metric_api_on_gpu():
    rimm = importlib.import_module("code-robchar_amd.rim_metrics")
    g = load_json("metrics.json")
    for k, vec in g["vectors"].items():
        v = g["values"][k]
        a = np.array(vec, dtype=np.float64)
        assert abs(rimm.wd_from_ideal(a) - v["wd_from_ideal"]) < 1e-14
        assert np.array_equal(a, np.sort(np.array(vec, dtype=np.float64)))
        for p in (0, 1, 2, 3):
            assert abs(rimm.RIM_p(np.array(vec, dtype=np.float64), p) - v[f"RIM_{p}"]) < 1e-13
    slab = np.array(g["slab"], dtype=np.float64)
    tab = rimm.metric_table(slab)[""]
    for name, want in g["slab_metrics"].items():
        assert np.allclose(tab[name], want, atol=1e-14, rtol=0, equal_nan=True), name
    for name, fn in rimm.__metric_name_to_metric__.items():           # mcsim.py:144-183 under the reference's names, on the GPU
        assert np.allclose(list(fn(slab.copy())), g["slab_metrics"][name], atol=1e-14, rtol=0, equal_nan=True), name
    assert abs(rimm.Q(slab[0], 0.95) + g["slab_metrics"]["Q th. 0.95"][0]) < 1e-15
    # get_cdf (mcsim.py:42-47): GPU row sort + NumPy's own running sum = the reference's two arrays bit for bit, any length
    for row in (slab[0], np.random.default_rng(3).random(20001)):
        cdf, srt = rimm.get_cdf(row.copy())
        assert np.array_equal(srt, np.sort(row)) and np.array_equal(cdf, np.sort(row).cumsum() / np.sort(row).sum())
    with pytest.raises(TypeError):
        rimm.get_cdf(slab)                                            # the reference's 1-D guard (mcsim.py:35-39)
    # the reference's own unit-test identities (wd_sortof_fast_implementation.py:196-205)
    X = np.random.default_rng(0).normal(0.85, 0.8, size=10000).clip(min=0, max=1)
    mine = rimm.wd_from_ideal(X.copy())
    assert abs(np.sqrt(mine * mine + X.var()) - rimm.RIM_p(X, p=2)) < 1e-12
    from scipy.stats import wasserstein_distance
    assert abs(wasserstein_distance(X, np.ones_like(X)) - mine) < 1e-12
