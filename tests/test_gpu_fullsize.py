"""BASELINE-size parity: the GPU configurations at full size - size-independent properties (unitarity, gauge invariance,
reciprocity, RIM == mean infidelity) on SURVEY 8(d)'s synthetic controllers, and (round 5) every `out` against the
oracle with absolute AND relative bounds on delocalised, high-fidelity controller sets (tests/golden/highfid.npz);
config 4 whole on one GPU."""
import ctypes
import importlib
import json
import os
import pickle

import numpy as np
import pytest

from conftest import highfid_workload, load_json
from gpu_common import rand_ctrl
from oracle import philox_host
from oracle import robchar_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10
REL = 1e-9           # relative bound on samples with F > 1e-3 (measured: ~1e-13)


def compare(got, want, what):
    """absolute bound everywhere, relative bound where the fidelity is not tiny; returns (max abs, max rel, share F > 1e-3)"""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    big = want > 1e-3
    rel = float((err[big] / want[big]).max()) if big.any() else 0.0
    assert err.max() < TOL, (what, float(err.max()))
    assert rel < REL, (what, rel)
    return float(err.max()), rel, float(big.mean())


def oracle_chunked(ctrl, draws, N, a, b, h0, chunk=2000):
    return np.concatenate([orc.fidelity_eigh(ctrl, draws[:, k:k + chunk], N, a, b, h0_diag=h0) for k in range(0, draws.shape[1], chunk)],
                          axis=1)


@pytest.mark.parametrize("cfg", [(2, 5, 4, False), (3, 7, 6, False), (5, 10, 9, True)], ids=["config2", "config3", "config5"])
def test_full_size_properties(be, cfg):
    """BASELINE configs 2, 3, 5 at full size (100 x 10000; N=5 0->4, N=7 0->6, N=10 XXZ 0->9): size-independent
    properties instead of the oracle.

    (1) unitarity: sum over `out` of |U[out,in]|^2 = 1 for every sample;
    (2) gauge invariance: rotating each complex coupling by an arbitrary phase leaves the fidelity unchanged;
    (3) reciprocity |U[out,in]| = |U[in,out]|;  (4) a 2 % subsample against the oracle;
    (5) RIM from the reduction kernel == mean infidelity of the tensor; (6) the fast path is (all but) never left.
    """
    cid, N, out, xxz = cfg
    rng = np.random.default_rng(20220714 + cid)
    C, K = 100, 10000
    h0 = orc.xxz_delta(N) if xxz else None
    ctrl = rand_ctrl(rng, C, N)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    be.general_path_tiles(reset=True)
    F = [be.mc_fidelity(ctrl, draws, N, 0, o, h0_diag=h0) for o in range(N)]
    # the benchmark workloads stay on the wave-wide routes: the end-to-end launch (the BASELINE shape itself) at most a stray
    # tile; the N - 1 general-adjugate launches hand every sample with a pair closer than 4e-6 of the scale to the
    # in-register eigenvector repair (round 3: the fuzz campaign showed the adjugate numerators to be noisy below that) -
    # a fraction of a percent of the tiles (same result, checked below)
    n_repaired = be.general_path_tiles()
    print(f"config {cid}: {n_repaired} repaired tiles in {N} launches of 15 700")
    # measured (rounds 3 / 4, gpurun_out/*/pytest.log): 0 / 0 / 153-200 tiles for configs 2 / 3 / 5; the bound is ~10x that - a
    # regression of the repair RATE is a cost regression (every such tile runs the eigenvector route for its lanes)
    assert n_repaired <= 2000, n_repaired
    be.general_path_tiles(reset=True)
    be.mc_fidelity(ctrl, draws, N, 0, N - 1, h0_diag=h0)
    assert be.general_path_tiles() <= 2
    assert np.abs(sum(F) - 1.0).max() < 1e-11
    assert all((f >= 0).all() and (f <= 1 + 1e-12).all() for f in F)
    # gauge: (1 + g1 + i g2) -> e^{i theta} (1 + g1 + i g2)
    theta = rng.uniform(0, 2 * np.pi, size=(C, K, N))
    z = (1.0 + draws[..., 1] + 1j * draws[..., 2]) * np.exp(1j * theta)
    d2 = draws.copy()
    d2[..., 1] = z.real - 1.0
    d2[..., 2] = z.imag
    assert np.abs(be.mc_fidelity(ctrl, d2, N, 0, out, h0_diag=h0) - F[out]).max() < TOL
    assert np.abs(be.mc_fidelity(ctrl, draws, N, out, 0, h0_diag=h0) - F[out]).max() < TOL
    sel = rng.choice(K, 200, replace=False)
    want = orc.fidelity_eigh(ctrl, draws[:, sel], N, 0, out, h0_diag=h0)
    assert np.abs(F[out][:, sel] - want).max() < TOL
    mid = N // 2
    want = orc.fidelity_eigh(ctrl, draws[:, sel], N, 0, mid, h0_diag=h0)
    assert np.abs(F[mid][:, sel] - want).max() < TOL
    red = be.reduce_metrics(F[out])
    assert np.abs(red["rim1"][0] - (1 - F[out]).mean(axis=1)).max() < 1e-12
    assert np.array_equal(red["min"][0], F[out].min(axis=1))


def test_config4_rank_share_philox(be):
    """One rank's share of BASELINE config 4 (nspin=7, 0->3, 1000 x 100000 over 8 GPUs = 125 controllers x 1e5
    draws per GPU = 1.25e7 evaluations, 2.1 GB of draws): draws generated on the device (Philox), fidelity kernel,
    per-controller reductions and the K = 1e5 row sort; checked by a subsample against the oracle fed with the same
    (copied back) draws and by size-independent properties."""
    import torch
    N, C, K = 7, 125, 100000
    rng = np.random.default_rng(4)
    ctrl = rand_ctrl(rng, C, N)
    ct = torch.from_numpy(ctrl).cuda()
    draws = be.philox_normal((C, K, N, 3), seed=44, scale=0.05, as_torch=True)
    F3 = be.mc_fidelity(ct, draws, N, 0, 3)
    red = be.reduce_metrics(F3, dkw_eps=orc.compute_dkw_error(0.05, K), want_sorted=True)
    tot = sum(be.mc_fidelity(ct, draws, N, 0, o) for o in range(N))
    assert float((tot - 1).abs().max()) < 1e-11                               # unitarity
    srt = red["sorted"]
    assert bool((srt[:, 1:] >= srt[:, :-1]).all())                            # sortedness
    assert float((srt.sum(dim=1) - F3.sum(dim=1)).abs().max()) < 1e-6         # same multiset (checksum)
    assert torch.equal(srt[:, 0], F3.min(dim=1).values) and torch.equal(srt[:, -1], F3.max(dim=1).values)
    assert float((red["rim1"][0] - (1 - F3).mean(dim=1)).abs().max()) < 1e-12
    assert float((red["std"][0] - F3.std(dim=1, unbiased=False)).abs().max()) < 1e-12
    sel_c = [0, 57, 124]
    sel_k = torch.arange(0, K, 4999, device="cuda")
    sub = draws[sel_c][:, sel_k].cpu().numpy()
    want = orc.fidelity_eigh(ctrl[sel_c], sub, N, 0, 3)
    assert np.abs(F3[sel_c][:, sel_k].cpu().numpy() - want).max() < TOL


# ----------------------------------------------------------------------------------------------------------------
# BASELINE config 4, the WHOLE problem on one GPU
# ----------------------------------------------------------------------------------------------------------------
def test_config4_whole_problem_one_gpu(be):
    """nspin=7, 0->3, 1000 controllers x 100 000 perturbations = 1e8 evaluations per level, device Philox draws
    (2.1e9 normals = 16.8 GB, never on the host), fidelity kernel + per-controller reductions; checked by
    (1) a subsample against the oracle fed with the HOST-regenerated Philox elements, (2) unitarity on a controller
    block, (3) RIM == mean infidelity and std == tensor std for every controller, (4) the metrics-only sharded C entry
    (ndev = 1, chunked through its 4 GiB workspace) reproducing the same metric rows bit for bit."""
    import torch
    N, C, K, a, b, sigma, seed = 7, 1000, 100000, 0, 3, 0.05, 2024
    rng = np.random.default_rng(20220714 + 4)
    ctrl = rand_ctrl(rng, C, N)
    ct = torch.from_numpy(ctrl).cuda()
    draws = be.philox_normal((C, K, N, 3), seed=seed, scale=sigma, as_torch=True)
    be.general_path_tiles(reset=True)
    F = be.mc_fidelity(ct, draws, N, a, b)
    eps = orc.compute_dkw_error(0.05, K)
    red = be.reduce_metrics(F, dkw_eps=eps)
    torch.cuda.synchronize()
    n_repaired = be.general_path_tiles()
    print(f"config 4: {n_repaired} of 1 563 000 tiles with a sample on the eigenvector repair route")
    # (general adjugate weights: pairs closer than 4e-6 of the scale, or a sample the sum-rule guard rejects; measured 3 in
    # round 3 - the bound is ~20x the measurement, not a percentage of the launch)
    assert n_repaired <= 64, n_repaired
    assert float((red["rim1"][0] - (1 - F).mean(dim=1)).abs().max()) < 1e-12
    assert float((red["std"][0] - F.std(dim=1, unbiased=False)).abs().max()) < 1e-12
    assert torch.equal(red["min"][0], F.min(dim=1).values)
    assert bool(((F >= 0) & (F <= 1 + 1e-12)).all())
    # unitarity on the first 64 controllers (7 launches x 6.4e6 evaluations)
    tot = sum(be.mc_fidelity(ct[:64], draws[:64], N, a, o) for o in range(N))
    assert float((tot - 1).abs().max()) < 1e-11
    # subsample vs oracle on host-regenerated stream elements
    per_ctrl = K * N * 3
    for c in (0, 333, 999):
        for k in (0, 31337, 99999):
            off = c * per_ctrl + k * N * 3
            g = philox_host.philox_normal(seed, off, N * 3, sigma).reshape(1, 1, N, 3)
            assert np.abs(g - draws[c, k].cpu().numpy()).max() < 1e-15
            want = orc.fidelity_eigh(ctrl[c:c + 1], g, N, a, b)[0, 0]
            assert abs(float(F[c, k]) - want) < TOL
    want_rows = {k: red[k].cpu().numpy() for k in ("rim1", "std", "min", "q")}
    del draws, F, tot
    torch.cuda.empty_cache()
    got = be.mc_metrics_sharded(ctrl, K, N, a, b, seed=seed, offset=0, sigma=sigma, devices=[0], dkw_eps=eps)
    for k in want_rows:
        assert np.array_equal(got[k], want_rows[k]), k


@pytest.mark.parametrize("cid", [2, 3, 5], ids=["config2", "config3", "config5"])
def test_full_size_on_delocalised_controllers_every_out(be, highfid, cid):
    """BASELINE configs 2, 3, 5 at full size (100 x 10 000, sigma 0.05): EVERY `out` of the timed kernels against the oracle on a
    subsample of 100 draws per controller, the BASELINE pair on ALL 10^6 samples (config 3) / every 10th draw (2, 5); the
    workload's own statistics are asserted first (median fidelity, share above 1e-3) - they are what gives the bounds teeth."""
    N, a, b, ctrl, h0 = highfid_workload(cid, 100)
    C, K = 100, 10000
    rng = np.random.default_rng(20220714 + 50 + cid)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    be.polish_tiles(reset=True)
    F = [be.mc_fidelity(ctrl, draws, N, a, o, h0_diag=h0) for o in range(N)]
    off = be.polish_tiles() / (N * C * ((K + 63) // 64))
    med, share = float(np.median(F[b])), float((F[b] > 1e-3).mean())
    print(f"config {cid} delocalised: median F {med:.4f}, mean {F[b].mean():.4f}, share F > 1e-3 {share:.4f}, "
          f"tiles off the one-step path {off:.3f}")
    assert med >= 0.1 and share >= 0.9 and F[b].mean() >= 0.3
    assert np.abs(sum(F) - 1.0).max() < 1e-11                                  # unitarity, every sample
    sel = rng.choice(K, 100, replace=False)
    worst = (0.0, 0.0)
    for o in range(N):
        e, r, _ = compare(F[o][:, sel], orc.fidelity_eigh(ctrl, draws[:, sel], N, a, o, h0_diag=h0), f"config {cid} out {o}")
        worst = (max(worst[0], e), max(worst[1], r))
    step = 1 if cid == 3 else 10
    e, r, _ = compare(F[b][:, ::step], oracle_chunked(ctrl, draws[:, ::step], N, a, b, h0), f"config {cid} BASELINE pair")
    print(f"config {cid}: every out max|dF| {worst[0]:.2e} rel {worst[1]:.2e}; pair ({a},{b}) on {C * K // step} samples {e:.2e} rel {r:.2e}")
    # reciprocity on the delocalised set
    assert np.abs(be.mc_fidelity(ctrl, draws, N, b, a, h0_diag=h0) - F[b]).max() < TOL
    # and the REFERENCE ITSELF on this workload class: make_golden.py's (controller, 4 injected draws) -> fidelity tables
    tag = f"c{cid}"
    rows = highfid[tag + "_draws"].shape[0]
    got = be.mc_fidelity(ctrl[:rows], highfid[tag + "_draws"], N, a, b, h0_diag=h0)
    compare(got, highfid[tag + "_fid"], f"config {cid} vs the reference's own outputs")
    got0 = be.mc_fidelity(ctrl[:rows], np.zeros((rows, 1, N, 3)), N, a, b, h0_diag=h0)[:, 0]
    compare(got0, highfid[tag + "_fid_noiseless"], f"config {cid} noiseless vs the reference")
    if cid == 2:                                                               # the shipped cache's sigma_sim = 0 row (reference-authored)
        assert np.abs(got0 - highfid["c2_shipped_sigma0_fid"]).max() < TOL


def test_config4_whole_problem_on_shipped_controllers(be, highfid):
    """BASELINE config 4 whole on one GPU (N = 7, 0 -> 3, 1000 x 100 000, device Philox draws: 16.8 GB) on the reference's 100
    shipped 0 -> 3 L-BFGS controllers tiled to 1000: EVERY `out` (seven launches of 1e8 evaluations) against the oracle on
    20 000 random (controller, draw) pairs whose draws are copied back from the device, absolute and relative bounds; the
    fused-Philox route (no draw tensor) must reproduce the pair (0, 3) bit for bit."""
    import torch
    N, a, b, ctrl, h0 = highfid_workload(4, 1000)
    C, K, sigma, seed = 1000, 100000, 0.05, 20220714 + 4
    ct = torch.from_numpy(ctrl).cuda()
    draws = be.philox_normal((C, K, N, 3), seed=seed, scale=sigma, as_torch=True)
    rng = np.random.default_rng(44)
    ci, ki = rng.integers(0, C, 20000), rng.integers(0, K, 20000)
    cit, kit = torch.from_numpy(ci).cuda(), torch.from_numpy(ki).cuda()
    sub = draws[cit, kit].cpu().numpy()[:, None]                               # (20000, 1, N, 3)
    tot = None
    worst = (0.0, 0.0)
    for o in range(N):
        F = be.mc_fidelity(ct, draws, N, a, o)
        if o == b:
            med, share = float(F.median()), float((F > 1e-3).double().mean())
            print(f"config 4 on shipped controllers: median F {med:.4f}, mean {float(F.mean()):.4f}, share F > 1e-3 {share:.4f}")
            assert med >= 0.1 and share >= 0.9
            Fb = F.clone()
        got = F[cit, kit].cpu().numpy()
        want = orc.fidelity_eigh(ctrl[ci], sub, N, a, o)[:, 0]
        e, r, _ = compare(got, want, f"config 4 out {o}")
        worst = (max(worst[0], e), max(worst[1], r))
        blk = F[:40].clone()
        tot = blk if tot is None else tot + blk
        del F
    print(f"config 4 on shipped controllers: every out, 20 000 samples each: max|dF| {worst[0]:.2e} rel {worst[1]:.2e}")
    assert float((tot - 1).abs().max()) < 1e-11                                # unitarity on a 40-controller block (4e6 samples)
    del draws, tot
    torch.cuda.empty_cache()
    fused = be.mc_fidelity_philox(ct, K, N, a, b, seed, sigma=sigma)
    assert torch.equal(fused, Fb)
    # the REFERENCE ITSELF on these controllers: make_golden.py's (controller, 4 injected draws) -> fidelity table
    compare(be.mc_fidelity(ctrl[:100], highfid["c4_draws"], N, a, b), highfid["c4_fid"], "config 4 vs the reference's own outputs")
