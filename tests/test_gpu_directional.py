"""directional_perturbation (noise_model.py:150-201) on the GPU: the (index, a, b) entry with its class partition, the
complex symmetric QL route for non-Hermitian samples and its Pade-expm repair pass, the device-resident pipeline
against the host pipeline."""
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


def _dense_layout(N, idx, ab):
    """(draws (n, N, 3), diag_imag (n, N)) of directional samples given by (direction index, a, b) - the reference's
    semantics restated with NumPy (noise_model.py:160-167 for the list, :190-199 for the element pair)."""
    dirs = [(0, 0), (N - 1, N - 1)]
    for d in range(1, N - 1):
        for o in (-1, 0, 1):
            dirs.append((d, d + o))
    dirs += [(0, 1), (1, 0), (N - 2, N - 1), (N - 1, N - 2)]
    n = idx.size
    draws, imag = np.zeros((n, N, 3)), np.zeros((n, N))
    for s in range(n):
        p, q = dirs[idx[s]]
        a, b = ab[s]
        if p == q:
            draws[s, p, 0], imag[s, p] = a, -b                   # z[p,p] = a + ib overwritten by a - ib
        elif p == q + 1:
            draws[s, p, 1], draws[s, p, 2] = a, b                # z[p][p-1] = a + ib: lower element of bond p
        else:
            draws[s, q, 1], draws[s, q, 2] = a, -b               # z[p][p+1] = a + ib -> lower element z[q][p] = a - ib
    return draws, imag, len(dirs)


def test_directional_device_pipeline_equals_host_pipeline():
    """`directional_perturbation.fidelity_batch`: the device-resident pipeline (RNG parse, layout, class split on the
    GPU) against the round-2 host pipeline on the same stream - fidelities to 1e-10, identical generator state - for a
    chain, a ring, and out < in."""
    noise = importlib.import_module("code-robchar_amd.noise")
    rng = np.random.default_rng(5)
    for (N, a, b, topo) in ((7, 0, 6, "chain"), (5, 4, 1, "chain"), (6, 0, 3, "ring")):
        x = rand_ctrl(rng, 9, N)
        out = {}
        for mode in ("host", "device"):
            np.random.seed(77)
            nm = noise.directional_perturbation(Nspin=N, inspin=a, outspin=b, noise=0.05, topo=topo)
            out[mode] = (nm.fidelity_batch(x, 700, draws=mode), np.random.normal())
        assert np.abs(out["host"][0] - out["device"][0]).max() < TOL, (N, a, b, topo)
        assert out["host"][1] == out["device"][1]


@pytest.mark.parametrize("N", [2, 5, 7, 10, 12])
def test_complex_diagonal_route_vs_expm_kernel(be, N, monkeypatch):
    """`rc_mc_fidelity_nh_f64_async` on a chain: the lane-per-sample complex symmetric QL route (csym_core.h) against the
    oracle's per-sample scipy expm (small) and against the dense Pade-expm kernel alone (RC_NH_EXPM_ONLY=1) on a larger
    batch: directional-style single-site imaginary entries, imaginary parts everywhere, a NaN controller row, K = 1 lists
    (one controller row per sample - what the directional pipeline hands over), and large imaginary parts (|Im| up to 1:
    modes growing like e^{30}, results up to 1e25 - if a complex-orthogonal rotation breaks down there the sample is marked
    and the expm pass recomputes it; either way the two routes must agree, to 1e-7 RELATIVE at that dynamic range)."""
    import torch
    rng = np.random.default_rng(2200 + N)
    C, K = 5, 300
    ctrl = rand_ctrl(rng, C, N)
    ctrl[3] = np.nan
    for kind in ("single", "all", "large"):
        draws = 0.05 * rng.standard_normal((C, K, N, 3))
        imag = np.zeros((C, K, N))
        if kind == "single":
            site = rng.integers(0, N, (C, K))
            np.put_along_axis(imag, site[..., None], 0.05 * rng.standard_normal((C, K, 1)), axis=2)
        elif kind == "all":
            imag = 0.1 * rng.standard_normal((C, K, N))
        else:
            imag = rng.uniform(-1, 1, (C, K, N)) * (rng.random((C, K, N)) < 0.3)
        for (a, b) in ((0, N - 1), (N - 1, N // 2)):
            monkeypatch.delenv("RC_NH_EXPM_ONLY", raising=False)
            got = be.mc_fidelity_nonhermitian(ctrl, draws, imag, N, a, b)
            monkeypatch.setenv("RC_NH_EXPM_ONLY", "1")
            ref = be.mc_fidelity_nonhermitian(ctrl, draws, imag, N, a, b)
            monkeypatch.delenv("RC_NH_EXPM_ONLY", raising=False)
            assert np.isnan(got[3]).all() and np.isnan(ref[3]).all()
            ok = [0, 1, 2, 4]
            scale = np.maximum(1.0, np.abs(ref[ok]))
            rtol = 1e-7 if kind == "large" else 1e-9
            assert (np.abs(got[ok] - ref[ok]) <= rtol * scale).all(), (N, kind, a, b, np.abs(got[ok] - ref[ok]).max())
            if kind != "large":
                want = orc.fidelity_expm_loop(ctrl[:2], draws[:2, :40], N, a, b, diag_imag=imag[:2, :40])
                assert np.abs(got[:2, :40] - want).max() < TOL * max(1.0, np.abs(want).max()), (N, kind, a, b)
    # one controller row per sample (K = 1)
    M = 1000
    rows = rand_ctrl(rng, M, N)
    d1 = 0.05 * rng.standard_normal((M, 1, N, 3))
    i1 = 0.05 * rng.standard_normal((M, 1, N))
    got = be.mc_fidelity_nonhermitian(rows, d1, i1, N, 0, N - 1)
    want = orc.fidelity_expm_loop(rows[:60], d1[:60], N, 0, N - 1, diag_imag=i1[:60])
    assert np.abs(got[:60] - want).max() < TOL * max(1.0, np.abs(want).max())


def test_complex_symmetric_route_marks_near_defective_samples(be):
    """rc_mc_fidelity_nh_f64_async next to an exceptional point (a block [[i y, J], [J, -i y]], y -> J: the eigenvalues
    coalesce and no complex-orthogonal eigenbasis exists): the QL route's conditioning guard marks the samples it cannot
    carry and the Pade-expm pass recomputes them - the result agrees with the oracle's per-sample expm at every distance
    from the exceptional point, exactly at it, and with a cancelled bond beside the block (round 3 had no such guard: a
    finite but inaccurate number would have gone through)."""
    rng = np.random.default_rng(5)
    for N in (2, 3, 5, 8, 12):
        C, K = 2, 70
        for delta in (1e-2, 1e-5, 1e-8, 1e-11, 0.0):
            ctrl = np.empty((C, N + 1))
            ctrl[:, :N] = rng.uniform(-3, 3, (C, N))
            ctrl[:, 1] = ctrl[:, 0]
            ctrl[:, N] = rng.uniform(2, 30, C)
            draws = np.zeros((C, K, N, 3))
            draws[..., 2:, 1:] = 0.02 * rng.standard_normal((C, K, max(N - 2, 0), 2))
            if N > 2:
                draws[..., 2, 1] = -1.0                                  # bond 1-2 cancelled exactly
            imag = np.zeros((C, K, N))
            imag[..., 0], imag[..., 1] = 1.0 - delta, -(1.0 - delta)
            for (a, b) in ((0, 1), (0, 0), (1, 0), (0, N - 1)):
                want = orc.fidelity_expm_loop(ctrl, draws, N, a, b, diag_imag=imag)
                got = be.mc_fidelity_nonhermitian(ctrl, draws, imag, N, a, b)
                assert np.isfinite(got).all()
                assert (np.abs(got - want) <= 1e-9 * np.maximum(1.0, want)).all(), (N, delta, a, b, np.abs(got - want).max())


@pytest.mark.parametrize("N", [2, 3, 4, 5, 7, 8, 10, 12])
def test_directional_entry_vs_oracle(be, N):
    """rc_mc_fidelity_directional_f64_async: fidelities of `directional_perturbation` samples straight from (direction index,
    a, b) - class partition on the device, bond directions through the real tridiagonal routes, diagonal directions through
    the complex symmetric QL route - against the oracle's per-sample scipy.linalg.expm of the dense (possibly non-Hermitian)
    matrix: every direction of the list, every class of (in, out), XXZ offsets and non-unit couplings, a NaN-padded
    controller, sample counts that are no multiple of anything."""
    import torch
    rng = np.random.default_rng(600 + N)
    C, K = 5, 173
    ndir = 3 * N if N > 2 else 6
    dev = torch.device("cuda", torch.cuda.current_device())
    for trial, (a, b) in enumerate(((0, N - 1), (N - 1, 0), (0, N // 2), (N // 2, N // 2), (min(1, N - 1), 0))):
        ctrl = rand_ctrl(rng, C, N)
        ctrl[3] = np.nan
        idx = rng.integers(0, ndir, C * K).astype(np.int32)
        idx[:ndir] = np.arange(ndir)                                 # every direction at least once
        ab = 0.05 * rng.standard_normal((C * K, 2)) * (1.0 if trial % 2 else 4.0)
        draws, imag, nd = _dense_layout(N, idx, ab)
        assert nd == ndir
        h0d = orc.xxz_delta(N) if trial in (1, 3) else None
        h0o = rng.uniform(0.5, 1.5, N - 1) if trial == 2 else None
        want = orc.fidelity_expm_loop(ctrl, draws.reshape(C, K, N, 3), N, a, b, diag_imag=imag.reshape(C, K, N), h0_diag=h0d,
                                      h0_offdiag=h0o)
        got = be.mc_fidelity_directional(torch.from_numpy(ctrl).to(dev), torch.from_numpy(idx).to(dev),
                                         torch.from_numpy(ab).to(dev), N, a, b, K, h0_diag=h0d, h0_offdiag=h0o).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got[3]).all()
        err = np.nanmax(np.abs(got - want) / np.maximum(1.0, np.nan_to_num(want)))
        assert err < TOL, (N, a, b, err)


def test_directional_entry_refuses_what_it_does_not_cover(be):
    import torch
    lib = importlib.import_module("code-robchar_amd._lib")
    dev = torch.device("cuda", torch.cuda.current_device())
    ctrl = torch.zeros((1, 14), dtype=torch.float64, device=dev)
    idx = torch.zeros((4,), dtype=torch.int32, device=dev)
    ab = torch.zeros((4, 2), dtype=torch.float64, device=dev)
    with pytest.raises(lib.RobCharHipError):
        be.mc_fidelity_directional(ctrl, idx, ab, 13, 0, 12, 4)         # N > 12: the caller builds the dense layout instead


def test_directional_entry_multi_chunk_path(be, monkeypatch):
    """The partition passes of the directional entry run in chunks (2^23 samples by default); with the test knob
    RC_DIR_FID_CHUNK the chunk shrinks so that a small problem takes the multi-chunk path - chunk-relative idx / ab / fid
    pointers, `p.first`, the expm list's `sp_first`, one stream-ordered workspace per chunk: the result must equal the
    oracle to 1e-10 and the one-chunk run to rounding (1e-13: the class partition packs the samples of a CHUNK into waves, so the
    tile a sample shares its wave-uniform decisions with depends on the chunking - same routes, other neighbours), with a NaN controller, controllers that straddle chunk boundaries
    (K = 173 against chunks of 64 / 192 / 448 samples) and the expm-only repair route (RC_NH_EXPM_ONLY is not involved: the
    marked list is what the diagonal route cannot settle - forced here by a degenerate complex diagonal)."""
    import torch
    rng = np.random.default_rng(77)
    dev = torch.device("cuda", torch.cuda.current_device())
    for N, (a, b) in ((7, (0, 6)), (5, (0, 2)), (10, (9, 0))):
        C, K = 6, 173
        ndir = 3 * N
        ctrl = rand_ctrl(rng, C, N)
        ctrl[2] = np.nan
        ctrl[4, :N] = 0.0                                              # flat diagonal: diagonal directions with coinciding levels
        idx = rng.integers(0, ndir, C * K).astype(np.int32)
        ab = 0.05 * rng.standard_normal((C * K, 2))
        args = (torch.from_numpy(ctrl).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(ab).to(dev), N, a, b, K)
        monkeypatch.delenv("RC_DIR_FID_CHUNK", raising=False)
        one = be.mc_fidelity_directional(*args).cpu().numpy()
        draws, imag, _ = _dense_layout(N, idx, ab)
        want = orc.fidelity_expm_loop(ctrl, draws.reshape(C, K, N, 3), N, a, b, diag_imag=imag.reshape(C, K, N))
        assert np.array_equal(np.isnan(one), np.isnan(want)) and np.nanmax(np.abs(one - want)) < TOL
        for chunk in (64, 192, 448):
            monkeypatch.setenv("RC_DIR_FID_CHUNK", str(chunk))
            got = be.mc_fidelity_directional(*args).cpu().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(one)), (N, chunk)
            assert np.nanmax(np.abs(got - one)) < 1e-13 and np.nanmax(np.abs(got - want)) < TOL, (N, chunk, np.nanmax(np.abs(got - one)))
    monkeypatch.delenv("RC_DIR_FID_CHUNK", raising=False)


def test_directional_entry_validates_its_tensors(be):
    """A wrong `out` (dtype, shape, contiguity), non-tensor inputs or controllers on another device are refused before the
    kernel could write C * K doubles through them."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    N, C, K = 5, 3, 40
    ctrl = torch.from_numpy(rand_ctrl(np.random.default_rng(1), C, N)).to(dev)
    idx = torch.zeros((C * K,), dtype=torch.int32, device=dev)
    ab = torch.zeros((C * K, 2), dtype=torch.float64, device=dev)
    ok = torch.empty((C, K), dtype=torch.float64, device=dev)
    assert be.mc_fidelity_directional(ctrl, idx, ab, N, 0, 4, K, out=ok) is ok
    for bad in (torch.empty((C, K), dtype=torch.float32, device=dev), torch.empty((C, K + 1), dtype=torch.float64, device=dev),
                torch.empty((K, C), dtype=torch.float64, device=dev).t(), torch.empty((C, K), dtype=torch.float64), np.empty((C, K))):
        with pytest.raises(ValueError):
            be.mc_fidelity_directional(ctrl, idx, ab, N, 0, 4, K, out=bad)
    with pytest.raises(ValueError):
        be.mc_fidelity_directional(ctrl, idx.cpu().numpy(), ab, N, 0, 4, K)
    with pytest.raises(ValueError):
        be.mc_fidelity_directional(ctrl, idx, ab.cpu(), N, 0, 4, K)


def test_directional_perturbation_on_gpu():
    """The `directional_perturbation` mirror end-to-end on the GPU against the reference's seeded run."""
    noise = importlib.import_module("code-robchar_amd.noise")
    g = load_json("directional.json")
    for case in g["cases"]:
        np.random.seed(case["seed"])
        nm = noise.directional_perturbation(Nspin=case["Nspin"], inspin=case["inspin"], outspin=case["outspin"],
                                            noise=case["sigma"])
        got = nm.fidelity_batch(np.array(case["controllers"]), case["K"], ham_noisy=True)
        assert abs(np.random.normal() - case["rng_after"]) < 1e-15
        assert np.abs(got - np.array(case["fid"])).max() < TOL
        x = np.array(case["controllers"][0])
        np.random.seed(case["seed"])
        nm2 = noise.directional_perturbation(Nspin=case["Nspin"], inspin=case["inspin"], outspin=case["outspin"],
                                             noise=case["sigma"])
        assert abs(nm2.evaluate_noisy_fidelity(x, ham_noisy=True) - case["fid"][0][0]) < TOL
