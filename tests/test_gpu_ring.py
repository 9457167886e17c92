"""Ring topology and the dense kernels: lane-per-sample Householder + QL (all-fp64), the mixed-precision ring route with
its repair launch and stream-ordered repair list, the packed complex Jacobi kernel, the Pade-expm kernel; full-size
properties of the ring."""
import ctypes
import importlib
import json
import os
import pickle

import numpy as np
import pytest

from conftest import highfid_workload, load_json
from gpu_common import rand_ctrl, _h0
from oracle import philox_host
from oracle import robchar_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_jacobi_kernel_ring_golden_and_cross_check(be, kernel_cases):
    """The wave-per-sample complex Hermitian Jacobi kernel: ring-topology outputs of the unmodified reference
    (golden), and - as an independent on-device cross-check - chain / XXZ cases against the reference too."""
    worst_ring = worst_chain = 0.0
    for case in kernel_cases:
        h0 = _h0(case)
        for s in range(case["draws"].shape[0]):
            got = be.mc_fidelity(case["ctrl"], case["draws"][s], case["N"], case["inspin"], case["outspin"],
                                 h0_diag=h0, ring=case["mode"] == "ring", kernel="jacobi")
            err = np.abs(got - case["fid"][s]).max()
            if case["mode"] == "ring":
                worst_ring = max(worst_ring, err)
                auto = be.mc_fidelity(case["ctrl"], case["draws"][s], case["N"], case["inspin"], case["outspin"],
                                      ring=True)                      # auto -> the mixed-precision ring route (N <= 10)
                hh = be.mc_fidelity(case["ctrl"], case["draws"][s], case["N"], case["inspin"], case["outspin"],
                                    ring=True, kernel="ring_hh")      # the all-fp64 Householder + QL ring kernel
                assert np.abs(auto - case["fid"][s]).max() < TOL and np.abs(hh - case["fid"][s]).max() < TOL
            else:
                worst_chain = max(worst_chain, err)
    assert worst_ring < TOL and worst_chain < TOL, (worst_ring, worst_chain)


@pytest.mark.parametrize("N", [2, 3, 8, 9, 13, 16])
def test_jacobi_kernel_random(be, N):
    rng = np.random.default_rng(N)
    C, K = 5, 37
    ctrl = rand_ctrl(rng, C, N)
    ctrl[3] = np.nan
    draws = 0.1 * rng.standard_normal((C, K, N, 3))
    for ring in (False, True):
        got = be.mc_fidelity(ctrl, draws, N, 0, N - 1, ring=ring, kernel="jacobi")
        want = orc.fidelity_eigh(ctrl, draws, N, 0, N - 1, ring=ring and N > 2)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.nanmax(np.abs(got - want)) < TOL, (N, ring)
    # chain kernel vs Jacobi kernel on the same device inputs
    a = be.mc_fidelity(ctrl, draws, N, 0, N - 1, kernel="tridiag_ql")
    b = be.mc_fidelity(ctrl, draws, N, 0, N - 1, kernel="jacobi")
    assert np.nanmax(np.abs(a - b)) < TOL


def test_expm_kernel_golden_and_nonhermitian(be, kernel_cases):
    """The dense Pade-expm kernel (wave per sample): (1) as a third independent cross-check on the reference's
    chain / XXZ / ring outputs, (2) on the reference's seeded `directional_perturbation` run, whose diagonal
    directions are NON-Hermitian (fidelities may exceed 1), (3) tiny and huge |T H| (all Pade orders, many squarings)."""
    worst = 0.0
    for case in kernel_cases[::3]:
        for s in (0, 2):
            got = be.mc_fidelity(case["ctrl"], case["draws"][s], case["N"], case["inspin"], case["outspin"],
                                 h0_diag=_h0(case), ring=case["mode"] == "ring", kernel="expm")
            worst = max(worst, np.abs(got - case["fid"][s]).max())
    assert worst < TOL, worst
    g = load_json("directional.json")
    for case in g["cases"]:
        n, C, K = case["Nspin"], case["C"], case["K"]
        draws = np.zeros((C, K, n, 3))
        imag = np.zeros((C, K, n))
        for s, (idx, (a, b)) in enumerate(zip(case["index"], case["ab"])):
            draws[s // K, s % K], imag[s // K, s % K] = orc.directional_to_layout(n, idx, a, b)
        got = be.mc_fidelity_nonhermitian(np.array(case["controllers"]), draws, imag, n, case["inspin"], case["outspin"])
        assert np.abs(got - np.array(case["fid"])).max() < TOL
    rng = np.random.default_rng(8)
    for N in (2, 6, 16):
        ctrl = rand_ctrl(rng, 6, N)
        ctrl[:, N] = [1e-4, 5e-3, 0.05, 0.2, 3.0, 60.0]          # norms from 1e-3 to 1e3: every Pade order
        ctrl[3] *= 0.3
        draws = 0.1 * rng.standard_normal((6, 9, N, 3))
        imag = 0.1 * rng.standard_normal((6, 9, N))
        got = be.mc_fidelity_nonhermitian(ctrl, draws, imag, N, 0, N - 1)
        want = orc.fidelity_expm_loop(ctrl, draws, N, 0, N - 1, diag_imag=imag)
        assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max()), N


@pytest.mark.parametrize("N", list(range(2, 17)))
def test_ring_kernels_vs_oracle(be, N):
    """Ring topology (noise_model.py:83-85): the lane-per-sample routes (AUTO = the mixed-precision route + repair launch,
    `ring_hh` = all-fp64; N = 3..10 through the dense Householder reduction, N = 11..16 - round 5 - through the folded band
    reduction) and the wave-per-sample Jacobi kernel against the oracle: random and near-degenerate (translation-invariant)
    controllers, every class of in/out pair, XXZ diagonal, NaN rows, ragged K, sigma up to 0.3."""
    rng = np.random.default_rng(300 + N)
    C, K = 7, 131
    ctrl = rand_ctrl(rng, C, N)
    ctrl[1, :N] = rng.uniform(-1e-6, 1e-6, N)
    ctrl[2, N] = -ctrl[2, N]
    ctrl[4] = np.nan
    lib = importlib.import_module("code-robchar_amd._lib")
    for sigma in (0.0, 0.05, 0.3):
        draws = sigma * rng.standard_normal((C, K, N, 3))
        for (a, b) in ((0, N - 1), (0, N // 2), (N - 1, 1 % N), (1 % N, 1 % N)):
            want = orc.fidelity_eigh(ctrl, draws, N, a, b, ring=True)
            kernels = ["auto", "jacobi"] + (["ring_hh"] if N >= 3 else [])
            for kern in kernels:
                got = be.mc_fidelity(ctrl, draws, N, a, b, ring=True, kernel=kern)
                assert np.array_equal(np.isnan(got), np.isnan(want))
                assert np.nanmax(np.abs(got - want)) < TOL, (N, sigma, a, b, kern)
    h0 = orc.xxz_delta(N, ring=N > 2)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    want = orc.fidelity_eigh(ctrl, draws, N, 0, N - 1, h0_diag=h0, ring=True)
    assert np.nanmax(np.abs(be.mc_fidelity(ctrl, draws, N, 0, N - 1, h0_diag=h0, ring=True) - want)) < TOL
    with pytest.raises(lib.RobCharHipError, match="ring-topology"):
        be.mc_fidelity(ctrl, draws, N, 0, N - 1, ring=False, kernel="ring_hh")


def test_ring_full_size_properties(be):
    """N = 7 ring at BASELINE size (100 x 10 000): unitarity over `out`, both directions against the oracle on a subsample,
    agreement of the two ring kernels, and only a few percent of the samples on the repair list."""
    rng = np.random.default_rng(77)
    N, C, K = 7, 100, 10000
    ctrl = rand_ctrl(rng, C, N)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    be.general_path_tiles(reset=True)
    F = [be.mc_fidelity(ctrl, draws, N, 0, o, ring=True) for o in range(N)]
    n_rep = be.general_path_tiles()                        # repaired waves of 64 samples (pairs closer than 1e-3 of the scale)
    print(f"ring N = 7: {n_rep} repaired waves in {N} launches of 15 625")
    assert n_rep <= 10 * N, n_rep                          # measured: 7 (one repair wave per launch: ~450 listed samples of 1e6)
    assert np.abs(sum(F) - 1.0).max() < 1e-11
    # (no reciprocity check: the perturbed couplings thread a flux through the ring, time reversal is broken and
    #  |U[3,0]| != |U[0,3]| in general - unlike the chain, which is gauge-equivalent to a real matrix)
    sel = rng.choice(K, 100, replace=False)
    assert np.abs(F[3][:, sel] - orc.fidelity_eigh(ctrl, draws[:, sel], N, 0, 3, ring=True)).max() < TOL
    rev = be.mc_fidelity(ctrl, draws, N, 3, 0, ring=True)
    assert np.abs(rev[:, sel] - orc.fidelity_eigh(ctrl, draws[:, sel], N, 3, 0, ring=True)).max() < TOL
    assert np.abs(rev - F[3]).max() > 1e-6
    jac = be.mc_fidelity(ctrl[:10], draws[:10], N, 0, 3, ring=True, kernel="jacobi")
    assert np.abs(jac - F[3][:10]).max() < TOL


@pytest.mark.parametrize("N", [3, 5, 7, 10, 11, 13, 16])
def test_ring_mixed_route_and_repair(be, N):
    """Ring topology, AUTO = the mixed-precision route (sparse fp32 Householder + fp32 QL starting values, fp64 Halley on
    chi_ring, two-path cofactor weights) + the repair launch behind it.  (1) random rings: parity with the oracle and with
    the all-fp64 ring kernel for every class of (in, out); only the samples with a pair closer than 1e-3 of the scale - a few
    percent - are listed for the repair kernel; (2) a translation-invariant ring - degenerate pairs k <-> -k, split only by
    the noise - lists EVERY sample: the repair kernel recomputes them all through the all-fp64 route, lane per sample, NaN
    controller rows stay NaN, ragged K; (3) the splitting scanned from 1e-8 to 1e-2: the regime in which the two-path
    weights lose digits (the fuzz campaign of round 3 found 6e-10 there with the chain route's 4e-6 threshold)."""
    rng = np.random.default_rng(1300 + N)
    C, K = 12, 1000                                        # ragged: 1000 = 15 x 64 + 40
    ctrl = rand_ctrl(rng, C, N)
    ctrl[5] = np.nan
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    be.general_path_tiles(reset=True)
    for (a, b) in ((0, N - 1), (N - 1, 0), (0, N // 2), (N // 2, 1), (1, 1)):
        got = be.mc_fidelity(ctrl, draws, N, a, b, ring=True)
        want = orc.fidelity_eigh(ctrl, draws, N, a, b, ring=True)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.nanmax(np.abs(got - want)) < TOL, (N, a, b, np.nanmax(np.abs(got - want)))
        hh = be.mc_fidelity(ctrl, draws, N, a, b, ring=True, kernel="ring_hh")
        assert np.nanmax(np.abs(got - hh)) < TOL
    # repaired waves of 64 samples, 5 launches (more levels, more pairs closer than 1e-3 of the scale: ~N^2 / 1000 of the samples)
    assert be.general_path_tiles() <= (0.15 if N <= 10 else 0.35) * 5 * (C - 1) * K / 64 + 5
    flat = ctrl.copy()
    flat[:, :N] = rng.uniform(-1e-6, 1e-6, (C, N))
    flat[5] = np.nan
    tiny = 1e-7 * rng.standard_normal((C, K, N, 3))
    be.general_path_tiles(reset=True)
    got = be.mc_fidelity(flat, tiny, N, 0, N // 2, ring=True)
    want = orc.fidelity_eigh(flat, tiny, N, 0, N // 2, ring=True)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got[5]).all()
    assert np.nanmax(np.abs(got - want)) < TOL
    assert be.general_path_tiles() >= (C - 1) * K // 64                 # every sample of every real controller was repaired
    for lg in range(-8, -1):
        d = 0.2 * 10.0 ** lg * rng.standard_normal((C, K, N, 3))
        for (a, b) in ((0, 0), (N - 1, 1)):
            got = be.mc_fidelity(flat, d, N, a, b, ring=True)
            want = orc.fidelity_eigh(flat, d, N, a, b, ring=True)
            assert np.nanmax(np.abs(got - want)) < 2e-11, (N, lg, a, b, np.nanmax(np.abs(got - want)))


def test_ring_route_on_fresh_side_streams(be):
    """The ring route's repair list is per (device, stream) and lives in stream-ordered memory: the FIRST ring launch on a
    stream nobody has used before (torch side streams are non-blocking: not ordered after the null stream) must give the
    right answer - counters zeroed on that stream -, a larger problem on the same stream grows the list behind the kernels
    that still read the old one, rc_reserve_ring pre-sizes it, rc_release_stream hands it back and the next launch starts
    over.  Flat diagonals (translation-invariant rings) list EVERY sample: a counter that was not zero, or a list that was
    freed too early, shows up as wrong or NaN fidelities."""
    import torch
    lib = importlib.import_module("code-robchar_amd._lib").load()
    rng = np.random.default_rng(41)
    N = 6
    dev = torch.device("cuda", torch.cuda.current_device())
    for trial in range(3):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            for (C, K) in ((3, 500), (9, 1500), (2, 100)):               # grow, then a smaller problem in the larger list
                ctrl = rand_ctrl(rng, C, N)
                if trial != 1:
                    ctrl[:, :N] = rng.uniform(-1e-6, 1e-6, (C, N))       # every sample listed for the repair kernel
                draws = (1e-7 if trial != 1 else 0.05) * rng.standard_normal((C, K, N, 3))
                if trial == 2 and (C, K) == (3, 500):
                    assert lib.rc_reserve_ring(dev.index or 0, ctypes.c_void_p(st.cuda_stream), 9 * 1500) == 0
                got = be.mc_fidelity(torch.from_numpy(ctrl).to(dev), torch.from_numpy(draws).to(dev), N, 0, N // 2, ring=True)
                want = orc.fidelity_eigh(ctrl, draws, N, 0, N // 2, ring=True)
                st.synchronize()
                assert np.abs(got.cpu().numpy() - want).max() < TOL, (trial, C, K)
            assert lib.rc_release_stream(dev.index or 0, ctypes.c_void_p(st.cuda_stream)) == 0
            assert lib.rc_release_stream(dev.index or 0, ctypes.c_void_p(st.cuda_stream)) == 0      # nothing left: still fine
            ctrl = rand_ctrl(rng, 4, N)
            draws = 0.05 * rng.standard_normal((4, 300, N, 3))
            got = be.mc_fidelity(torch.from_numpy(ctrl).to(dev), torch.from_numpy(draws).to(dev), N, 1, 4, ring=True)
            st.synchronize()
            assert np.abs(got.cpu().numpy() - orc.fidelity_eigh(ctrl, draws, N, 1, 4, ring=True)).max() < TOL
            assert lib.rc_release_stream(dev.index or 0, ctypes.c_void_p(st.cuda_stream)) == 0
        st.synchronize()


def test_release_stream_wrapper_and_ring_stream_context(be):
    """`backend.release_stream` / `backend.ring_stream`: the library-side repair list of a side stream that ran ring launches is
    handed back (stream-ordered) when the stream is retired; results on the side stream equal the main stream's; releasing a
    stream that holds nothing is fine; more streams than the library's cap (16 per device) evict the least recently used."""
    import torch
    rng = np.random.default_rng(9)
    N, C, K = 6, 4, 300
    ctrl = rand_ctrl(rng, C, N)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    dev = torch.device("cuda", torch.cuda.current_device())
    ct, dt = torch.from_numpy(ctrl).to(dev), torch.from_numpy(draws).to(dev)
    want = be.mc_fidelity(ct, dt, N, 0, 3, ring=True)
    torch.cuda.synchronize()
    assert np.abs(want.cpu().numpy() - orc.fidelity_eigh(ctrl, draws, N, 0, 3, ring=True)).max() < TOL
    for _ in range(20):                                                # 20 short-lived streams: beyond the cap of 16
        with be.ring_stream() as st:
            st.wait_stream(torch.cuda.default_stream(dev))
            got = be.mc_fidelity(ct, dt, N, 0, 3, ring=True)
        st.synchronize()
        assert torch.equal(got, want)        # bit for bit: the repair kernel runs per-lane sweeps (its list is packed in arrival order)
    streams = [torch.cuda.Stream(dev) for _ in range(20)]              # ... and 20 that are never released
    for st in streams:
        with torch.cuda.stream(st):
            got = be.mc_fidelity(ct, dt, N, 0, 3, ring=True)
        st.synchronize()
        assert torch.equal(got, want)        # bit for bit: the repair kernel runs per-lane sweeps (its list is packed in arrival order)
    be.release_stream(streams[-1])
    be.release_stream(streams[-1])                                     # nothing left: still fine
    be.release_stream()                                                # the current stream
