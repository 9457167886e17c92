"""GPU tests (``-m gpu``) of the round-4 changes: stream hygiene of the ring route (stream-ordered repair list,
rc_reserve_ring / rc_release_stream), the a-posteriori sum-rule guard on the device, the complex symmetric route's
conditioning guard."""
import ctypes
import importlib
import os

import numpy as np
import pytest

from oracle import robchar_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def be():
    mod = importlib.import_module("code-robchar_amd.backend")
    lib = importlib.import_module("code-robchar_amd._lib")
    assert lib.require_gpu() >= 1
    return mod


def rand_ctrl(rng, C, N):
    x = np.empty((C, N + 1))
    x[:, :N] = rng.uniform(-10, 10, (C, N))
    x[:, N] = rng.uniform(2, 30, C)
    return x


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


def test_sum_rule_guard_on_the_device(be):
    """The a-posteriori guard (tridiag_core.h: kSumRuleGuard) inside the kernels.  (1) It must not send healthy samples to the
    eigenvector route: the benchmark-style workloads of every weight mode and (in, out) class - same site, neighbours, two
    apart, far apart, end to end - stay (almost) off the repair path.  (2) The configuration in which the round-3 fuzz
    campaign found its worst chain error (2.6e-11: N = 6, in = out = 1, |bias| <= 1, a weak bond between mirror halves -
    recurrence noise in the adjugate numerators next to a pair just above the 4e-6 hand-over threshold) now comes out at
    a few 1e-12: the guard catches what the threshold let through."""
    rng = np.random.default_rng(77)
    for N in (5, 7, 10, 13):
        C, K = 40, 640
        ctrl = rand_ctrl(rng, C, N)
        draws = 0.05 * rng.standard_normal((C, K, N, 3))
        be.general_path_tiles(reset=True)
        pairs = [(0, N - 1), (0, 0), (N // 2, N // 2), (N - 1, N - 1), (0, 1), (N // 2, N // 2 - 1), (0, 2), (1, N - 2), (N - 1, 0)]
        for (a, b) in pairs:
            got = be.mc_fidelity(ctrl, draws, N, a, b)
            want = orc.fidelity_eigh(ctrl, draws, N, a, b)
            assert np.abs(got - want).max() < 1e-11, (N, a, b, np.abs(got - want).max())
        tiles = len(pairs) * C * (K // 64)
        rep = be.general_path_tiles()
        print(f"guard, N = {N}: {rep} of {tiles} tiles with a repaired sample on random controllers")
        assert rep <= 0.01 * tiles + 2, (N, rep, tiles)
    # the round-3 worst case, regenerated (scripts/fuzz_parity.py, seed 2150, configuration 17: N = 6, a = b = 1)
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from host_fuzz_guard import configs
    for meta, ctrl, draws, h0 in configs(2150, 18):
        if meta["it"] != 17:
            continue
        assert (meta["N"], meta["a"], meta["b"]) == (6, 1, 1)
        want = orc.fidelity_eigh(ctrl, draws, 6, 1, 1, h0_diag=h0)
        for kern in ("auto", "tridiag_adj"):
            err = np.abs(be.mc_fidelity(ctrl, draws, 6, 1, 1, h0_diag=h0, kernel=kern) - want).max()
            print(f"round-3 worst fuzz case ({kern}): max |dF| = {err:.2e} (round 3: 2.57e-11)")
            assert err < 1e-11, err


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
        assert torch.equal(got, want)
    streams = [torch.cuda.Stream(dev) for _ in range(20)]              # ... and 20 that are never released
    for st in streams:
        with torch.cuda.stream(st):
            got = be.mc_fidelity(ct, dt, N, 0, 3, ring=True)
        st.synchronize()
        assert torch.equal(got, want)
    be.release_stream(streams[-1])
    be.release_stream(streams[-1])                                     # nothing left: still fine
    be.release_stream()                                                # the current stream


def test_directional_device_walk_equals_host_walk(be):
    """The sample chain of `directional_perturbation`'s RNG consumption walked ON THE DEVICE (round 4: block / superblock
    composition of entry -> exit maps, k_draws.inc.h) against the host walk of round 3 (RC_DIR_WALK=host) and against the
    hand-over from a device pass that is declared failed (RC_DIR_WALK=fallback): indices, normals and generator state
    identical bit for bit - same kernels for the values, only the walk differs -, for sample counts around the block
    (2048 positions) and superblock (64 / 256 blocks) boundaries, with and without a cached normal on entry, ndir with and
    without rejection."""
    import os
    cases = ((19, 1, False), (19, 7, True), (19, 320, False), (19, 330, True), (1, 512, False), (1, 513, True), (4, 5000, False),
             (28, 82000, True), (19, 84000, False), (33, 250000, True), (19, 1000000, False),
             (19, 9000000, True))             # (more than one pass of 2^23 samples: 256-block superblocks, state carried over)
    try:
        for ndir, n, cached in cases:
            out = {}
            for mode in ("device", "host", "fallback"):
                os.environ["RC_DIR_WALK"] = mode
                np.random.seed(4000 + ndir + n % 11)
                np.random.normal(size=3 if cached else 4)
                idx, ab = be.directional_draws_device(n, ndir, 0.05)
                st = np.random.get_state()
                out[mode] = (idx.cpu().numpy(), ab.cpu().numpy(), st[1].copy(), st[2], st[3], st[4])
            for mode in ("host", "fallback"):
                for x, y in zip(out["device"], out[mode]):
                    assert np.array_equal(x, y), (ndir, n, cached, mode)
    finally:
        os.environ.pop("RC_DIR_WALK", None)


def test_settled_lanes_keep_their_polished_eigenvalues(be):
    """Regression fixture of the round-4 fuzz campaign (seeds 4020 / 4092 / 4117, |bias| ~ 100, |T| ~ 70 .. 95: the worst chain
    cases, 1.05e-11 .. 1.41e-11): healthy samples whose 64-sample TILE took the tile-wide all-fp64 QL because of a neighbour
    had their polished eigenvalues replaced by QL eigenvalues (error ~ N eps scale = 2e-13 absolute - a phase error of 1e-11
    at that T).  Settled lanes now keep what they had: every tile of the fixture (input = one controller row + the tile's 64
    samples, so the wave composition is the campaign's) comes out below 5e-12 in both eigenvalue-only weight modes."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_r4_tile_fallback.npz"))
    keys = sorted(k[:-5] for k in d.files if k.endswith("_ctrl"))
    assert len(keys) == 15
    worst = 0.0
    for key in keys:
        N, a, b = (int(v) for v in d[key + "_meta"])
        h0 = d[key + "_h0"] if d[key + "_h0"].size else None
        ctrl, draws, want = d[key + "_ctrl"], d[key + "_draws"], d[key + "_want"]
        assert np.abs(orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0) - want).max() < 1e-13     # the fixture's own oracle values
        before = np.abs(d[key + "_gpu_round4_before"] - want).max()
        for kern in ("auto", "tridiag_adj"):
            err = np.abs(be.mc_fidelity(ctrl, draws, N, a, b, h0_diag=h0, kernel=kern) - want).max()
            worst = max(worst, err)
            assert err < 5e-12, (key, kern, err, before)
    print(f"round-4 fuzz worst tiles: max |dF| = {worst:.2e} (before: up to 1.41e-11)")


@pytest.mark.parametrize("N", list(range(2, 17)))
def test_philox_draws_inside_the_fidelity_kernel(be, N):
    """rc_mc_fidelity_philox_f64_async: the counter-based draws generated where they are consumed.  BIT-IDENTICAL to the
    two-kernel route (rc_draws_philox_f64_async -> rc_mc_fidelity_f64_async) - same generator routine, same per-sample
    arithmetic - for every N, both eigenvalue-only weight modes, odd and even stream offsets (the Box-Muller pair grid
    straddles samples), ragged K, a NaN-padded controller, one scale per controller row; against the oracle on host-regenerated
    elements (oracle/philox_host.py) as well."""
    import torch
    from oracle import philox_host
    rng = np.random.default_rng(3000 + N)
    dev = torch.device("cuda", torch.cuda.current_device())
    for (a, b) in ((0, N - 1), (N // 2, 0)):
        for (C, K, off) in ((3, 130, 0), (4, 64, 7), (2, 257, 123456789012345)):
            ctrl = rand_ctrl(rng, C, N)
            if C == 4:
                ctrl[2] = np.nan
            ct = torch.from_numpy(ctrl).to(dev)
            seed = int(rng.integers(1, 2 ** 40))
            draws = be.philox_normal((C, K, N, 3), seed, scale=0.05, offset=off, device=dev, as_torch=True)
            want = be.mc_fidelity(ct, draws, N, a, b)
            got = be.mc_fidelity_philox(ct, K, N, a, b, seed, offset=off, sigma=0.05)
            assert torch.equal(torch.isnan(got), torch.isnan(want))
            assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want)), (N, a, b, C, K, off)
            # the oracle on host-regenerated elements
            host = philox_host.philox_normal(seed, off, C * K * N * 3, 0.05).reshape(C, K, N, 3)
            ref = orc.fidelity_eigh(ctrl, host, N, a, b)
            assert np.nanmax(np.abs(got.cpu().numpy() - ref)) < TOL
            # one scale per controller row (all sigma levels of an algorithm in one launch)
            sig = torch.tensor(rng.uniform(0.0, 0.1, C), device=dev)
            got_r = be.mc_fidelity_philox(ct, K, N, a, b, seed, offset=off, sigma=sig)
            for c in range(C):
                w = be.mc_fidelity(ct[c:c + 1], be.philox_normal((1, K, N, 3), seed, scale=float(sig[c]), offset=off + c * K * N * 3,
                                                                device=dev, as_torch=True), N, a, b)
                assert torch.equal(torch.nan_to_num(got_r[c:c + 1]), torch.nan_to_num(w)), (N, c)


@pytest.mark.parametrize("N", [14, 15, 16])
def test_philox_fused_kernel_follows_the_chain_kernels_weight_mode(be, N):
    """End-to-end pairs at N = 15, 16 run the GENERAL adjugate instantiation (two waves per SIMD; the end-to-end one needs more
    than 256 registers there) - in the fused kernel as in the chain kernel, or the two routes would hand different samples to the
    eigenvector repair (hand-over thresholds 1e-7 / 4e-6 of the scale) and stop being bit-identical.  8e4 samples: ~1e-3 of them
    have a pair between the two thresholds.  And the routing rule: the fused kernel is chosen where it is the faster route."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(77 + N)
    C, K = 4, 20000
    ctrl = rand_ctrl(rng, C, N)
    ct = torch.from_numpy(ctrl).to(dev)
    draws = be.philox_normal((C, K, N, 3), 5, scale=0.05, offset=3, device=dev, as_torch=True)
    be.general_path_tiles(reset=True)
    want = be.mc_fidelity(ct, draws, N, 0, N - 1)
    repaired = be.general_path_tiles()
    got = be.mc_fidelity_philox(ct, K, N, 0, N - 1, 5, offset=3, sigma=0.05)
    assert torch.equal(got, want), N
    print(f"N = {N}: {repaired} tiles with a repaired sample in the two-kernel route, fused result identical")
    assert be.philox_fused_pays(13, 2, 7) and be.philox_fused_pays(14, 0, 13) and be.philox_fused_pays(14, 13, 0)
    assert not be.philox_fused_pays(14, 0, 7) and not be.philox_fused_pays(15, 0, 14) and not be.philox_fused_pays(16, 3, 9)


def test_philox_fused_kernel_repairs_degenerate_lanes(be):
    """The rare paths of the fused kernel regenerate their draws element by element: a controller whose end sites sit at the
    same energy with sigma = 0 (every sample exactly degenerate when both end bonds are cut is not reachable through random
    draws, so: sigma = 0 and a mirror-symmetric controller -> every lane of every tile takes the eigenvector repair)."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    N = 6
    x = np.array([[1.0, -2.0, 0.5, 0.5, -2.0, 1.0, 7.0]])                  # mirror-symmetric: degenerate pairs only if decoupled
    ct = torch.from_numpy(x).to(dev)
    be.general_path_tiles(reset=True)
    got = be.mc_fidelity_philox(ct, 200, N, 1, 4, seed=5, sigma=0.0)
    want = orc.fidelity_eigh(x, np.zeros((1, 200, N, 3)), N, 1, 4)
    assert np.abs(got.cpu().numpy() - want).max() < TOL
    # flat diagonal + zero noise: a chain with a flat diagonal has distinct levels; make two levels coincide instead via a cut bond
    h0o = np.ones(N - 1)
    h0o[2] = 0.0                                                              # chain cut in the middle: two identical halves
    got = be.mc_fidelity_philox(ct, 200, N, 1, 1, seed=5, sigma=0.0, h0_offdiag=h0o)
    want = orc.fidelity_eigh(x, np.zeros((1, 200, N, 3)), N, 1, 1, h0_offdiag=h0o)
    assert np.abs(got.cpu().numpy() - want).max() < TOL
    assert be.general_path_tiles() >= 4                                       # every tile of the second launch was repaired


def test_mcdatasim_philox_fused_route_equals_draw_tensor_route(tmp_path, monkeypatch):
    """`MCDataSim(rng_mode="philox")` now generates its draws inside the fidelity kernel (all sigma levels of an algorithm in one
    launch, one scale per tiled controller row); ROBCHAR_PHILOX_FUSED=0 takes the round-3 route (draw tensor + fidelity
    kernel, level batching).  Same stream elements, same arithmetic: the (L, C, K) tensors - NaN rows of a short controller
    list included -, the metric rows and NumPy's stream position (the burned draw per level, mcsim.py:425) are identical."""
    import importlib, json, os
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    monkeypatch.chdir(tmp_path)
    N, C, K = 7, 41, 130
    rng = np.random.default_rng(8)
    le = {}
    for a in ("ppo", "lbfgs"):
        x = np.empty((C if a == "ppo" else C - 6, N + 1))
        x[:, :N] = rng.uniform(-10, 10, x[:, :N].shape)
        x[:, N] = rng.uniform(2, 30, x.shape[0])
        le[a] = {("%d" % N if a == "lbfgs" else "0.05"): {"controller": x.tolist()}}
    noises = np.linspace(0, 0.1, 5)
    res = {}
    for tag, flag in (("fused", "1"), ("tensor", "0")):
        os.makedirs(f"experiments/{tag}")
        json.dump(le, open(f"experiments/{tag}/ppo_spin_{N}_0-3_c_{C}", "w"))
        monkeypatch.setenv("ROBCHAR_PHILOX_FUSED", flag)
        np.random.seed(21)
        sim = mcmod.MCDataSim(experiment_name=tag, Nspin=N, inspin=0, outspin=3, noises=noises, bootreps=K, training_noise=0.05,
                              numcontrollers=C, verbose=False, rng_mode="philox", seed=9, cache_format="json")
        fids = sim.get_fid_dists()
        met = sim.get_metrics_dict()
        res[tag] = ({a: np.array(fids[a], dtype=float) for a in fids}, met, np.random.normal())
    (f1, m1, p1), (f0, m0, p0) = res["fused"], res["tensor"]
    assert p1 == p0 and set(f1) == {"ppo", "lbfgs"}
    for a in f1:
        assert f1[a].shape == (5, C, K) and np.array_equal(f1[a], f0[a], equal_nan=True), a
        for k in m1[a]:
            assert np.array_equal(np.array(m1[a][k], dtype=float), np.array(m0[a][k], dtype=float), equal_nan=True), (a, k)


def test_mcdatasim_random_configurations_product_fuzz():
    """scripts/fuzz_mcdatasim.py, a short block of it: random `MCDataSim` calls - the GPU against the oracle-backed host route, the
    host-drawn against the device-continued legacy stream (generator state identical), the fused against the draw-tensor Philox route
    (bit for bit), the single-process multi-device route with 1 / 2 / 3 listed devices (identical)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEED="3:4", NCFG="8", FUZZ_DEVICES="1", RC_ALLOW_DUPLICATE_DEVICES="1")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_mcdatasim.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "16 random MCDataSim configurations x 10 runs" in r.stdout, r.stdout[-2000:]
    print(r.stdout.strip().splitlines()[-6:])


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
