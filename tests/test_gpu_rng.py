"""The random streams on the device: NumPy's legacy MT19937 / polar stream continued on the GPU (state AND normals bit-
identical), the interleaved randint / normal consumption of directional_perturbation, counter-based Philox draws as a
generator kernel and inside the fidelity kernel."""
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


# ----------------------------------------------------------------------------------------------------------------
# the reference's own RNG stream on the device
# ----------------------------------------------------------------------------------------------------------------
def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and tuple(a[2:]) == tuple(b[2:])


def test_philox_device_draws(be):
    """Counter-based device draws: regenerated element by element on the host (oracle/philox_host.py), any
    slice independently addressable, N(0,1) moments, and fidelities from device draws match the oracle fed with
    the SAME (copied back) draws."""
    import torch
    from oracle import philox_host
    seed, n = 0x1234ABCD5678EF01, 100003
    got = be.philox_normal((n,), seed, scale=0.05, offset=7)
    want = philox_host.philox_normal(seed, 7, n, 0.05)
    assert np.abs(got - want).max() < 1e-15
    part = be.philox_normal((1000,), seed, scale=0.05, offset=7 + 5000)
    assert np.array_equal(part, got[5000:6000])
    big = be.philox_normal((4_000_000,), 99, as_torch=True)
    assert abs(float(big.mean())) < 3e-3 and abs(float(big.std()) - 1) < 3e-3
    assert abs(float((big ** 4).mean()) - 3) < 5e-2
    N, C, K = 7, 5, 321
    rng = np.random.default_rng(0)
    ctrl = rand_ctrl(rng, C, N)
    d = be.philox_normal((C, K, N, 3), 2024, scale=0.05, as_torch=True)
    f = be.mc_fidelity(torch.from_numpy(ctrl).cuda(), d, N, 0, 6)
    want = orc.fidelity_eigh(ctrl, d.cpu().numpy(), N, 0, 6)
    assert np.abs(f.cpu().numpy() - want).max() < TOL


@pytest.mark.parametrize("seed,periods,period,skip", [(0, 1, 1, 0), (1, 1, 2, 0), (5, 1, 7, 1), (4, 3, 1001, 1),
                                                      (99, 11, 6241, 1), (7, 1, 100001, 0), (31337, 1000, 301, 1)])
def test_legacy_device_stream_vs_numpy(be, seed, periods, period, skip):
    """`rc_draws_legacy_f64`: NumPy's legacy normal stream continued on the GPU.  Generator state afterwards identical
    to NumPy's bit for bit (key, pos, has_gauss, cached value) - and (round 5) so are the NORMALS: the device evaluates the C
    library's log operation for operation (`backend.legacy_device_exact()`: verified against this host's log() by the library)."""
    rng = np.random.default_rng(seed)
    scales = rng.uniform(0.0, 0.2, periods)
    for prefix in (0, 1):                       # start with / without a cached normal
        np.random.seed(seed)
        if prefix:
            np.random.normal()
        got = be.legacy_normal_periods(periods, period, skip, scales).cpu().numpy()
        st = np.random.get_state()
        np.random.seed(seed)
        if prefix:
            np.random.normal()
        want = np.empty((periods, period - skip))
        for p in range(periods):
            z = np.random.normal(scale=scales[p], size=period)
            want[p] = z[skip:]
        assert _same_state(st, np.random.get_state())
        assert np.abs(got - want).max() <= 8 * np.finfo(float).eps * max(1e-300, np.abs(want).max())
        assert be.legacy_device_exact()                           # this image's glibc IS the one the device restates
        assert np.array_equal(got, want), float((got != want).mean())


def test_legacy_device_stream_paper_scale_multi_segment(be):
    """6.6e7 normals (the paper's four algorithms: 11 levels x 4 x 1000 x 100 x 15, + burns) = 1.7e8 raw words: crosses
    the 2^27-word segment boundary of the generator (carry block, attempts straddling segments, ranks continuing); state
    identical to NumPy's, values identical bit for bit, timing printed."""
    import time
    import torch
    noises = np.linspace(0, 0.1, 11)
    per = 4 * 1000 * 100 * 5 * 3
    np.random.seed(2022)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = be.legacy_normal_periods(11, 1 + per, 1, noises)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    st = np.random.get_state()
    np.random.seed(2022)
    t0 = time.perf_counter()
    want = np.empty((11, per))
    for j, s in enumerate(noises):
        np.random.normal(scale=s)
        want[j] = np.random.normal(scale=s, size=per)
    t_host = time.perf_counter() - t0
    assert _same_state(st, np.random.get_state())
    assert np.array_equal(got.cpu().numpy(), want)                # 6.6e7 normals, every one NumPy's own bits (round 5)
    print(f"legacy stream, {11 * per:.2e} normals: device {t_dev * 1e3:.1f} ms, numpy {t_host * 1e3:.1f} ms")


def test_directional_draws_on_the_device(be):
    """`rc_directional_draws_legacy_dev`: the interleaved randint / normal(size=2) consumption of
    `directional_perturbation.perturbation()` (noise_model.py:183-189) continued on the GPU - raw words from jump-ahead
    sub-streams, per-position sample lengths, host walk, emit - against NumPy ITSELF sample by sample (small n) and
    against the bit-identical host emulation (large n): indices identical, generator state identical (key, pos,
    has_gauss, cached value), normals identical too (round 5); entered with and without a cached normal; ndir with and
    without rejection, ndir = 1 (randint consumes nothing)."""
    import ctypes
    lib = importlib.import_module("code-robchar_amd._lib")

    def host(n, ndir, sigma):
        name, key, pos, has_gauss, cached = np.random.get_state()
        st = lib.Mt19937State()
        ctypes.memmove(st.key, np.ascontiguousarray(key, dtype=np.uint32).ctypes.data, 624 * 4)
        st.pos, st.has_gauss, st.gauss = int(pos), int(has_gauss), float(cached)
        idx, ab = np.empty(n, dtype=np.int32), np.empty((n, 2))
        assert lib.load().rc_directional_draws_legacy(ctypes.byref(st), n, ndir, sigma, ctypes.c_void_p(idx.ctypes.data),
                                                       ctypes.c_void_p(ab.ctypes.data)) == 0
        return idx, ab, (np.frombuffer(st.key, dtype=np.uint32).copy(), int(st.pos), int(st.has_gauss), float(st.gauss))

    for ndir, n, cached in ((19, 50, False), (19, 50, True), (1, 33, True), (4, 1000, False), (28, 20000, True),
                            (32, 4097, False), (33, 300000, True), (19, 1000000, False)):
        np.random.seed(1000 + ndir + n % 7)
        np.random.normal(size=3 if cached else 4)            # odd count: the generator holds a cached normal
        state0 = np.random.get_state()
        assert bool(state0[3]) == cached
        if n <= 1000:                                         # NumPy itself, call by call
            want_idx, want_ab = np.empty(n, dtype=np.int64), np.empty((n, 2))
            for i in range(n):
                want_idx[i] = np.random.randint(low=0, high=ndir)
                want_ab[i] = np.random.normal(scale=0.05, size=2)
            want_state = np.random.get_state()
            want_state = (want_state[1], want_state[2], want_state[3], want_state[4])
        else:
            want_idx, want_ab, want_state = host(n, ndir, 0.05)
        np.random.set_state(state0)
        idx, ab = be.directional_draws_device(n, ndir, 0.05)
        got_state = np.random.get_state()
        assert np.array_equal(idx.cpu().numpy(), want_idx), (ndir, n)
        assert np.array_equal(got_state[1], want_state[0]) and got_state[2] == want_state[1], (ndir, n)
        assert got_state[3] == want_state[2] and got_state[4] == want_state[3], (ndir, n)
        assert be.legacy_device_exact()
        assert np.array_equal(ab.cpu().numpy(), want_ab), (ndir, n)      # (round 5) the normals too: glibc's log on the device


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
