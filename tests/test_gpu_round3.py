"""GPU tests (``-m gpu``) of the round-3 changes to the chain kernels' rare paths: close eigenvalue pairs stay on the
wave-wide route down to gaps of 1e-12 of the spectral scale, truly degenerate samples are repaired in registers (rows-mode
QL for the bad lanes only), and what either costs is bounded."""
import importlib

import numpy as np
import pytest

from oracle import robchar_oracle as orc
from test_host_core import _close_pair_matrix

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


def _inject(ctrl_row, draws_row, d, e):
    """Make one sample's matrix exactly tridiag(d, e): g0 = d - x, g1 = e - 1, g2 = 0 (h0 = 0, J = 1)."""
    N = len(d)
    draws_row[:, 0] = d - ctrl_row[:N]
    draws_row[1:, 1] = e - 1.0
    draws_row[:, 2] = 0.0


@pytest.mark.parametrize("N", [5, 7, 10, 13])
def test_close_pairs_stay_on_the_wave_wide_route(be, N):
    """Samples whose spectrum has a pair 1e-3 ... 1e-10 apart WITH O(1) weights on both members (Jacobi matrices with a
    prescribed spectrum, injected through the draws).  Round 2 sent everything closer than 1e-7 of the scale to the
    per-sample general routine (a ~100 us single-lane straggler at N >= 10); now the tile-wide all-fp64 QL (tight split
    tolerance) + product-formula weights carry them for the END-TO-END weights (no general-path tile); the general adjugate
    weights have a 4e-6 threshold (their numerators are recurrences evaluated beside their own roots - the fuzz campaign of
    round 3) and hand such samples to the in-register eigenvector repair.  Parity 1e-10 either way."""
    rng = np.random.default_rng(900 + N)
    C, K = 4, 256
    ctrl = rand_ctrl(rng, C, N)
    ctrl[:, N] = rng.uniform(15, 30, C)
    draws = 0.02 * rng.standard_normal((C, K, N, 3))
    gaps = []
    for c in range(C):
        for k in range(0, K, 7):                       # several per tile, most tiles
            delta = 10.0 ** rng.uniform(-10, -3)
            d, e, true, j = _close_pair_matrix(N, delta, rng)
            shift = rng.uniform(-3, 3)
            _inject(ctrl[c], draws[c, k], d + shift, e)
            gaps.append(delta)
    for (a, b) in ((0, N - 1), (0, N // 2), (N - 2, 1)):
        be.general_path_tiles(reset=True)
        be.polish_tiles(reset=True)
        got = be.mc_fidelity(ctrl, draws, N, a, b)
        ref = orc.fidelity_eigh(ctrl, draws, N, a, b)
        assert np.abs(got - ref).max() < TOL, (N, a, b, np.abs(got - ref).max())
        assert be.polish_tiles() > 0
        if (a, b) == (0, N - 1):
            assert be.general_path_tiles() == 0, (N, a, b)          # end-to-end weights: wave-wide down to 1e-12 of the scale
        else:
            assert be.general_path_tiles() > 0                      # general adjugate weights: below 4e-6 the eigenvector route
    assert min(gaps) < 1e-8


def test_fuzz_regression_cut_chain_near_degenerate(be):
    """The inputs the round-3 fuzz campaign failed on (see tests/test_host_core.py, same fixture) on the GPU: every chain
    kernel, 1e-10."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_r3_adjugate_cut_chain.npz"))
    for name in z["names"]:
        name = str(name)
        N, a, b = (int(v) for v in z[name + "_meta"])
        ctrl, draws = z[name + "_ctrl"], z[name + "_draws"]
        h0 = z[name + "_h0"] if z[name + "_h0"].size else None
        want = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0)
        for kern in ("auto", "tridiag_adj", "tridiag_ql"):
            got = be.mc_fidelity(ctrl, draws, N, a, b, h0_diag=h0, kernel=kern)
            assert np.abs(got - want).max() < TOL, (name, kern, np.abs(got - want).max())


def _degenerate_sample(ctrl, draws, c, k, N):
    """Mirror-symmetric controller + draws, chain cut in the middle: the two halves have the SAME spectrum (exactly
    degenerate pairs) - the eigenvalue-only weight formulas cannot serve it."""
    cut = N // 2
    draws[c, k, :, 0] = 0.0
    draws[c, k, cut, 1] = -1.0
    draws[c, k, cut, 2] = 0.0
    if N % 2:                                            # odd N: the middle site is cut off on both sides
        draws[c, k, cut + 1, 1] = -1.0
        draws[c, k, cut + 1, 2] = 0.0
    for i in range(1, cut):
        draws[c, k, N - i, 1:] = draws[c, k, i, 1:]


@pytest.mark.parametrize("N,xxz", [(10, True), (7, False)])
def test_rare_path_cost_is_bounded(be, N, xxz):
    """A 1e6-evaluation launch in which ONE sample is exactly degenerate must cost at most 1.3x the clean launch (round 2:
    the per-sample LDS routine ran ~100 us at N = 10 - longer than the whole launch; measured now: 120.4 -> 120.8 us).
    Also a launch with a degenerate sample in one tile of a hundred (158 of 15 700 tiles; measured 1.33x at N = 10 - each
    such tile runs the fp32 QL, the stepping attempts, the tile-wide fp64 QL and then the rows-mode QL for its one lane):
    bounded by 1.6x."""
    import torch
    rng = np.random.default_rng(77 + N)
    C, K = 100, 10000
    ctrl = rand_ctrl(rng, C, N)
    cut = N // 2
    ctrl[3, N - cut:N] = ctrl[3, :cut][::-1]             # controller 3 is mirror-symmetric
    ctrl[:, N - cut:N][::2] = ctrl[:, :cut][::2, ::-1]    # ... and so is every second one
    h0 = orc.xxz_delta(N) if xxz else None                # (the XXZ offsets are mirror-symmetric themselves)
    clean = 0.05 * rng.standard_normal((C, K, N, 3))
    one = clean.copy()
    _degenerate_sample(ctrl, one, 2, 4711, N)
    many = clean.copy()
    tiles_per_ctrl = (K + 63) // 64
    hit = []
    for t in range(0, C * tiles_per_ctrl, 50):
        c, k = divmod(t, tiles_per_ctrl)
        if c % 2 == 0:
            _degenerate_sample(ctrl, many, c, k * 64 + 5, N)
            hit.append((c, k * 64 + 5))
    dev = be.compute_device()
    ct = torch.from_numpy(ctrl).to(dev)
    tens = {name: torch.from_numpy(x).to(dev) for name, x in (("clean", clean), ("one", one), ("many", many))}
    out = torch.empty((C, K), dtype=torch.float64, device=dev)

    def run(name):
        return be.mc_fidelity(ct, tens[name], N, 0, N - 1, h0_diag=h0, out=out)

    # results first: the degenerate samples are right (transfer across a cut: 0) and really took the repair path
    be.general_path_tiles(reset=True)
    got = run("one").cpu().numpy()
    assert be.general_path_tiles() == 1
    sel = np.arange(4700, 4730)
    assert np.abs(got[2, sel] - orc.fidelity_eigh(ctrl[2:3], one[2:3, sel], N, 0, N - 1, h0_diag=h0)[0]).max() < TOL
    assert got[2, 4711] < 1e-20
    be.general_path_tiles(reset=True)
    got = run("many").cpu().numpy()
    assert be.general_path_tiles() == len(hit)
    for (c, k) in hit[:: max(1, len(hit) // 20)]:
        assert got[c, k] < 1e-20
    # cost: medians of interleaved launches (HIP events), after a settling burst
    for _ in range(300):
        run("clean")
    torch.cuda.synchronize()
    times = {"clean": [], "one": [], "many": []}
    for rep in range(40):
        for name in times:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                run(name)
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b) / 4)
    med = {k: float(np.median(v)) for k, v in times.items()}
    print(f"N={N}: clean {med['clean'] * 1e3:.1f} us, one degenerate sample {med['one'] * 1e3:.1f} us, "
          f"{len(hit)} degenerate samples {med['many'] * 1e3:.1f} us")
    assert med["one"] <= 1.3 * med["clean"], med
    assert med["many"] <= 1.6 * med["clean"], med


@pytest.mark.parametrize("N", [3, 5, 7, 10])
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
    assert be.general_path_tiles() <= 0.15 * 5 * (C - 1) * K / 64 + 5       # repaired waves of 64 samples, 5 launches
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


@pytest.mark.parametrize("rng_mode", ["legacy", "philox"])
def test_mcdatasim_level_batching_is_transparent(tmp_path, monkeypatch, rng_mode):
    """`MCDataSim._run_algo` sends all sigma levels of an algorithm through ONE fidelity launch when one process makes the
    draws on the device; the level-by-level route (a buffer cap of zero) and the route with several generator calls per
    algorithm (a draw cap below one level pair) must give the same fidelities bit for bit and leave NumPy's stream at the
    same position (mcsim.py:422-460: noise outer, controller middle, draw inner, one burned draw per level)."""
    import importlib, json, os
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    monkeypatch.chdir(tmp_path)
    N, C, K = 5, 37, 24
    rng = np.random.default_rng(3)
    le = {}
    for a in ("ppo", "lbfgs"):
        x = np.empty((C if a == "ppo" else C - 5, N + 1))          # lbfgs: fewer controllers than asked for -> NaN rows
        x[:, :N] = rng.uniform(-10, 10, x[:, :N].shape)
        x[:, N] = rng.uniform(2, 30, x.shape[0])
        le[a] = {("%d" % N if a == "lbfgs" else "0.05"): {"controller": x.tolist()}}
    noises = np.linspace(0, 0.1, 6)
    results = []
    for tag, caps in (("batched", {}), ("level_by_level", {"_BATCH_LEVELS_MAX_BYTES": 0}),
                      ("several_generator_calls", {"_LEGACY_DEVICE_MAX_DRAWS": 2 * C * K * N * 3 + 7})):
        os.makedirs(f"experiments/{tag}")
        json.dump(le, open(f"experiments/{tag}/ppo_spin_{N}_0-2_c_{C}", "w"))
        for k, v in caps.items():
            monkeypatch.setattr(mcmod.MCDataSim, k, v)
        np.random.seed(11)
        sim = mcmod.MCDataSim(experiment_name=tag, Nspin=N, inspin=0, outspin=2, noises=noises, bootreps=K,
                              training_noise=0.05, numcontrollers=C, verbose=False, rng_mode=rng_mode, seed=5,
                              cache_format="json")
        fids = sim.get_fid_dists()
        results.append(({a: np.array(fids[a], dtype=float) for a in fids}, np.random.normal()))
        monkeypatch.undo()
        monkeypatch.chdir(tmp_path)
    ref, pos = results[0]
    assert set(ref) == {"ppo", "lbfgs"} and ref["ppo"].shape == (6, C, K) and np.isnan(ref["lbfgs"][:, C - 5:]).all()
    for got, p in results[1:]:
        assert p == pos
        for a in ref:
            assert np.array_equal(got[a], ref[a], equal_nan=True), a
