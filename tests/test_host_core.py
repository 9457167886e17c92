"""The exact per-sample arithmetic of the HIP kernel (code-robchar_amd/csrc/tridiag_core.h) compiled for the
host with g++ and checked against the oracle and the golden vectors - runs without a GPU.  The library built
here is a TEST HARNESS (tests/host/host_core.cpp); the product never loads it."""
import ctypes
import importlib
import os
import subprocess

import numpy as np
import pytest

from oracle import robchar_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    out = tmp_path_factory.mktemp("hostcore") / "librc_hosttest.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(out),
                    os.path.join(ROOT, "tests", "host", "host_core.cpp")], check=True)
    lib = ctypes.CDLL(str(out))
    lib.rc_host_general_calls.restype = ctypes.c_longlong

    def fid(ctrl, draws, N, a, b, h0d=None, vec=False):
        lib.rc_host_set_variant({False: 0, True: 1, "adj": 2}[vec])
        C, K = draws.shape[:2]
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        h0d = np.zeros(N) if h0d is None else np.ascontiguousarray(h0d, dtype=np.float64)
        h0o = np.ones(max(N - 1, 1))
        res = np.empty((C, K))
        rc = lib.rc_host_chain_fidelity(N, ctrl.ctypes.data_as(P), h0d.ctypes.data_as(P), h0o.ctypes.data_as(P),
                                        draws.ctypes.data_as(P), ctypes.c_longlong(C), ctypes.c_longlong(K),
                                        a, b, res.ctypes.data_as(P))
        assert rc == 0
        return res
    fid.general_calls = lib.rc_host_general_calls

    def general(ctrl, draws, N, a, b):
        """Direct entry to the general per-sample routine (any N <= 32)."""
        C, K = draws.shape[:2]
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        h0d, h0o = np.zeros(N), np.ones(max(N - 1, 1))
        res = np.empty((C, K))
        lib.rc_host_chain_fidelity_general(N, ctrl.ctypes.data_as(P), h0d.ctypes.data_as(P), h0o.ctypes.data_as(P),
                                           draws.ctypes.data_as(P), ctypes.c_longlong(C), ctypes.c_longlong(K),
                                           a, b, res.ctypes.data_as(P))
        return res
    fid.general = general

    def merge_sort_row(row):
        row = np.ascontiguousarray(row, dtype=np.float64)
        res = np.empty_like(row)
        assert lib.rc_host_merge_sort_row(row.ctypes.data_as(P), ctypes.c_longlong(row.size), res.ctypes.data_as(P)) == 0
        return res
    fid.merge_sort_row = merge_sort_row
    fid.lib_path = str(out)
    return fid


def test_core_vs_golden(host, kernel_cases):
    worst = 0.0
    for case in kernel_cases:
        if case["mode"] == "ring":
            continue
        h0 = orc.xxz_delta(case["N"]) if case["mode"] == "xxz" else None
        for s in range(case["draws"].shape[0]):
            got = host(case["ctrl"], case["draws"][s], case["N"], case["inspin"], case["outspin"], h0)
            worst = max(worst, np.abs(got - case["fid"][s]).max())
    assert worst < 1e-11, worst


@pytest.mark.parametrize("N", [2, 3, 5, 7, 10, 16])
def test_core_vs_oracle_random(host, N):
    rng = np.random.default_rng(N)
    C, K = 20, 50
    ctrl = np.empty((C, N + 1))
    ctrl[:, :N] = rng.uniform(-10, 10, (C, N))
    ctrl[:, N] = rng.uniform(2, 30, C)
    ctrl[:3, :N] = rng.uniform(-1e-7, 1e-7, (3, N))          # clustered eigenvalues
    ctrl[3, :N] = 0.0                                        # exactly uniform chain
    draws = 0.1 * rng.standard_normal((C, K, N, 3))
    draws[:, :4] = 0
    got = host(ctrl, draws, N, 0, N - 1)
    want = orc.fidelity_eigh(ctrl, draws, N, 0, N - 1)
    assert np.abs(got - want).max() < 1e-11


def test_core_extreme_inputs(host):
    """Decoupled sites (coupling draws cancel J exactly), huge biases, zero time."""
    N = 6
    ctrl = np.zeros((4, N + 1))
    ctrl[:, N] = [0.0, 5.0, 30.0, 17.0]
    ctrl[2, :N] = [1e3, -1e3, 5e2, 0, 1, 2]
    draws = np.zeros((4, 3, N, 3))
    draws[1, :, 3, 1] = -1.0           # e_2 = |1 + (-1)| = 0: chain cut between sites 2 and 3
    got = host(ctrl, draws, N, 0, 5)
    want = orc.fidelity_eigh(ctrl, draws, N, 0, 5)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() < 1e-11
    assert np.abs(got[0]).max() < 1e-28 and np.abs(got[1]).max() < 1e-28


@pytest.mark.parametrize("vec", [False, True, "adj"])
def test_both_weight_variants(host, vec, kernel_cases):
    """Eigenvector rows accumulated through the QL sweeps (vec) vs. the adjugate formula from eigenvalues only:
    both against the golden vectors and nasty spectra (near-degenerate, resonant ends, mirror-symmetric)."""
    worst = 0.0
    for case in kernel_cases:
        if case["mode"] == "ring":
            continue
        h0 = orc.xxz_delta(case["N"]) if case["mode"] == "xxz" else None
        for s in range(case["draws"].shape[0]):
            got = host(case["ctrl"], case["draws"][s], case["N"], case["inspin"], case["outspin"], h0, vec=vec)
            worst = max(worst, np.abs(got - case["fid"][s]).max())
    assert worst < 1e-11, worst
    rng = np.random.default_rng(5)
    for N in (5, 7, 10, 16):
        C, K = 12, 30
        ctrl = np.empty((C, N + 1))
        ctrl[:, :N] = rng.uniform(-10, 10, (C, N))
        ctrl[:, N] = rng.uniform(2, 30, C)
        ctrl[0:3, N - 1] = ctrl[0:3, 0] + np.array([1e-9, 1e-12, 0.0])       # resonant ends, tiny splitting
        ctrl[3:6, :N] = (ctrl[3:6, :N] + ctrl[3:6, N - 1::-1]) / 2             # mirror-symmetric bias
        ctrl[6, :N] = 0.0
        draws = 0.05 * rng.standard_normal((C, K, N, 3))
        draws[:, :5] = 0.0
        for (a, b) in ((0, N - 1), (0, N // 2), (N // 2, N // 2), (N - 2, 1)):
            got = host(ctrl, draws, N, a, b, vec=vec)
            want = orc.fidelity_eigh(ctrl, draws, N, a, b)
            assert np.abs(got - want).max() < 1e-11, (N, a, b, vec)


@pytest.mark.parametrize("N", [3, 9, 17, 24, 32])
def test_general_routine_any_N(host, N):
    """The general per-sample routine (the kernel's rare path, and the whole kernel for 16 < N <= 32)."""
    rng = np.random.default_rng(3000 + N)
    C, K = 5, 40
    ctrl = np.empty((C, N + 1))
    ctrl[:, :N] = rng.uniform(-10, 10, (C, N))
    ctrl[:, N] = rng.uniform(2, 30, C)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    draws[0, ::3, N // 2, 1] = -1.0                   # cut chains
    draws[0, ::3, N // 2, 2] = 0.0
    for (a, b) in ((0, N - 1), (1, N // 2), (N - 1, N - 1)):
        got = host.general(ctrl, draws, N, a, b)
        assert np.abs(got - orc.fidelity_eigh(ctrl, draws, N, a, b)).max() < 1e-11, (N, a, b)


@pytest.mark.parametrize("K", [1, 15, 16, 17, 100, 1000, 4097, 10000, 16384])
def test_merge_path_row_sort_schedule(host, K):
    """The merge-path index logic of the row sort (csrc/sort_core.h), executed thread by thread on the host: ragged
    lengths, many ties, already sorted / reversed input."""
    rng = np.random.default_rng(K)
    for row in (rng.random(K), np.round(rng.random(K), 1), np.sort(rng.random(K)), np.sort(rng.random(K))[::-1].copy(),
                np.zeros(K)):
        assert np.array_equal(host.merge_sort_row(row), np.sort(row))


# ----------------------------------------------------------------------------------------------------------------
# NumPy's legacy normal stream through legacy_rng_core.h (the header of the device kernels), on the host
# ----------------------------------------------------------------------------------------------------------------
def _legacy_host(lib, n_periods, period, skip, scales):
    st = np.random.get_state()
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos, hg, g = ctypes.c_int(int(st[2])), ctypes.c_int(int(st[3])), ctypes.c_double(float(st[4]))
    out = np.full(n_periods * (period - skip), np.nan)
    sc = np.ascontiguousarray(scales, dtype=np.float64)
    rc = lib.rc_host_legacy_normals(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), ctypes.byref(hg),
                                    ctypes.byref(g), ctypes.c_longlong(n_periods), ctypes.c_longlong(period),
                                    ctypes.c_longlong(skip), sc.ctypes.data_as(P), out.ctypes.data_as(P))
    assert rc == 0
    return out, ("MT19937", key, pos.value, hg.value, g.value)


def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


@pytest.mark.parametrize("seed,n", [(0, 1), (1, 2), (12345, 7), (4, 1000), (99, 6241), (7, 100001)])
def test_legacy_stream_core_equals_numpy(host, seed, n):
    """uint32 stream, tempering, attempt boundaries, cached second normal and the final generator state are
    BIT-identical to NumPy's legacy `RandomState` (host libm on both sides, so the normals are bit-identical too)."""
    lib = ctypes.CDLL(host.lib_path)
    np.random.seed(seed)
    got, st = _legacy_host(lib, 1, n, 0, [1.0])
    np.random.seed(seed)
    want = np.random.standard_normal(n)
    assert np.array_equal(got, want)
    assert _same_state(st, np.random.get_state())
    # continuing from a state with a cached normal and pos in the middle of a block
    np.random.set_state(st)
    got2, st2 = _legacy_host(lib, 1, 2 * n + 1, 0, [0.3])
    want2 = np.random.normal(scale=0.3, size=2 * n + 1)
    assert np.array_equal(got2, want2) and _same_state(st2, np.random.get_state())


def test_restated_glibc_log_equals_libm_bit_for_bit(host):
    """`rcl::log_glibc_fma` (legacy_rng_core.h; constants: glibc_log_data.h) is what the device evaluates for the legacy
    stream's normals: glibc's table-driven log restated with the fused / unfused pairing of the `__log_fma` build.  Against
    the C library's own log() on 10^7 arguments - uniform in (0, 1), both sides of 1 (the separate near-1 polynomial), tiny:
    not one bit of difference.  (NumPy's legacy gauss calls exactly that log(): with division and square root correctly
    rounded on both sides the device's normals are NumPy's - tests/test_gpu_*: array_equal.)  Skipped where the host's libm
    is another one (then `rc_legacy_log_is_host_exact()` is 0 and the Python layer draws on the host)."""
    lib = ctypes.CDLL(host.lib_path)
    lib.rc_host_log_mismatches.restype = ctypes.c_longlong
    product = importlib.import_module("code-robchar_amd._lib").load()
    if not product.rc_legacy_log_is_host_exact():
        pytest.skip("this host's libm is not the glibc the device routine restates")
    first = ctypes.c_double(0.0)
    bad = lib.rc_host_log_mismatches(ctypes.c_longlong(10_000_000), ctypes.c_ulonglong(20220714), ctypes.byref(first))
    assert bad == 0, (bad, first.value)


def test_legacy_stream_periods_match_mcsim_consumption(host):
    """period / skip / scales = the reference's consumption (mcsim.py:425 + noise_model.py:137-146): per sigma level one
    burned `rng(scale=sigma)` draw, then C*K*3N draws scaled by sigma."""
    lib = ctypes.CDLL(host.lib_path)
    noises = [0.0, 0.03, 0.1]
    C, K, N = 3, 5, 4
    np.random.seed(77)
    np.random.normal()                                   # leave a cached normal behind: odd starting position
    got, st = _legacy_host(lib, len(noises), 1 + C * K * N * 3, 1, noises)
    np.random.seed(77)
    np.random.normal()
    want = []
    for s in noises:
        np.random.normal(scale=s)
        want.append(np.random.normal(scale=s, size=(C, K, N, 3)))
    assert np.array_equal(got.reshape(len(noises), C, K, N, 3), np.array(want))
    assert _same_state(st, np.random.get_state())


# ----------------------------------------------------------------------------------------------------------------
# ring topology: Householder tridiagonalisation + shared QL (hermitian_core.h), on the host
# ----------------------------------------------------------------------------------------------------------------
def _ring_host(lib, ctrl, draws, N, a, b, h0d=None, corner=1.0, force_general=False):
    C, K = draws.shape[:2]
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
    draws = np.ascontiguousarray(draws, dtype=np.float64)
    h0d = np.zeros(N) if h0d is None else np.ascontiguousarray(h0d, dtype=np.float64)
    h0o = np.ones(N - 1)
    res = np.empty((C, K))
    lib.rc_host_ring_fidelity.argtypes = [ctypes.c_int, P, P, P, ctypes.c_double, P, ctypes.c_longlong, ctypes.c_longlong,
                                          ctypes.c_int, ctypes.c_int, P, ctypes.c_int]
    rc = lib.rc_host_ring_fidelity(N, ctrl.ctypes.data_as(P), h0d.ctypes.data_as(P), h0o.ctypes.data_as(P), corner,
                                   draws.ctypes.data_as(P), C, K, a, b, res.ctypes.data_as(P), int(force_general))
    assert rc == 0
    return res


@pytest.mark.parametrize("N", [3, 4, 5, 7, 8, 10, 11, 12, 13, 16])
def test_ring_core_vs_oracle(host, N):
    lib = ctypes.CDLL(host.lib_path)
    rng = np.random.default_rng(100 + N)
    C, K = 5, 40
    ctrl = np.empty((C, N + 1))
    ctrl[:, :N] = rng.uniform(-10, 10, (C, N))
    ctrl[:, N] = rng.uniform(2, 30, C)
    ctrl[1, :N] = rng.uniform(-1e-6, 1e-6, N)             # near-degenerate diagonal (translation-invariant ring: degenerate pairs)
    ctrl[2, N] = -ctrl[2, N]
    for sigma in (0.0, 0.05, 0.3):
        draws = sigma * rng.standard_normal((C, K, N, 3))
        for (a, b) in ((0, N - 1), (0, N // 2), (1, 1), (N - 1, 2 % N)):
            want = orc.fidelity_eigh(ctrl, draws, N, a, b, ring=True)
            got = _ring_host(lib, ctrl, draws, N, a, b)
            assert np.abs(got - want).max() < 1e-11, (N, sigma, a, b)
            gen = _ring_host(lib, ctrl, draws, N, a, b, force_general=True)
            assert np.abs(gen - want).max() < 1e-11
    # XXZ diagonal on a ring, and corner = 0 degenerates to the chain (tau = 0 reflectors)
    h0 = orc.xxz_delta(N, ring=True)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    assert np.abs(_ring_host(lib, ctrl, draws, N, 0, N - 1, h0d=h0) - orc.fidelity_eigh(ctrl, draws, N, 0, N - 1, h0_diag=h0, ring=True)).max() < 1e-11
    chain = orc.fidelity_eigh(ctrl, draws, N, 0, N - 1)
    assert np.abs(_ring_host(lib, ctrl, draws, N, 0, N - 1, corner=0.0) - chain).max() < 1e-11
    zero = np.zeros((C, 3, N, 3))
    assert np.abs(_ring_host(lib, ctrl, zero, N, 0, N - 1, corner=0.0) - orc.fidelity_eigh(ctrl, zero, N, 0, N - 1)).max() < 1e-11


@pytest.mark.parametrize("N", [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16])
def test_ring_mixed_route_vs_oracle(host, N):
    """The mixed-precision ring route (hermitian_core.h: ring_fidelity_mixed - sparse fp32 Householder + fp32 QL for the
    starting values, fp64 Halley on chi_ring = P_full - c^2 P_inner - Phi, two-path cofactor weights) against the oracle's
    dense eigh, every (in, out) pair incl. in == out and out < in (the ring breaks time reversal), XXZ offsets, sigma from
    0 to 0.3; and the fraction of samples that needed the all-fp64 fallback."""
    lib = ctypes.CDLL(host.lib_path)
    lib.rc_host_ring_mixed_fallbacks.restype = ctypes.c_longlong
    rng = np.random.default_rng(300 + N)
    C, K = 6, 48
    ctrl = np.empty((C, N + 1))
    ctrl[:, :N] = rng.uniform(-10, 10, (C, N))
    ctrl[:, N] = rng.uniform(2, 30, C)
    ctrl[2, N] = -ctrl[2, N]
    before = lib.rc_host_ring_mixed_fallbacks()
    total = 0
    for sigma in (0.0, 0.05, 0.3):
        draws = sigma * rng.standard_normal((C, K, N, 3))
        for a in range(N):
            for b in range(N):
                if N > 5 and (a * N + b) % 3 and (a, b) not in ((0, N - 1), (N - 1, 0)):
                    continue                                    # every third pair at the larger sizes
                want = orc.fidelity_eigh(ctrl, draws, N, a, b, ring=True)
                got = _ring_host(lib, ctrl, draws, N, a, b, force_general=2)
                assert np.abs(got - want).max() < 1e-11, (N, sigma, a, b, np.abs(got - want).max())
                total += C * K
    h0 = orc.xxz_delta(N, ring=True)
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    for (a, b) in ((0, N - 1), (N // 2, 0)):
        want = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0, ring=True)
        assert np.abs(_ring_host(lib, ctrl, draws, N, a, b, h0d=h0, force_general=2) - want).max() < 1e-11
    # random spectra: the mixed route carries most samples itself (a pair closer than 1e-3 of the scale - a few percent of the
    # samples at these sizes - is handed to the eigenvector route)
    assert lib.rc_host_ring_mixed_fallbacks() - before <= 0.12 * total
    # translation-invariant ring (degenerate pairs k <-> -k, split only by the noise): the fallback must take over, same answer
    ctrl[:, :N] = rng.uniform(-1e-6, 1e-6, (C, N))
    draws = 1e-7 * rng.standard_normal((C, K, N, 3))
    before = lib.rc_host_ring_mixed_fallbacks()
    for (a, b) in ((0, N - 1), (1, N // 2)):
        want = orc.fidelity_eigh(ctrl, draws, N, a, b, ring=True)
        assert np.abs(_ring_host(lib, ctrl, draws, N, a, b, force_general=2) - want).max() < 1e-11
    if N >= 3:
        assert lib.rc_host_ring_mixed_fallbacks() > before


@pytest.mark.parametrize("N", [5, 7, 8, 10, 12, 13])
def test_mixed_precision_eigenvalues_close_pairs(host, N):
    """The mixed-precision eigenvalue path (fp32 QL + fp64 Halley step, N = 3..13; tridiag_core.h) on the spectra it
    finds hardest: two resonant sites far apart, everything between them detuned by 2..8 J - an eigenvalue pair
    1e-5 .. 1e-2 apart in every sample, down to far closer than fp32 resolves.  Stepping path, critical-point guard and the
    all-fp64 fallback all run on the host exactly as in the kernel (one sample = one "wave").  Absolute AND relative
    agreement with the oracle (the transfer through a detuned chain is weak: 1e-9 .. 0.6), in both weight modes, and the
    general routine is (all but) never needed."""
    rng = np.random.default_rng(4242 + N)
    C, K = 12, 400
    ctrl = np.empty((C, N + 1))
    ctrl[:, :N] = rng.uniform(2.0, 8.0, (C, 1)) * (-1.0) ** np.arange(N) + rng.uniform(-0.5, 0.5, (C, N))
    ctrl[:, 0] = 1.0 + rng.uniform(-0.2, 0.2, C)
    ctrl[:, N - 1] = ctrl[:, 0] + rng.uniform(-1e-4, 1e-4, C)
    ctrl[:, N] = rng.uniform(20, 30, C)
    draws = 3e-4 * rng.standard_normal((C, K, N, 3))
    draws[0, :8] = 0.0                                               # noiseless rows: the gap is the controller's own
    before = host.general_calls()
    for (a, b) in ((0, N - 1), (0, N // 2), (N - 1, 1)):
        got = host(ctrl, draws, N, a, b)
        ref = orc.fidelity_eigh(ctrl, draws, N, a, b)
        assert np.abs(got - ref).max() < 1e-11, (N, a, b)
        assert (np.abs(got - ref) <= 1e-12 + 1e-7 * ref).all(), (N, a, b)
    assert host.general_calls() - before <= 3 * C * K // 1000


def test_mixed_precision_eigenvalues_property_sweep(host):
    """Property sweep of the kernel arithmetic on the host (hypothesis): any chain 3 <= N <= 13 (the mixed-precision range),
    any in / out pair, biases from tiny to +-50, noise from 0 to 1, times up to 60, optional XXZ offsets, an occasional
    exactly cancelled coupling - always within 1e-10 of the dense-eigh oracle (measured: <= 1e-12)."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(st.integers(3, 13), st.integers(0, 2**31 - 1), st.sampled_from([0.0, 1e-6, 0.05, 0.3, 1.0]),
           st.sampled_from([1e-3, 1.0, 10.0, 50.0]), st.booleans(), st.booleans())
    def run(N, seed, sigma, amp, xxz, cut):
        rng = np.random.default_rng(seed)
        C, K = 2, 24
        ctrl = np.empty((C, N + 1))
        ctrl[:, :N] = rng.uniform(-amp, amp, (C, N))
        ctrl[:, N] = rng.uniform(0.5, 60.0, C)
        draws = sigma * rng.standard_normal((C, K, N, 3))
        if cut:
            i = int(rng.integers(1, N))
            draws[0, ::3, i, 1] = -1.0                       # re = 1 + g1 = 0 and im = 0: the chain is cut there
            draws[0, ::3, i, 2] = 0.0
        a, b = int(rng.integers(0, N)), int(rng.integers(0, N))
        h0 = orc.xxz_delta(N) if xxz else None
        got = host(ctrl, draws, N, a, b, h0)
        want = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0)
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() < 1e-10, (N, a, b, sigma, amp, xxz, cut, float(np.abs(got - want).max()))

    run()


# ---------------------------------------------------------------------------------------------------------------
# The acceptance rule of the mixed-precision eigenvalue path under SYNTHETIC starting values (tridiag_core.h:
# mixed_refine).  Whatever the fp32 QL hands over, an ACCEPTED set must be the spectrum; a start that cannot be refined
# must come back as "escalate" (the kernel then runs the all-fp64 QL for the tile).
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def refine(host):
    lib = ctypes.CDLL(host.lib_path)
    PF = ctypes.POINTER(ctypes.c_float)

    def call(d, e, start, ok32=True):
        N = len(d)
        d = np.ascontiguousarray(d, dtype=np.float64)
        e = np.ascontiguousarray(e, dtype=np.float64)
        st = np.ascontiguousarray(start, dtype=np.float32)
        lam = np.empty(N)
        extra = ctypes.c_int(0)
        acc = lib.rc_host_mixed_refine(N, d.ctypes.data_as(P), e.ctypes.data_as(P), st.ctypes.data_as(PF), int(ok32),
                                       lam.ctypes.data_as(P), ctypes.byref(extra))
        assert acc in (0, 1)
        return bool(acc), lam, bool(extra.value)
    return call


def _spectrum(d, e):
    from scipy.linalg import eigvalsh_tridiagonal
    return eigvalsh_tridiagonal(np.asarray(d, float), np.asarray(e, float))


def _close_pair_matrix(N, delta, rng):
    """Tridiagonal (Jacobi) matrix with a PRESCRIBED spectrum - one eigenvalue pair `delta` apart, centred on zero (fp32
    starting values resolve fractions of the gap there), the other levels O(1..10) away - from the Lanczos process on
    diag(lam) with a random start vector (full reorthogonalisation, long double)."""
    while True:
        lam = np.sort(rng.uniform(-10, 10, N))
        if np.diff(lam).min() > 0.8:
            break
    j = int(rng.integers(0, N - 1))
    lam[j + 1:] -= (lam[j + 1] - lam[j]) - delta       # close the gap above level j to delta
    lam -= 0.5 * (lam[j] + lam[j + 1])
    L = lam.astype(np.longdouble)
    q = rng.uniform(0.3, 1.0, N).astype(np.longdouble)
    q /= np.sqrt((q * q).sum())
    Q, alpha, beta = [q], [], []
    for i in range(N):
        w = L * Q[i]
        a = (w * Q[i]).sum()
        alpha.append(a)
        w = w - a * Q[i] - (beta[-1] * Q[i - 1] if i else 0)
        for _ in range(2):
            for qq in Q:
                w = w - (w * qq).sum() * qq
        if i < N - 1:
            bnorm = np.sqrt((w * w).sum())
            beta.append(bnorm)
            Q.append(w / bnorm)
    d, e = np.array(alpha, dtype=np.float64), np.array(beta, dtype=np.float64)
    true = _spectrum(d, e)
    assert abs((true[j + 1] - true[j]) / delta - 1) < 1e-3 and int(np.argmin(np.diff(true))) == j
    return d, e, true, j


def _critical_point(lam, j):
    """Root of chi' between lam[j] and lam[j+1] (bisection on sum_k prod_{m != k} (mu - lam_m) / prod scale)."""
    def dchi(mu):
        return sum(np.prod([(mu - lam[m]) for m in range(len(lam)) if m != k]) for k in range(len(lam)))
    a, b = lam[j], lam[j + 1]
    fa = dchi(a)
    for _ in range(200):
        c = 0.5 * (a + b)
        fc = dchi(c)
        if (fc > 0) == (fa > 0):
            a, fa = c, fc
        else:
            b = c
    return 0.5 * (a + b)


def _check(acc, lam, true, tag):
    if acc:
        err = np.abs(np.sort(lam) - true).max()
        assert err <= 1e-13 * max(1.0, np.abs(true).max()), (tag, err)
    return acc


@pytest.mark.parametrize("N", [5, 7, 10, 13])
def test_acceptance_rule_synthetic_starts(refine, N):
    rng = np.random.default_rng(100 + N)
    n_acc = n_rej = n_crit_rej = 0
    for delta in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
        for rep in range(6):
            d, e, true, j = _close_pair_matrix(N, delta, rng)
            gap = true[j + 1] - true[j]
            c = _critical_point(true, j)
            exact = true.astype(np.float32)            # "exact" starts: the spectrum rounded to fp32
            # healthy starts are accepted (or escalated for the tightest pairs) and correct
            acc, lam, _ = refine(d, e, exact)
            n_acc += _check(acc, lam, true, ("exact", N, delta))
            # (i) ONE start at the critical point +- eps, every other start an exact root: the step there is tiny although
            # the start is no root - the rule must not accept it as one
            for eps in (0.0, 1e-12, 1e-10, 1e-8, 1e-7, 1e-6, -1e-12, -1e-9, -1e-6):
                for which in (j, j + 1):
                    st = exact.copy()
                    st[which] = np.float32(c + eps * gap / 1e-2)
                    acc, lam, extra = refine(d, e, st)
                    n_crit_rej += not acc
                    n_rej += not _check(acc, lam, true, ("critical", N, delta, eps, which))
            # (ii) two starts in the SAME basin (both next to the upper eigenvalue of the pair)
            for off in (0.05, 0.2, 0.45):
                st = exact.copy()
                st[j] = np.float32(true[j + 1] - off * gap)
                acc, lam, _ = refine(d, e, st)
                n_rej += not _check(acc, lam, true, ("same basin", N, delta, off))
            # (iii) a start a full gap (and more) off, on either side; a start on top of a FOREIGN eigenvalue
            for off in (-1.0, 1.0, -3.0, 3.0):
                st = exact.copy()
                st[j] = np.float32(true[j] + off * gap)
                acc, lam, _ = refine(d, e, st)
                n_rej += not _check(acc, lam, true, ("gap off", N, delta, off))
            st = exact.copy()
            st[j] = exact[(j + 3) % N]
            acc, lam, _ = refine(d, e, st)
            n_rej += not _check(acc, lam, true, ("duplicate", N, delta))
            # (v) the LAST start takes no step of its own (it is what the trace leaves) but sits in every other start's
            # Aberth sum: a poor one must show up in the acceptance rule (as the distance the trace moves it)
            g_last = true[N - 1] - true[N - 2]
            for off in (1e-4, 1e-3, 1e-2, 0.1, 0.4, -0.3):
                st = exact.copy()
                st[N - 1] = np.float32(true[N - 1] + off * g_last)
                acc, lam, _ = refine(d, e, st)
                n_rej += not _check(acc, lam, true, ("last start off", N, delta, off))
            # (iv) arbitrary starts with ok32 = False (fp32 QL hit its sweep cap)
            acc, lam, _ = refine(d, e, rng.uniform(-12, 12, N).astype(np.float32), ok32=False)
            n_rej += not _check(acc, lam, true, ("garbage", N, delta))
    assert n_acc > 0 and n_crit_rej > 0            # both outcomes occur: the test exercises acceptance AND rejection


@pytest.mark.parametrize("N", [4, 7, 10, 13])
def test_acceptance_rule_random_starts(refine, N):
    """Randomised: starts = spectrum + noise of every size from fp32 rounding to half a gap, at matrix scales 1e-2 ... 1e3
    (the fp32 uncertainty of a gap must scale with the matrix: |d| ~ 1e3 has fp32 ulps of 6e-5)."""
    rng = np.random.default_rng(200 + N)
    acc_n = 0
    for it in range(400):
        scale = 10.0 ** rng.uniform(-2, 3)
        d = rng.uniform(-1, 1, N) * scale
        e = rng.uniform(0.05, 1.0, N - 1) * scale * 10.0 ** rng.uniform(-3, 0)
        if it % 3 == 0:                                # near-degenerate neighbours
            k = int(rng.integers(0, N - 1))
            d[k + 1] = d[k] + scale * 10.0 ** rng.uniform(-7, -2)
            e[k] = scale * 10.0 ** rng.uniform(-7, -2)
        true = _spectrum(d, e)
        noise = 10.0 ** rng.uniform(-8, -1, N) * scale * rng.choice([-1, 1], N)
        if it % 4 == 1:
            noise[:] = 0.0
        st = (true + noise).astype(np.float32)
        if it % 2:
            st = st[rng.permutation(N)]                # the fp32 QL delivers its eigenvalues in no particular order
        acc, lam, _ = refine(d, e, st)
        if acc:
            acc_n += 1
            err = np.abs(np.sort(lam) - true).max()
            assert err <= 2e-13 * max(1.0, np.abs(true).max()), (N, it, scale, err)
    assert acc_n > 50


@pytest.mark.parametrize("ndir,n,cached", [(19, 300, False), (19, 300, True), (1, 40, True), (4, 5000, False),
                                           (32, 2000, True), (33, 20000, False), (28, 20000, True)])
def test_directional_parse_equals_numpy(host, ndir, n, cached):
    """The device-side parse of `directional_perturbation`'s RNG consumption (noise_model.py:183-189: per sample
    `np.random.randint(0, ndir)` then `normal(size=2)`), executed on the host through the same header functions the
    kernels use - per-position sample lengths, sequential walk over them, per-sample emit - against NumPy ITSELF call by
    call: indices, normals and the generator state afterwards bit-identical (libm's log here; the device's differs by ulps),
    entered with and without a cached normal."""
    lib = ctypes.CDLL(host.lib_path)
    np.random.seed(4000 + ndir + n)
    np.random.normal(size=3 if cached else 2)
    st0 = np.random.get_state()
    assert bool(st0[3]) == cached
    want_idx, want_ab = np.empty(n, dtype=np.int64), np.empty((n, 2))
    for i in range(n):
        want_idx[i] = np.random.randint(low=0, high=ndir)
        want_ab[i] = np.random.normal(scale=0.05, size=2)
    st1 = np.random.get_state()
    key = np.ascontiguousarray(st0[1], dtype=np.uint32).copy()
    pos, hg, g = ctypes.c_int(int(st0[2])), ctypes.c_int(int(st0[3])), ctypes.c_double(float(st0[4]))
    idx, ab = np.empty(n, dtype=np.int32), np.empty((n, 2))
    words = ((int(n * 7.2) + 8000) // 624 + 2) * 624
    rc = lib.rc_host_directional_parse(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), ctypes.byref(hg), ctypes.byref(g),
                                       ctypes.c_longlong(n), ndir, ctypes.c_double(0.05), ctypes.c_longlong(words),
                                       idx.ctypes.data_as(ctypes.c_void_p), ab.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    assert np.array_equal(idx, want_idx) and np.array_equal(ab, want_ab)
    assert np.array_equal(key, st1[1]) and pos.value == st1[2] and hg.value == st1[3] and g.value == st1[4]


def test_fuzz_regression_cut_chain_near_degenerate(host):
    """Inputs the round-3 fuzz campaign failed on (tests/golden/fuzz_r3_adjugate_cut_chain.npz: the GPU's inputs, dumped by
    scripts/fuzz_parity.py): cut chains with mirror-symmetric halves whose levels pair up across the cut at 1e-12 .. 5e-9 of
    the scale, in / out on ONE side.  With the general-adjugate weights taken down to gaps of 1e-12 of the scale (a threshold
    that is right for the end-to-end weights only) the errors were 2e-10 .. 2.7e-8: the numerators phi psi are recurrences
    evaluated beside their own roots.  The per-mode threshold (tridiag_core.h: kDegenerateGapAdjugate) sends them to the
    eigenvector route.  Last entry (`weak_bond_window_seed2150`, second campaign): mirror halves joined by a bond of 1.4e-7,
    pairs 5.8e-6 .. 4e-5 apart - just ABOVE that threshold, where the adjugate weights are used and cost 2.7e-11: within the
    bound, kept so that a change of the threshold or of the weights shows up here."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "fuzz_r3_adjugate_cut_chain.npz"))
    for name in z["names"]:
        name = str(name)
        N, a, b = (int(v) for v in z[name + "_meta"])
        ctrl, draws = z[name + "_ctrl"], z[name + "_draws"]
        h0 = z[name + "_h0"] if z[name + "_h0"].size else None
        want = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0)
        for vec in ("adj", True):
            got = host(ctrl, draws, N, a, b, h0d=h0, vec=vec)
            assert np.abs(got - want).max() < 1e-10, (name, vec, np.abs(got - want).max())


@pytest.mark.parametrize("N", [2, 3, 5, 7, 10, 12])
def test_complex_symmetric_ql_route_vs_oracle(host, N):
    """csym_core.h: chains with a COMPLEX diagonal (what `directional_perturbation`'s diagonal directions produce,
    noise_model.py:196-199) through the complex symmetric QL route - gauge to real couplings, implicit QL with
    complex-orthogonal rotations, two rows of Q, exp(-i T lam) with complex lam - against the oracle's per-sample
    scipy.linalg.expm of the non-Hermitian matrix: single-site imaginary entries of the size of the noise (the directional
    model), imaginary parts on every site, |Im| up to 0.5, every class of (in, out)."""
    lib = ctypes.CDLL(host.lib_path)
    rng = np.random.default_rng(700 + N)
    C, K = 3, 30
    ctrl = np.empty((C, N + 1))
    ctrl[:, :N] = rng.uniform(-10, 10, (C, N))
    ctrl[:, N] = rng.uniform(2, 30, C)
    total_fb = 0
    for kind in ("single", "all", "strong"):
        draws = 0.05 * rng.standard_normal((C, K, N, 3))
        imag = np.zeros((C, K, N))
        if kind == "single":
            site = rng.integers(0, N, (C, K))
            np.put_along_axis(imag, site[..., None], 0.05 * rng.standard_normal((C, K, 1)), axis=2)
            draws[..., 1:] = 0.0                                 # a diagonal direction perturbs nothing else
        elif kind == "all":
            imag = 0.05 * rng.standard_normal((C, K, N))
        else:
            imag = rng.uniform(-0.5, 0.5, (C, K, N))
        for (a, b) in ((0, N - 1), (N - 1, 0), (0, N // 2), (N // 2, N // 2)):
            want = orc.fidelity_expm_loop(ctrl, draws, N, a, b, diag_imag=imag)
            got = np.empty((C, K))
            fb = ctypes.c_longlong(0)
            h0d, h0o = np.zeros(N), np.ones(max(N - 1, 1))
            rc = lib.rc_host_csym_fidelity(N, ctrl.ctypes.data_as(P), h0d.ctypes.data_as(P), h0o.ctypes.data_as(P),
                                           np.ascontiguousarray(draws).ctypes.data_as(P), np.ascontiguousarray(imag).ctypes.data_as(P),
                                           ctypes.c_longlong(C), ctypes.c_longlong(K), a, b, got.ctypes.data_as(P), ctypes.byref(fb))
            assert rc == 0
            total_fb += fb.value
            okm = ~np.isnan(got)
            # relative to the size of the result: a growing mode (Im > 0) makes |U|^2 exceed 1 by orders of magnitude
            tol = 1e-10 * np.maximum(1.0, want)
            assert (np.abs(got - want)[okm] <= tol[okm]).all(), (N, kind, a, b, np.abs(got - want)[okm].max())
    assert total_fb <= 3                                           # the route carries (all but) everything itself


@pytest.mark.parametrize("N", [3, 4, 7, 10, 13, 16])
def test_sum_rule_guard_catches_a_perturbed_weight(host, N):
    """The a-posteriori guard of the eigenvalue-only weight modes (tridiag_core.h: chain_sum_rules_ok): sum_k w_k (lam_k - c)^m
    must equal ((H - c)^m)[out, in] for m = 0, 1, 2.  Every class of (in, out) - same site, neighbours, two apart, far apart,
    end to end - on random matrices: the untouched weights pass, ONE weight moved by 1e-11 (an error of ~2e-11 in a fidelity)
    is caught whatever its index.  An EIGENVALUE error is invisible to the rules below |out - in| by construction
    (Lagrange: they hold for any set of distinct eigenvalues) - pinned here so that nobody mistakes the guard for more."""
    lib = ctypes.CDLL(host.lib_path)
    lib.rc_host_guard_check.argtypes = [ctypes.c_int, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    rng = np.random.default_rng(40 + N)
    pairs = {(0, N - 1), (N - 1, 0), (0, 0), (N // 2, N // 2), (N - 1, N - 1), (0, 1), (N // 2, N // 2 - 1), (N - 1, N - 2),
             (0, 2), (N - 1, N - 3)} | ({(1, N - 2), (0, N // 2)} if N >= 5 else set())
    for trial in range(6):
        amp = (1.0, 10.0, 100.0)[trial % 3]
        d = rng.uniform(-amp, amp, N)
        e = np.abs(1 + 0.05 * rng.standard_normal(N - 1) + 0.05j * rng.standard_normal(N - 1))
        dp, ep = d.ctypes.data_as(P), e.ctypes.data_as(P)
        for (a, b) in pairs:
            assert lib.rc_host_guard_check(N, dp, ep, a, b, 0, 0.0, 0.0) == 1, (N, a, b, "clean weights rejected")
            for k in range(N):
                assert lib.rc_host_guard_check(N, dp, ep, a, b, k, 1e-11, 0.0) == 0, (N, a, b, k, "perturbed weight accepted")
                assert lib.rc_host_guard_check(N, dp, ep, a, b, k, -3e-12, 0.0) == 0, (N, a, b, k)
            # blind spot, by construction: an eigenvalue error (far pairs: every rule is an identity in the eigenvalues)
            if abs(a - b) >= 3:
                assert lib.rc_host_guard_check(N, dp, ep, a, b, N // 2, 0.0, 1e-9 * amp) == 1


@pytest.mark.parametrize("N", [2, 3, 5, 8])
def test_complex_symmetric_route_near_breakdown(host, N):
    """Near-defective matrices (round 4, conditioning guard of csym_core.h): two neighbouring sites with imaginary parts
    +-(J - delta) and equal real parts form a block [[i y, J], [J, -i y]] whose eigenvalues +-sqrt(J^2 - y^2) coalesce at
    y = J (an exceptional point: no complex-orthogonal eigenbasis).  Whatever the route returns unmarked must be accurate;
    close enough to the exceptional point it must mark the sample (NaN here, the expm pass on the device) instead of
    returning a finite but inaccurate number."""
    lib = ctypes.CDLL(host.lib_path)
    rng = np.random.default_rng(900 + N)
    C, K = 2, 8
    marked = {}
    for delta in (1e-2, 1e-5, 1e-8, 1e-11, 0.0):
        ctrl = np.empty((C, N + 1))
        ctrl[:, :N] = rng.uniform(-3, 3, (C, N))
        ctrl[:, 1] = ctrl[:, 0]                                  # equal real parts on sites 0 and 1
        ctrl[:, N] = rng.uniform(2, 30, C)
        draws = np.zeros((C, K, N, 3))
        draws[..., 2:, 1:] = 0.02 * rng.standard_normal((C, K, max(N - 2, 0), 2))
        if N > 2:
            draws[..., 2, 1], draws[..., 2, 2] = -1.0, 0.0       # bond 1-2 cancelled exactly: the block stands alone
        imag = np.zeros((C, K, N))
        imag[..., 0], imag[..., 1] = 1.0 - delta, -(1.0 - delta)
        for (a, b) in ((0, 1), (0, 0), (1, 0)):
            want = orc.fidelity_expm_loop(ctrl, draws, N, a, b, diag_imag=imag)
            got = np.empty((C, K))
            fb = ctypes.c_longlong(0)
            h0d, h0o = np.zeros(N), np.ones(max(N - 1, 1))
            assert lib.rc_host_csym_fidelity(N, ctrl.ctypes.data_as(P), h0d.ctypes.data_as(P), h0o.ctypes.data_as(P),
                                             draws.ctypes.data_as(P), imag.ctypes.data_as(P), ctypes.c_longlong(C),
                                             ctypes.c_longlong(K), a, b, got.ctypes.data_as(P), ctypes.byref(fb)) == 0
            okm = ~np.isnan(got)
            tol = 1e-10 * np.maximum(1.0, want)
            assert (np.abs(got - want)[okm] <= tol[okm]).all(), (N, delta, a, b, np.abs(got - want)[okm].max())
            marked[delta] = marked.get(delta, 0) + int((~okm).sum())
    print(f"N = {N}: samples handed to the expm pass per distance from the exceptional point: {marked}")
    assert marked[1e-2] == 0                                      # well-conditioned: the route carries it
    assert marked[0.0] == 3 * C * K                               # at the exceptional point itself: every sample marked


# ---- lock-step emulation of a wave (tests/host/host_wave.cpp): what wave-uniform decisions do to the OTHER lanes of a tile ----
@pytest.fixture(scope="module")
def wave(tmp_path_factory):
    """`tile(ctrl_row, draws (nk, N, 3), N, a, b, mode, h0d, old=False)` -> (fid, repaired, extra): ONE 64-lane tile of the chain
    kernel with every lane a host thread and every wave-level vote a barrier (mode: 0 rows / 1 general adjugate / 2 end-to-end,
    None = what RC_KERNEL_AUTO picks).  `old=True`: the build in which every lane of a tile that takes the tile-wide fp64 QL uses
    the QL's eigenvalues (-DRC_KEEP_SETTLED=0: the behaviour before round 4)."""
    d = tmp_path_factory.mktemp("hostwave")
    libs = {}
    for name, flags in (("new", []), ("old", ["-DRC_KEEP_SETTLED=0"]), ("newton", ["-DRC_STEP2_NEWTON_ALL=1"]),
                        ("allfirst", ["-DRC_STEP_ALL_FIRST=1"])):
        out = d / f"librc_hostwave_{name}.so"
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-pthread"] + flags
                       + ["-o", str(out), os.path.join(ROOT, "tests", "host", "host_wave.cpp")], check=True)
        libs[name] = ctypes.CDLL(str(out))
    PI = ctypes.POINTER(ctypes.c_int)

    def tile(ctrl, draws, N, a, b, mode=None, h0d=None, old=False, variant=None):
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64).reshape(-1)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        nk = draws.shape[0]
        h0 = np.zeros(32)
        if h0d is not None:
            h0[:N] = h0d
        h0o = np.ones(32)
        if mode is None:
            mode = 2 if {a, b} == {0, N - 1} else 1
        fid, rep, ex = np.empty(nk), np.zeros(nk, dtype=np.int32), np.zeros(nk, dtype=np.int32)
        rc = libs[variant or ("old" if old else "new")].rc_host_wave_chain_tile(
            N, ctrl.ctypes.data_as(P), h0.ctypes.data_as(P), h0o.ctypes.data_as(P), draws.ctypes.data_as(P), nk, a, b, mode,
            fid.ctypes.data_as(P), rep.ctypes.data_as(PI), ex.ctypes.data_as(PI))
        assert rc == 0
        return fid, rep, ex

    def ring_tile(ctrl, draws, N, a, b, route=0, h0d=None):
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64).reshape(-1)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        nk = draws.shape[0]
        h0 = np.zeros(32)
        if h0d is not None:
            h0[:N] = h0d
        h0o = np.ones(32)
        fid, rep, ex = np.empty(nk), np.zeros(nk, dtype=np.int32), np.zeros(nk, dtype=np.int32)
        rc = libs["new"].rc_host_wave_ring_tile(N, ctrl.ctypes.data_as(P), h0.ctypes.data_as(P), h0o.ctypes.data_as(P),
                                                draws.ctypes.data_as(P), nk, a, b, route, fid.ctypes.data_as(P),
                                                rep.ctypes.data_as(PI), ex.ctypes.data_as(PI))
        assert rc == 0
        return fid, rep, ex
    tile.ring = ring_tile
    return tile


def test_wave_emulation_settled_lanes_keep_their_eigenvalues(wave):
    """The round-4 fuzz fixture ON THE CPU: 15 tiles (|bias| ~ 100, |T| ~ 70 .. 95) in which a neighbour sends the tile through
    the tile-wide fp64 QL.  With one sample per "wave" (every other host test) nothing happens to the healthy samples; in
    lock-step the pre-fix build reproduces the GPU's error (1.4e-11 on the GPU, 1.5e-11 here) and the fixed build the GPU's
    result after the fix (2.1e-12) - the emulator sees what wave-uniform decisions do to the other lanes."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "fuzz_r4_tile_fallback.npz"))
    keys = sorted(k[:-5] for k in d.files if k.endswith("_ctrl"))
    assert len(keys) == 15
    worst = {False: 0.0, True: 0.0}
    for key in keys:
        N, a, b = (int(v) for v in d[key + "_meta"])
        h0 = d[key + "_h0"] if d[key + "_h0"].size else None
        want = d[key + "_want"][0]
        for old in (False, True):
            fid, rep, ex = wave(d[key + "_ctrl"][0], d[key + "_draws"][0], N, a, b, h0d=h0, old=old)
            assert (ex > 0).all()                                  # the tile left the one-step path - as every lane saw it
            worst[old] = max(worst[old], np.abs(fid - want).max())
    assert worst[False] < 5e-12, worst
    assert worst[True] > 8e-12, worst                              # (the harness must SEE the pre-fix behaviour)


@pytest.mark.parametrize("N", [3, 5, 7, 10, 13])
def test_wave_emulation_random_tiles_vs_oracle(wave, N):
    """Full 64-lane tiles in lock-step against the oracle, every weight mode: benchmark-style samples mixed with lanes that
    force the wave-uniform escalations on their neighbours - a resonant pair (stepping path), a pair closer than the mixed
    path separates (tile-wide fp64 QL), two cut-off sites at the same energy (exactly degenerate: per-lane repair) - and a ragged tile (nk < 64)."""
    rng = np.random.default_rng(900 + N)
    seen = {"stepping": 0, "repair": 0}
    for trial in range(4):
        nk = 64 if trial < 3 else 37
        x = np.concatenate([rng.uniform(-10, 10, N), [rng.uniform(2, 30)]])
        g = 0.05 * rng.standard_normal((nk, N, 3))
        if trial >= 1 and N >= 4:
            # lanes 5 and 41: two sites in resonance (eigenvalue pair ~1e-6 / ~1e-9 apart after the noise is removed there)
            for lane, gap in ((5, 1e-6), (41 % nk, 1e-9)):
                g[lane, :, 0] = 0.0
                g[lane, 1, 0] = x[0] - x[1] + gap                  # d_1 = d_0 + gap
                g[lane, 1, 1:] = (-1.0 + 1e-3, 0.0)                # weak bond 0-1: the pair splits by ~2e-3
        if trial >= 2 and N >= 4:
            # lane 17: both end sites cut off and at the same energy - an EXACTLY degenerate pair of decoupled levels: the
            # eigenvalue-only weights cannot take it (0 / 0), the lane goes to the per-lane eigenvector repair
            x[N - 1] = x[0]
            g[17, :, 0] = 0.0
            g[17, 1, 1:] = (-1.0, 0.0)
            g[17, N - 1, 1:] = (-1.0, 0.0)
        for (a, b) in ((0, N - 1), (0, N // 2), (N // 2, N // 2)):
            want = orc.fidelity_eigh(x[None, :], g[None], N, a, b)[0]
            for mode in (None, 0):
                fid, rep, ex = wave(x, g, N, a, b, mode=mode)
                assert np.abs(fid - want).max() < 1e-11, (N, trial, a, b, mode, np.abs(fid - want).max())
                seen["stepping"] += int(mode is None and (ex > 0).any())
                seen["repair"] += int(mode is None and (rep > 0).any())
    if N >= 4:                                                     # the adversarial lanes did force the escalations
        assert seen["stepping"] > 0 and seen["repair"] > 0, seen


@pytest.mark.parametrize("cid", [3, 5])
def test_wave_emulation_second_step_newton_for_all_eigenvalues(wave, cid):
    """-DRC_STEP2_NEWTON_ALL=1 (round 5 experiment, tridiag_core.h: newton_polish_all): a tile that fails the one-step
    acceptance takes ONE Newton step for every eigenvalue (independent chains) before any bookkeeping and is accepted on
    (N-1) s^2 <= 1e-14 gap.  In lock-step on the DELOCALISED controller sets (close pairs by construction: mirror-symmetric
    shipped L-BFGS controllers, N = 7; constructed N = 10 XXZ) against the oracle: same 1e-11 as the shipped route, and the
    Newton step does settle most of the tiles that left the one-step path (otherwise the variant has no point)."""
    from conftest import highfid_workload
    N, a, b, ctrl, h0 = highfid_workload(cid, 24)
    rng = np.random.default_rng(1700 + cid)
    stats = {"flagged": 0, "newton": 0, "tiles": 0}
    worst = {"new": 0.0, "newton": 0.0}
    for c in range(24):
        g = 0.05 * rng.standard_normal((64, N, 3))
        want = orc.fidelity_eigh(ctrl[c:c + 1], g[None], N, a, b, h0_diag=h0)[0]
        for variant in ("new", "newton", "allfirst"):
            fid, rep, ex = wave(ctrl[c], g, N, a, b, h0d=h0, variant=variant)
            worst[variant] = max(worst.get(variant, 0.0), float(np.abs(fid - want).max()))
            if variant == "newton":
                stats["tiles"] += 1
                stats["flagged"] += int((ex > 0).any())
                stats["newton"] += int((ex == 100).all())
    # (`allfirst` = -DRC_STEP_ALL_FIRST=1: the first stepping iteration for every eigenvalue, bookkeeping only behind it)
    assert worst["new"] < 1e-11 and worst["newton"] < 1e-11 and worst["allfirst"] < 1e-11, worst
    assert stats["flagged"] >= 3 and stats["newton"] >= 0.5 * stats["flagged"], stats


@pytest.mark.parametrize("N", [3, 4, 6, 7, 9, 10])
def test_wave_emulation_ring_tiles_vs_oracle(wave, N):
    """Ring topology in lock-step: the mixed-precision route for the tile, the all-fp64 route for the lanes it lists (together,
    as one repair wave), against the oracle - benchmark-style tiles, a tile of a translation-invariant ring (flat diagonal:
    every sample has pairs k <-> -k split only by the noise and is listed), mixed tiles, a ragged tile; both routes."""
    rng = np.random.default_rng(950 + N)
    listed = 0
    for trial in range(4):
        nk = 64 if trial < 3 else 29
        x = np.concatenate([rng.uniform(-10, 10, N), [rng.uniform(2, 30)]])
        g = 0.05 * rng.standard_normal((nk, N, 3))
        if trial == 1:
            x[:N] = rng.uniform(-1e-6, 1e-6, N) + 3.0              # translation-invariant ring
            g *= 1e-5
        if trial == 2:
            g[::7] *= 1e-7                                         # some lanes next to the controller's own spectrum ...
            x[1] = x[0] + 1e-5                                     # ... which has a resonant pair of sites
        for (a, b) in ((0, N - 1), (0, N // 2), (N // 2, N // 2), (N - 1, 1)):
            want = orc.fidelity_eigh(x[None, :], g[None], N, a, b, ring=True)[0]
            for route in (0, 1):
                fid, rep, ex = wave.ring(x, g, N, a, b, route=route)
                assert np.abs(fid - want).max() < 2e-11, (N, trial, a, b, route, np.abs(fid - want).max())
                listed += int(route == 0 and (rep > 0).any())
    assert listed > 0                                              # the repair wave did run
