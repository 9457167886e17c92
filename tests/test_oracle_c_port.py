"""The plain-C restatement of the reference-shaped path (oracle/expm_port.c: dense complex H, Pade
scaling-and-squaring expm) against the golden vectors and the NumPy oracle - CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import robchar_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.c_void_p


@pytest.fixture(scope="module")
def port():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "librc_oracle_port.so"))
    lib.rc_oracle_expm_fidelity.argtypes = [ctypes.c_int] * 3 + [P, P, ctypes.c_int, P, P, ctypes.c_longlong,
                                                                 ctypes.c_longlong, P, ctypes.c_int]

    def fid(ctrl, draws, N, a, b, h0d=None, ring=False, threads=2):
        ctrl = np.ascontiguousarray(ctrl, dtype=np.float64)
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        C, K = draws.shape[:2]
        out = np.empty((C, K))
        h = None if h0d is None else np.ascontiguousarray(h0d, dtype=np.float64)
        rc = lib.rc_oracle_expm_fidelity(N, a, b, h.ctypes.data if h is not None else None, None, int(ring),
                                         ctrl.ctypes.data, draws.ctypes.data, C, K, out.ctypes.data, threads)
        assert rc == 0
        return out
    return fid


def test_c_port_vs_golden(port, kernel_cases):
    worst = 0.0
    for case in kernel_cases:
        h0 = orc.xxz_delta(case["N"]) if case["mode"] == "xxz" else None
        for s in range(case["draws"].shape[0]):
            got = port(case["ctrl"], case["draws"][s], case["N"], case["inspin"], case["outspin"], h0,
                       ring=case["mode"] == "ring")
            worst = max(worst, np.abs(got - case["fid"][s]).max())
    assert worst < 1e-11, worst


def test_c_port_vs_numpy_oracle_and_nan(port):
    rng = np.random.default_rng(9)
    N, C, K = 7, 6, 40
    ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
    ctrl[3] = np.nan
    draws = 0.05 * rng.standard_normal((C, K, N, 3))
    got = port(ctrl, draws, N, 0, 6)
    want = orc.fidelity_eigh(ctrl, draws, N, 0, 6)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) < 1e-11
