"""pytest configuration: registers the ``gpu`` marker and exposes the golden fixtures.

CPU tests (``-m "not gpu"``): oracle vs golden vectors, host logic, C-ABI symbol export, gloo sharding.
GPU tests (``-m gpu``): parity of the HIP path (through the C-ABI) against the oracle and the fixtures.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """The native libraries are build artefacts (git-ignored): `__graft_entry__.build()` makes them, and so does this
    hook - `make` in both source directories at every session start.  The Makefiles list their sources, so this is a no-op
    on a fresh library and a rebuild when a source is newer (a suite never runs against a binary older than the tree).
    A failed build ends the session with the compiler's output.  ROBCHAR_TEST_NO_BUILD=1 skips the step (a GPU box that
    received the library with the snapshot and has nothing newer to build from)."""
    import shutil
    import subprocess
    if os.environ.get("ROBCHAR_TEST_NO_BUILD", "0") == "1" or shutil.which("make") is None:
        return
    for sub in (os.path.join("code-robchar_amd", "csrc"), "oracle"):
        d = os.path.join(ROOT, sub)
        if not os.path.exists(os.path.join(d, "Makefile")):
            continue
        # a tree copied to a GPU box arrives with the library but without the object files it was linked from: `make` would
        # compile everything again.  The library carries the hash of the sources it was made from; equal hashes = nothing to do.
        stamp, lib = os.path.join(d, "librobchar_hip.so.srchash"), os.path.join(d, "librobchar_hip.so")
        if os.path.exists(stamp) and os.path.exists(lib):
            h = subprocess.run(["make", "-s", "-C", d, "srchash"], capture_output=True, text=True)
            if h.returncode == 0 and h.stdout.strip() and h.stdout.strip() == open(stamp).read().strip():
                continue
        r = subprocess.run(["make", "-j4", "-C", d], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.exit(f"building the native library in {sub} failed:\n{(r.stdout + r.stderr)[-3000:]}", returncode=3)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def kernel_cases():
    z = load_npz("kernel_cases.npz")
    cases = []
    for name in z["names"]:
        name = str(name)
        n, mode, a, b = name.split("_")
        cases.append(dict(name=name, N=int(n[1:]), mode=mode, inspin=int(a), outspin=int(b),
                          ctrl=z[name + "_ctrl"], draws=z[name + "_draws"], fid=z[name + "_fid"]))
    return cases


@pytest.fixture(scope="session")
def shipped_sigma0():
    z = load_npz("shipped_sigma0.npz")
    out = []
    for name in z["names"]:
        name = str(name)
        parts = name.split("_")
        out.append(dict(name=name, N=int(parts[0][1:]), inspin=int(parts[1]), outspin=int(parts[2]),
                        ctrl=z[name + "_ctrl"], fid=z[name + "_fid"], navail=z[name + "_navail"]))
    return out


@pytest.fixture(scope="session")
def lbfgs_n7():
    z = load_npz("lbfgs_n7.npz")
    return {k: z[k] for k in z.files}


def highfid_workload(cid, C=None):
    """The DELOCALISED controller set of BASELINE config `cid` (tests/golden/highfid.npz + lbfgs_n7.npz; make_golden.py
    `highfid`): (N, in, out, controllers tiled to C rows, h0_diag or None).  SURVEY 8(d)'s uniform random biases are
    Anderson-localised - median fidelity 1e-7 at N = 7 -, so an absolute 1e-10 bound says little on them; these sets have
    mean fidelities of 0.44 - 0.68 at sigma = 0.05."""
    z = load_npz("highfid.npz")
    if cid == 2:
        N, a, b, rows, h0 = 5, 0, 4, z["c2_ctrl"], None
    elif cid == 3:
        N, a, b, rows, h0 = 7, 0, 6, load_npz("lbfgs_n7.npz")["ctrl_0-6"], None
    elif cid == 4:
        N, a, b, rows, h0 = 7, 0, 3, load_npz("lbfgs_n7.npz")["ctrl_0-3"], None
    elif cid == 5:
        N, a, b, rows, h0 = 10, 0, 9, z["c5_ctrl"], z["c5_h0_diag"]
    else:
        raise ValueError(cid)
    C = rows.shape[0] if C is None else C
    return N, a, b, np.ascontiguousarray(rows[np.arange(C) % rows.shape[0]]), h0


@pytest.fixture(scope="session")
def highfid():
    z = load_npz("highfid.npz")
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def be():
    """The C-ABI wrappers (`code-robchar_amd.backend`) on a box with a GPU - every GPU test goes through them."""
    import importlib
    mod = importlib.import_module("code-robchar_amd.backend")
    lib = importlib.import_module("code-robchar_amd._lib")
    assert lib.require_gpu() >= 1
    return mod


@pytest.fixture
def workdir(tmp_path, monkeypatch):
    """A scratch directory with the reference's `experiments/` tree as the working directory."""
    monkeypatch.chdir(tmp_path)
    os.mkdir("experiments")
    return tmp_path
