"""The C-ABI library loads and exports every function include/robchar_hip.h declares (no compute calls)."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "robchar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    libmod = importlib.import_module("code-robchar_amd._lib")
    lib = libmod.load()
    names = declared_functions()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in robchar_hip.h but not exported"
    assert sorted(libmod.EXPORTS) == names
    # the loaded binary is the one this header describes (a stale library fails here, not somewhere in a kernel call)
    header = open(os.path.join(ROOT, "include", "robchar_hip.h")).read()
    assert lib.rc_version() == int(re.search(r"#define\s+RC_ABI_VERSION\s+(\d+)", header).group(1))
    assert isinstance(lib.rc_device_count(), int)


def test_argument_validation_without_gpu():
    libmod = importlib.import_module("code-robchar_amd._lib")
    lib = libmod.load()
    # argument checks happen before any HIP call
    rc = lib.rc_mc_fidelity_f64(0, 99, 0, 0, None, None, 0, None, None, 1, 1, None)
    assert rc == -1 and b"N must be" in lib.rc_last_error()
    rc = lib.rc_mc_fidelity_f64(0, 5, 0, 7, None, None, 0, None, None, 1, 1, None)
    assert rc == -1 and b"out of range" in lib.rc_last_error()
    assert lib.rc_mc_fidelity_f64(0, 5, 0, 2, None, None, 0, None, None, 0, 10, None) == 0     # empty batch
    assert lib.rc_reduce_f64(0, None, 0, 5, None, 0, 0.0, None, None, None, None, None) == 0
    assert lib.rc_reduce_f64(0, None, 3, 5, None, 9, 0.0, None, None, None, None, None) == -1
    assert lib.rc_set_fidelity_kernel(17) == -1 and lib.rc_set_fidelity_kernel(3) == 0 and lib.rc_set_fidelity_kernel(0) == 0
    assert lib.rc_reserve_ring(0, None, -1) == -1 and b"samples" in lib.rc_last_error()        # (ABI 5)


def test_round2_entries_validate_arguments_without_gpu():
    """The multi-device, legacy-stream and JSON entries reject bad arguments before any HIP call (CPU box: no GPU)."""
    import numpy as np
    libmod = importlib.import_module("code-robchar_amd._lib")
    lib = libmod.load()
    z = ctypes.c_void_p(0)
    one = np.ones(8)
    p = ctypes.c_void_p(one.ctypes.data)
    # rc_mc_fidelity_sharded_f64(ndev, devices, kernel, N, in, out, h0d, h0o, ring, ctrl, draws, C, K, fid)
    assert lib.rc_mc_fidelity_sharded_f64(0, z, 0, 5, 0, 2, z, z, 0, p, p, 1, 1, p) == -1 and b"ndev" in lib.rc_last_error()
    assert lib.rc_mc_fidelity_sharded_f64(1, z, 0, 99, 0, 2, z, z, 0, p, p, 1, 1, p) == -1 and b"N must be" in lib.rc_last_error()
    assert lib.rc_mc_fidelity_sharded_f64(1, z, 0, 5, 0, 2, z, z, 0, p, z, 1, 1, p) == -1            # NULL draws
    assert lib.rc_mc_fidelity_sharded_f64(1, z, 0, 5, 0, 2, z, z, 0, z, z, 0, 10, z) == 0            # empty batch
    # rc_mc_metrics_sharded_f64(..., ctrl, draws, seed, offset, sigma, C, K, thr, nq, eps, rim1, std, min, q, fid)
    args = [1, z, 0, 5, 0, 2, z, z, 0, p, z, 1, 0, 0.05, 1, 1]
    assert lib.rc_mc_metrics_sharded_f64(*args, z, 9, 0.0, p, z, z, z, z) == -1 and b"nq" in lib.rc_last_error()
    assert lib.rc_mc_metrics_sharded_f64(*args, z, 0, 0.0, z, z, z, z, z) == -1 and b"no output" in lib.rc_last_error()
    # rc_draws_legacy_f64(device, stream, state, n_periods, period, skip, scales, out)
    st = libmod.Mt19937State()
    assert lib.rc_draws_legacy_f64(0, z, z, 1, 4, 0, p, p) == -1 and b"state" in lib.rc_last_error()
    st.pos = 700
    assert lib.rc_draws_legacy_f64(0, z, ctypes.byref(st), 1, 4, 0, p, p) == -1 and b"pos" in lib.rc_last_error()
    st.pos = 624
    assert lib.rc_draws_legacy_f64(0, z, ctypes.byref(st), 1, 4, 5, p, p) == -1                       # skip > period
    assert lib.rc_draws_legacy_f64(0, z, ctypes.byref(st), 0, 4, 0, p, p) == 0                        # nothing to draw
    # rc_json_*: pure host code, works here
    shape = (ctypes.c_longlong * 2)(2, 3)
    cap = lib.rc_json_bound_f64(2, shape)
    buf = ctypes.create_string_buffer(int(cap))
    data = np.arange(6, dtype=np.float64) / 4
    n = lib.rc_json_encode_f64(ctypes.c_void_p(data.ctypes.data), 2, shape, buf, cap, 1)
    assert buf.raw[:n] == b"[[0.0, 0.25, 0.5], [0.75, 1.0, 1.25]]"
    assert lib.rc_json_encode_f64(ctypes.c_void_p(data.ctypes.data), 2, shape, buf, 3, 1) == -1      # capacity too small
    assert lib.rc_json_bound_f64(9, shape) == -1 and lib.rc_json_write_f64(-1, z, 2, shape, 1) == -1


def test_build_flags_and_the_fused_route_rule():
    """ABI 6: the in-tree library is the product build (no experiment switch), and `rc_philox_fused_pays` is the one copy of
    the fused-route rule - the Python layer asks it, and ROBCHAR_PHILOX_FUSED=0 is read per call on both sides."""
    libmod = importlib.import_module("code-robchar_amd._lib")
    be = importlib.import_module("code-robchar_amd.backend")
    lib = libmod.load()
    assert lib.rc_build_flags() == 0 and libmod.build_flags() == 0
    for n in range(2, 17):
        for a, b in ((0, n - 1), (n - 1, 0), (0, n // 2)):
            want = n <= 13 or (n == 14 and {a, b} == {0, n - 1})
            assert bool(lib.rc_philox_fused_pays(n, a, b)) == want == be.philox_fused_pays(n, a, b), (n, a, b)
    assert lib.rc_philox_fused_pays(17, 0, 16) == 0 and lib.rc_philox_fused_pays(1, 0, 0) == 0
    os.environ["ROBCHAR_PHILOX_FUSED"] = "0"
    try:
        assert lib.rc_philox_fused_pays(7, 0, 6) == 0 and not be.philox_fused_pays(7, 0, 6)
    finally:
        del os.environ["ROBCHAR_PHILOX_FUSED"]
    assert lib.rc_philox_fused_pays(7, 0, 6) == 1


def test_rccl_entries_validate_arguments_without_gpu():
    """rc_comm_init / rc_mc_metrics_gathered_f64 (ABI 6) reject bad arguments before any HIP or RCCL call."""
    libmod = importlib.import_module("code-robchar_amd._lib")
    lib = libmod.load()
    h = ctypes.c_void_p()
    assert lib.rc_comm_init(0, None, ctypes.byref(h)) == -1 and b"ndev" in lib.rc_last_error()
    assert lib.rc_comm_init(1, None, None) == -1 and b"NULL" in lib.rc_last_error()
    assert lib.rc_comm_size(None) == 0 and lib.rc_comm_destroy(None) == 0
    z = ctypes.c_void_p(0)
    assert lib.rc_mc_metrics_gathered_f64(None, 0, 5, 0, 2, z, z, 0, z, z, 0, 0, 0.05, 1, 1, z, 0, 0.0, z, z, z, z) == -1
    assert b"communicator" in lib.rc_last_error()


def test_loader_refuses_an_experiment_build(tmp_path):
    """A library that reports a timing-experiment switch (results knowingly wrong) is refused by `_lib.load()` unless
    ROBCHAR_ALLOW_EXPERIMENT_LIB=1; a library without `rc_build_flags` (older ABI) is refused as stale.  Checked in a child
    interpreter with stub libraries - the guard runs before anything else of the ABI is bound."""
    import subprocess
    import sys
    for body, name in (("int rc_build_flags(void) { return 1; }", "exp"), ("int rc_version(void) { return 5; }", "old")):
        src = tmp_path / f"{name}.c"
        src.write_text(body + "\n")
        subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(tmp_path / f"lib{name}.so"), str(src)], check=True)
    code = ("import importlib, sys; sys.path.insert(0, %r); m = importlib.import_module('code-robchar_amd._lib')\n"
            "try:\n    m.load(); print('LOADED')\nexcept m.RobCharHipError as e:\n    print('REFUSED', e)\n" % ROOT)
    def run(lib, allow=None):
        env = dict(os.environ, ROBCHAR_HIP_LIB=str(tmp_path / lib))
        env.pop("ROBCHAR_ALLOW_EXPERIMENT_LIB", None)
        if allow:
            env["ROBCHAR_ALLOW_EXPERIMENT_LIB"] = allow
        return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout
    out = run("libexp.so")
    assert out.startswith("REFUSED") and "EXPERIMENT_NO_STEPPING" in out
    assert run("libold.so").startswith("REFUSED") and "predates ABI 6" in run("libold.so")
    # allowed explicitly: the guard lets it through (the stub then fails at the first missing symbol, not at the guard)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROBCHAR_HIP_LIB=str(tmp_path / "libexp.so"),
                                                                ROBCHAR_ALLOW_EXPERIMENT_LIB="1"), capture_output=True, text=True)
    assert "REFUSED" not in out.stdout and "rc_version" in out.stderr


def test_plain_c99_client_compiles_and_links(tmp_path):
    """`include/robchar_hip.h` is a C header: `tests/host/c_client.c` (C99, -pedantic -Wall -Wextra -Werror) compiles against it
    and links against the library with the C compiler alone - no C++ runtime, no Python in the binding.  (It is RUN on the GPU
    box: tests/test_gpu_chain.py.)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "code-robchar_amd", "csrc")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                        "-o", str(tmp_path / "c_client"), os.path.join(root, "tests", "host", "c_client.c"),
                        "-L", libdir, "-lrobchar_hip", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # the same header as C++11 (a C++ integrator)
    r = subprocess.run(["g++", "-std=c++11", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", os.path.join(root, "include"),
                        os.path.join(root, "tests", "host", "c_client.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_hip_host_client_compiles_and_links(tmp_path):
    """`tests/host/hip_client.cpp` (a HIP host program on the enqueue-only entries, its own streams and device buffers) builds
    with hipcc against the header and the library; it is RUN by tests/test_gpu_chain.py on the GPU box."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "code-robchar_amd", "csrc")
    r = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-o", str(tmp_path / "hip_client"),
                        os.path.join(root, "tests", "host", "hip_client.cpp"), "-L", libdir, "-lrobchar_hip", f"-Wl,-rpath,{libdir}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
