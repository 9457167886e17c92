"""The C-ABI library loads and exports every function include/robchar_hip.h declares (no compute calls)."""
import ctypes
import importlib
import os
import re

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
    assert lib.rc_version() == 2
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
