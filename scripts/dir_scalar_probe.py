#!/usr/bin/env python3
"""Where the first call of the non-Hermitian entry spends its time (one-off initialisation costs on a fresh process)."""
import importlib, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
be = importlib.import_module("code-robchar_amd.backend")
rng = np.random.default_rng(0)
N = 7
def T(label, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(f"{label}: {(time.perf_counter() - t0) * 1e3:.2f} ms"); return r
x = np.concatenate([rng.uniform(-10, 10, N), [11.0]])[None, :]
g = 0.05 * rng.standard_normal((1, 8, N, 3))
im = np.zeros((1, 8, N)); im[0, :, 2] = 0.01
T("chain kernel, first call (numpy in)", lambda: be.mc_fidelity(x, g, N, 0, 6))
T("chain kernel, second call", lambda: be.mc_fidelity(x, g, N, 0, 6))
T("ring kernel, first call", lambda: be.mc_fidelity(x, g, N, 0, 6, ring=True))
os.environ["RC_NH_EXPM_ONLY"] = "1"
T("nh entry, expm only, first call", lambda: be.mc_fidelity_nonhermitian(x, g, im, N, 0, 6))
T("nh entry, expm only, second call", lambda: be.mc_fidelity_nonhermitian(x, g, im, N, 0, 6))
os.environ.pop("RC_NH_EXPM_ONLY")
T("nh entry, csym route, first call", lambda: be.mc_fidelity_nonhermitian(x, g, im, N, 0, 6))
T("nh entry, csym route, second call", lambda: be.mc_fidelity_nonhermitian(x, g, im, N, 0, 6))
T("jacobi kernel, first call", lambda: be.mc_fidelity(x, g, N, 0, 6, ring=True, kernel="jacobi"))
T("reduce, first call", lambda: be.reduce_metrics(np.random.rand(4, 100)))
