#!/usr/bin/env python3
"""Timing of the device-side legacy normal stream (rc_draws_legacy_f64) at one paper-scale algorithm (1.65e7 normals)
and at the whole paper scale (6.6e7); development aid (run under rocprofv3 --kernel-trace --stats for the split)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
be = importlib.import_module("code-robchar_amd.backend")
noises = np.linspace(0, 0.1, 11)
for per, label in ((1000 * 100 * 5 * 3, "1 algorithm"), (4 * 1000 * 100 * 5 * 3, "4 algorithms")):
    for rep in range(3):
        np.random.seed(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = be.legacy_normal_periods(11, 1 + per, 1, noises)
        torch.cuda.synchronize()
        print(f"{label}: {11 * per:.2e} normals in {(time.perf_counter() - t0) * 1e3:.1f} ms")
