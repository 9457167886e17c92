#!/usr/bin/env python3
"""Kernel time of the counter-based draw generator (rc_draws_philox_f64_async) for config 4's 2.1e9 normals, HIP events."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
be = importlib.import_module("code-robchar_amd.backend")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000 * 100000 * 21
out = torch.empty((n,), dtype=torch.float64, device="cuda")
for _ in range(2):
    be.philox_normal((n,), seed=7, scale=0.05, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    be.philox_normal((n,), seed=7, scale=0.05, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"philox_normal_kernel: {n:.3e} normals in {ms:.3f} ms = {n / ms / 1e6:.1f} G normals/s, {8 * n / ms / 1e6:.0f} GB/s written (lib {os.environ.get('ROBCHAR_HIP_LIB', 'in-tree')})")
