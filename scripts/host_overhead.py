#!/usr/bin/env python3
"""Host-side enqueue cost of the two hot-path calls (tiny problem, so GPU time is negligible)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
N, C, K = 7, 1, 64
ctrl = torch.rand((C, N + 1), dtype=torch.float64, device="cuda")
draws = torch.zeros((C, K, N, 3), dtype=torch.float64, device="cuda")
out = torch.empty((C, K), dtype=torch.float64, device="cuda")
red = be.reduce_metrics(out)
views = {k: red[k] for k in ("rim1", "std", "min", "q")}
for name, fn in (("mc_fidelity", lambda: be.mc_fidelity(ctrl, draws, N, 0, 6, out=out)),
                 ("reduce_metrics(out=)", lambda: be.reduce_metrics(out, dkw_eps=0.01, out=views)),
                 ("reduce_metrics(alloc)", lambda: be.reduce_metrics(out, dkw_eps=0.01))):
    for _ in range(100): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2000): fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"{name}: {dt/2000*1e6:.1f} us per call (host enqueue)")
