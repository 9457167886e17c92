#!/usr/bin/env python3
"""Where the benchmark step's time goes: fidelity kernel back-to-back vs + reductions (same stream / side stream)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
N, C, K = 7, 100, 10000
rng = np.random.default_rng(0)
ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
ct = torch.from_numpy(ctrl).cuda()
draws = torch.from_numpy(0.05 * rng.standard_normal((C, K, N, 3))).cuda()
fid = [torch.empty((C, K), dtype=torch.float64, device="cuda") for _ in range(3)]
red = {"rim1": torch.empty((3, C), dtype=torch.float64, device="cuda"), "std": torch.empty((3, C), dtype=torch.float64, device="cuda"),
       "min": torch.empty((3, C), dtype=torch.float64, device="cuda"), "q": torch.empty((3, 2, C), dtype=torch.float64, device="cuda")}
side = torch.cuda.Stream(priority=-1)
main = torch.cuda.current_stream()
ev = [torch.cuda.Event() for _ in range(3)]; ev2 = [torch.cuda.Event() for _ in range(3)]

def run(mode, steps=400):
    for it in range(steps + 40):
        if it == 40:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        b = it % 3
        if mode == "side" and it >= 3:
            main.wait_event(ev2[b])
        be.mc_fidelity(ct, draws, N, 0, N - 1, out=fid[b])
        if mode == "same":
            be.reduce_metrics(fid[b], dkw_eps=0.0136, out=red)
        elif mode == "side":
            ev[b].record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev[b])
                be.reduce_metrics(fid[b], dkw_eps=0.0136, out=red)
                ev2[b].record(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6

def host_only(steps=400):
    t0 = time.perf_counter()
    for it in range(steps):
        be.mc_fidelity(ct[:1], draws[:1, :64], N, 0, N - 1, out=fid[0][:1, :64])
    t = (time.perf_counter() - t0) / steps * 1e6
    torch.cuda.synchronize()
    return t

for mode in ("fid", "same", "side", "fid", "side"):
    print(f"{mode:5s}: {run(mode):7.1f} us/step")
print(f"host enqueue cost of mc_fidelity alone (tiny launch): {host_only():.1f} us")
