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

G = 8
big = [torch.empty((G * C, K), dtype=torch.float64, device="cuda") for _ in range(2)]
redg = {"rim1": torch.empty((3, G * C), dtype=torch.float64, device="cuda"), "std": torch.empty((3, G * C), dtype=torch.float64, device="cuda"),
        "min": torch.empty((3, G * C), dtype=torch.float64, device="cuda"), "q": torch.empty((3, 2, G * C), dtype=torch.float64, device="cuda")}
gev = [torch.cuda.Event() for _ in range(2)]; gev2 = [torch.cuda.Event() for _ in range(2)]

def run_group(steps=400):
    """8 fidelity launches back-to-back into one (8C, K) block, ONE event, ONE reduction launch over the block."""
    for it in range(steps + 40):
        if it == 40:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        g, grp = it % G, (it // G) % 2
        if g == 0 and it >= 2 * G:
            main.wait_event(gev2[grp])
        be.mc_fidelity(ct, draws, N, 0, N - 1, out=big[grp][g * C:(g + 1) * C])
        if g == G - 1:
            gev[grp].record(main)
            with torch.cuda.stream(side):
                side.wait_event(gev[grp])
                be.reduce_metrics(big[grp], dkw_eps=0.0136, out=redg)
                gev2[grp].record(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6

def host_only(steps=400):
    t0 = time.perf_counter()
    for it in range(steps):
        be.mc_fidelity(ct[:1], draws[:1, :64], N, 0, N - 1, out=fid[0][:1, :64])
    t = (time.perf_counter() - t0) / steps * 1e6
    torch.cuda.synchronize()
    return t

for mode in ("fid", "same", "side", "group", "fid", "side", "group"):
    t = run_group() if mode == "group" else run(mode)
    print(f"{mode:5s}: {t:7.1f} us/step")
print(f"host enqueue cost of mc_fidelity alone (tiny launch): {host_only():.1f} us")
