#!/usr/bin/env python3
"""Kernel time of the headline shape as a function of time since the GPU left idle: groups of 16 back-to-back launches
bracketed by events, printed as a series (development aid: shows the clock / power transient a short benchmark run sits in)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
be = importlib.import_module("code-robchar_amd.backend")
from bench import make_controllers, legacy_draws
N, C, K = 7, 100, 10000
ctrl = torch.from_numpy(make_controllers(3, C, N, 0)).cuda()
draws = [torch.from_numpy(legacy_draws(12345 + 7919 * t, C, K, N)).cuda() for t in range(3)]
out = torch.empty((C, K), dtype=torch.float64, device="cuda")
for idle in (0.5, 0.0):
    torch.cuda.synchronize(); time.sleep(idle)
    G, NG = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 400
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(NG + 1)]
    ev[0].record()
    for g in range(NG):
        for j in range(G):
            be.mc_fidelity(ctrl, draws[(g * G + j) % 3], N, 0, 6, out=out)
        ev[g + 1].record()
    torch.cuda.synchronize()
    us = np.array([ev[g].elapsed_time(ev[g + 1]) for g in range(NG)]) / G * 1e3
    t = np.cumsum(us * G) / 1e3
    print(f"after {idle} s idle: us per launch by group of 16 (time since start in ms):")
    for g in list(range(0, 24)) + list(range(24, NG, max(1, NG // 24))):
        print(f"  group {g:4d} t={t[g]:8.2f} ms  {us[g]:6.2f} us")

# a bare synchronize in the middle of a long run: does the ~30 us gap restart the transient?
torch.cuda.synchronize()
G, NG = 16, 150
for gap in ("sync", "sync+barrier-like 300us"):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * NG + 2)]
    for g in range(NG):
        ev[g].record()
        for j in range(G):
            be.mc_fidelity(ctrl, draws[(g * G + j) % 3], N, 0, 6, out=out)
    ev[NG].record()
    torch.cuda.synchronize()
    if gap != "sync":
        time.sleep(300e-6)
    for g in range(NG):
        ev[NG + 1 + g].record()
        for j in range(G):
            be.mc_fidelity(ctrl, draws[(g * G + j) % 3], N, 0, 6, out=out)
    ev[2 * NG + 1].record()
    torch.cuda.synchronize()
    before = np.array([ev[g].elapsed_time(ev[g + 1]) for g in range(NG)]) / G * 1e3
    after = np.array([ev[NG + 1 + g].elapsed_time(ev[NG + 2 + g]) for g in range(NG)]) / G * 1e3
    print(f"gap = {gap}: last 5 groups before {np.round(before[-5:], 1)}; after the gap, groups 0..11: {np.round(after[:12], 1)}; 12..23: {np.round(after[12:24], 1)}; 100..105: {np.round(after[100:106], 1)}")
