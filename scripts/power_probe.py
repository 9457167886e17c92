#!/usr/bin/env python3
"""Samples rocm-smi (sclk, socket power, cap) WHILE the fidelity kernel runs back-to-back for a few seconds.  Diagnostic."""
import importlib, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
N, C, K = int(os.environ.get("NN", "7")), 100, 10000
RING = os.environ.get("RING", "0") == "1"           # ring topology (lane-per-sample Householder + QL kernel)
OUT = int(os.environ.get("OUT", N - 1))
rng = np.random.default_rng(0)
ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
ct = torch.from_numpy(ctrl).cuda()
draws = torch.from_numpy(0.05 * rng.standard_normal((C, K, N, 3))).cuda()
fid = torch.empty((C, K), dtype=torch.float64, device="cuda")
samples = []
def sampler():
    for _ in range(6):
        time.sleep(0.6)
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power (W)" in l])
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
while th.is_alive():
    for _ in range(200):
        be.mc_fidelity(ct, draws, N, 0, OUT, out=fid, ring=RING)
    torch.cuda.synchronize(); n += 200
dt = time.perf_counter() - t0
print(f"N={N} out={OUT} ring={RING}: {n} launches, {dt / n * 1e6:.1f} us per launch")
for s in samples: print(s)
