#!/usr/bin/env python3
"""Where the directional pipeline's wall time goes (N = 7, 100 x 10 000): the RNG continuation on the device
(rc_directional_draws_legacy_dev: raw words, per-position lengths, D2H + host walk, emit), the fidelity pass
(rc_mc_fidelity_directional_f64_async) and the D2H of the result, each bracketed by synchronisations."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
noise = importlib.import_module("code-robchar_amd.noise")
rng = np.random.default_rng(0)
N, C, K = 7, 100, 10000
x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
nm = noise.directional_perturbation(Nspin=N, inspin=0, outspin=6, noise=0.05)
dev = torch.device("cuda", torch.cuda.current_device())
ctrl = torch.from_numpy(x).to(dev)
acc = {}
def stage(name, t0):
    torch.cuda.synchronize()
    acc.setdefault(name, []).append(time.perf_counter() - t0)
for rep in range(10):
    np.random.seed(1)
    torch.cuda.synchronize()
    t = time.perf_counter(); idx, ab = be.directional_draws_device(C * K, 3 * N, 0.05); stage("draws (device RNG continuation)", t)
    t = time.perf_counter(); fid = be.mc_fidelity_directional(ctrl, idx, ab, N, 0, 6, K); stage("fidelity pass", t)
    t = time.perf_counter(); host = nm._to_host(fid); stage("D2H of the (C, K) result", t)
    np.random.seed(1)
    t = time.perf_counter(); f = nm.fidelity_batch(x, K, draws="device"); stage("fidelity_batch, end to end", t)
for k, v in acc.items():
    print(f"{k:36s} median {np.median(v[2:]) * 1e3:7.3f} ms   min {min(v[2:]) * 1e3:7.3f} ms")
