import importlib, sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
noise = importlib.import_module("code-robchar_amd.noise")
rng = np.random.default_rng(0)
N, C, K = 7, 100, 10000
x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
nm = noise.directional_perturbation(Nspin=N, inspin=0, outspin=6, noise=0.05)
np.random.seed(1); nm.fidelity_batch(x[:2], 10)
for rep in range(2):
    np.random.seed(1)
    t = time.perf_counter(); f = nm.fidelity_batch(x, K); dt = time.perf_counter() - t
    print(f"directional N=7 100 x 10000: {dt*1e3:.1f} ms -> {C*K/dt:.3e} evals/s")
