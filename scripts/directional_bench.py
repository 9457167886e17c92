"""End-to-end throughput of `directional_perturbation.fidelity_batch` (noise_model.py:150-201) at N = 7, 100 x 10 000:
draws="device" (round 3: RNG parse, layout and class split on the GPU) vs draws="host" (round 2: host emulation of the
stream, NumPy layout, H2D of the (C, K, N, 3) tensor)."""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
noise = importlib.import_module("code-robchar_amd.noise")
rng = np.random.default_rng(0)
N, C, K = 7, 100, 10000
x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
nm = noise.directional_perturbation(Nspin=N, inspin=0, outspin=6, noise=0.05)
for mode in ("device", "host"):
    np.random.seed(1); nm.fidelity_batch(x[:2], 10, draws=mode)
    for rep in range(8 if mode == "device" else 2):
        np.random.seed(1)
        t = time.perf_counter(); f = nm.fidelity_batch(x, K, draws=mode); dt = time.perf_counter() - t
        print(f"directional N=7 100 x 10000 draws={mode}: {dt*1e3:.1f} ms -> {C*K/dt:.3e} evals/s")
