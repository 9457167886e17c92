#!/usr/bin/env python3
"""End-to-end `MCDataSim` run at the paper's scale on synthetic controllers: 4 algorithms x 11 sigma_sim levels x
1000 controllers x 100 draws (what one `get_metrics_dict()` call of the reference's figure scripts triggers, e.g.
generate_fig3.py:272-274), including the legacy-stream draws on the host and the `.mc` / `.mcm` JSON caches.
Prints where the wall time goes.  (The reference's own cost for this call: 4.4e6 evaluations x ~70 us = ~5 min
on one core)"""
import importlib, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
be = importlib.import_module("code-robchar_amd.backend")

N, a, b, C, K = 5, 0, 2, 1000, 100
rng = np.random.default_rng(0)
def ctrls():
    x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
    return x.tolist()
le = {alg: {"0.05": {"controller": ctrls()}} for alg in ("nmplus", "snob", "ppo")}
le["lbfgs"] = {str(N): {"controller": ctrls()}}
with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    os.makedirs("experiments/demo")
    json.dump(le, open(f"experiments/demo/ppo_spin_{N}_{a}-{b}_c_{C}.le", "w"))
    import torch                                                              # first import on a fresh box: ~10 s
    be.mc_fidelity(np.zeros((1, N + 1)), np.zeros((1, 1, N, 3)), N, a, b)     # load the library / warm the GPU
    np.random.seed(1)
    t0 = time.perf_counter()
    sim = mcmod.MCDataSim(experiment_name="demo", Nspin=N, inspin=a, outspin=b, bootreps=K, training_noise=0.05,
                          numcontrollers=C, filemarker=".le", verbose=False)
    t1 = time.perf_counter()
    metrics = sim.get_metrics_dict()
    t2 = time.perf_counter()
    again = sim.get_metrics_dict()
    t3 = time.perf_counter()
    evals = 4 * 11 * C * K
    sizes = {f: os.path.getsize(os.path.join("experiments/demo", f)) for f in os.listdir("experiments/demo") if ".mc" in f}
    print(f"controllers loaded in {t1 - t0:.2f}s; cold get_metrics_dict: {t2 - t1:.2f}s for {evals:.2e} evaluations "
          f"({evals / (t2 - t1):.3g} evals/s end-to-end incl. host RNG + JSON); warm (cache hit): {t3 - t2:.2f}s")
    print("cache files:", {k: f"{v/1e6:.1f} MB" for k, v in sizes.items()})
