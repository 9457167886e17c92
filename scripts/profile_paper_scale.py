#!/usr/bin/env python3
"""cProfile of a cold paper-scale `MCDataSim.get_metrics_dict()` (4 algorithms x 11 levels x 1000 controllers x 100 draws, N = 5,
device Philox draws, metrics only) - where do the 22 ms go on the host?  Development aid (needs a GPU)."""
import cProfile, importlib, io, json, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
tmp = tempfile.mkdtemp(prefix="robchar_prof_"); os.chdir(tmp); os.makedirs("experiments")
def make(exp, N, out_spin, algos, C):
    rng = np.random.default_rng(5); le = {}
    for a in algos:
        x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
        le[a] = {("%d" % N if a == "lbfgs" else "0.05"): {"controller": x.tolist()}}
    os.makedirs(f"experiments/{exp}", exist_ok=True)
    json.dump(le, open(f"experiments/{exp}/ppo_spin_{N}_0-{out_spin}_c_{C}", "w"))
def run(exp, prof=None, **kw):
    make(exp, 5, 2, ["ppo", "snob", "nmplus", "lbfgs"], 1000)
    np.random.seed(1)
    sim = mcmod.MCDataSim(experiment_name=exp, Nspin=5, inspin=0, outspin=2, noises=np.linspace(0, 0.1, 11), bootreps=100,
                          training_noise=0.05, numcontrollers=1000, verbose=False, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if prof: prof.enable()
    sim.get_metrics_dict()
    torch.cuda.synchronize()
    if prof: prof.disable()
    return time.perf_counter() - t0
mode = dict(rng_mode=sys.argv[1] if len(sys.argv) > 1 else "philox", seed=7, cache_format=sys.argv[2] if len(sys.argv) > 2 else "none")
print("warm", run("w0", **mode)); print("plain", [round(run(f"p{i}", **mode), 4) for i in range(3)])
pr = cProfile.Profile(); print("profiled", run("q", pr, **mode))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
