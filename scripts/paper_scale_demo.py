#!/usr/bin/env python3
"""End-to-end `MCDataSim.get_metrics_dict()` at the paper's scale (4 algorithms x 11 levels x 1000 controllers x 100
draws, N = 5) on a cold cache, with a cProfile breakdown of where the host time goes.
usage: python scripts/paper_scale_demo.py [--rng philox|legacy] [--cache none|json|npy] [--profile]"""
import argparse, cProfile, importlib, json, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--rng", default="philox")
ap.add_argument("--cache", default="none")
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
N, C, K = 5, 1000, 100
rng = np.random.default_rng(5)
le = {}
for a in ("ppo", "snob", "nmplus", "lbfgs"):
    x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
    le[a] = {("%d" % N if a == "lbfgs" else "0.05"): {"controller": x.tolist()}}
os.chdir(tempfile.mkdtemp())
os.makedirs("experiments")
def run(exp):
    os.makedirs(f"experiments/{exp}")
    json.dump(le, open(f"experiments/{exp}/ppo_spin_{N}_0-2_c_{C}", "w"))
    np.random.seed(1)
    t0 = time.perf_counter()
    sim = mcmod.MCDataSim(experiment_name=exp, Nspin=N, inspin=0, outspin=2, bootreps=K, training_noise=0.05,
                          numcontrollers=C, verbose=False, rng_mode=args.rng, seed=7, cache_format=args.cache)
    t1 = time.perf_counter()
    met = sim.get_metrics_dict()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1
run("warm")
for i in range(3):
    a, b = run(f"e{i}")
    print(f"construct {a * 1e3:.1f} ms (controller file load), get_metrics_dict cold {b * 1e3:.1f} ms")
if args.profile:
    pr = cProfile.Profile()
    pr.enable(); run("prof"); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
