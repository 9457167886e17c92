#!/usr/bin/env python3
"""Where BASELINE config 4 through the product API spends its wall time (`end_to_end.c4_level_api`: one sigma level, 1000
controllers x 100 000 draws, device Philox draws, metrics only; the kernels take 4.8 + 4.8 + 0.3 ms): cProfile of the cold
`MCDataSim.get_metrics_dict()` call + a stage split with synchronisations (allocation of the 16.8 GB draw tensor, generator,
fidelity, reduction).  Development aid (needs a GPU)."""
import cProfile, importlib, io, json, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
be = importlib.import_module("code-robchar_amd.backend")
tmp = tempfile.mkdtemp(prefix="robchar_prof_"); os.chdir(tmp); os.makedirs("experiments")
N, C, K = 7, 1000, 100000
def make(exp):
    rng = np.random.default_rng(5)
    x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
    os.makedirs(f"experiments/{exp}", exist_ok=True)
    json.dump({"ppo": {"0.05": {"controller": x.tolist()}}}, open(f"experiments/{exp}/ppo_spin_{N}_0-3_c_{C}", "w"))
    return x
def run(exp, prof=None):
    make(exp)
    np.random.seed(1)
    sim = mcmod.MCDataSim(experiment_name=exp, Nspin=N, inspin=0, outspin=3, noises=np.array([0.05]), bootreps=K,
                          training_noise=0.05, numcontrollers=C, verbose=False, rng_mode="philox", seed=11, cache_format="none")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if prof: prof.enable()
    sim.get_metrics_dict()
    torch.cuda.synchronize()
    if prof: prof.disable()
    return time.perf_counter() - t0
print("warm", run("w0")); print("plain", [round(run(f"p{i}"), 4) for i in range(4)])
pr = cProfile.Profile(); print("profiled", run("q", pr))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:7000])
# stage split, synchronised
x = torch.from_numpy(make("s")).cuda()
def t(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for rep in range(3):
    d, ta = t(lambda: torch.empty((C, K, N, 3), dtype=torch.float64, device="cuda"))
    _, tg = t(lambda: be.philox_normal(d.shape, 11, scale=0.05, out=d))
    f, tf = t(lambda: be.mc_fidelity(x, d, N, 0, 3))
    r, tr = t(lambda: be.reduce_packed(f, 0.0043))
    _, tc = t(lambda: r.cpu())
    print(f"alloc {ta:.3f} ms  philox {tg:.3f}  fidelity {tf:.3f}  reduce {tr:.3f}  d2h rows {tc:.3f}")
    del d, f, r
