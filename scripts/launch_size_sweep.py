#!/usr/bin/env python3
"""What a launch BOUNDARY costs the fidelity kernel (round 5): the same kernel on C = 25 ... 4000 controllers x K draws per launch,
back-to-back launches on one stream inside ONE pair of HIP events (no per-launch markers), time per launch fitted as a C + b:
b = what every launch pays whatever its size (dispatch gap, pipeline fill - the first round of waves all wait for their draws
at once -, drain), a = the steady-state cost per controller.

usage: python3 scripts/launch_size_sweep.py [--N 7] [--out end|mid] [--K 10000] [--xxz] [--total 4e8]
"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
be = importlib.import_module("code-robchar_amd.backend")

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=7)
ap.add_argument("--K", type=int, default=10000)
ap.add_argument("--out", default="end")
ap.add_argument("--xxz", action="store_true")
ap.add_argument("--sizes", default="25,50,100,200,400,1000,4000")
ap.add_argument("--total", type=float, default=4e8, help="evaluations per timed figure (launches = total / (C K), at least 20)")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
N, K = args.N, args.K
o = N - 1 if args.out == "end" else N // 2
sizes = [int(v) for v in args.sizes.split(",")]
Cmax = max(sizes)
rng = np.random.default_rng(20220714 + N)
ctrl = np.empty((Cmax, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (Cmax, N)); ctrl[:, N] = rng.uniform(2, 30, Cmax)
ct = torch.from_numpy(ctrl).cuda()
# device draws (Philox): one tensor of the largest size, every smaller launch reads a rotating window of it (so that a small
# launch's draws do not sit in the Infinity Cache from the launch before)
draws = be.philox_normal((Cmax, K, N, 3), seed=N, scale=0.05, as_torch=True)
out = torch.empty((Cmax, K), dtype=torch.float64, device="cuda")
h0 = None
if args.xxz:
    from oracle import robchar_oracle as orc
    h0 = orc.xxz_delta(N)
st = torch.cuda.current_stream()
res = {}
for C in sizes:
    nl = max(20, int(args.total / (C * K)))
    nwin = max(1, Cmax // C)

    def run(n):
        for j in range(n):
            w = (j % nwin) * C
            be.mc_fidelity(ct[w:w + C], draws[w:w + C], N, 0, o, h0_diag=h0, out=out[w:w + C])
    run(max(10, nl // 4))
    torch.cuda.synchronize()
    v = []
    for r in range(args.reps):
        run(max(5, nl // 10))                               # lead-in behind the synchronisation gap
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); run(nl); e1.record(st)
        torch.cuda.synchronize()
        v.append(e0.elapsed_time(e1) / nl * 1e3)
    res[C] = float(np.median(v))
    print(f"N={N} 0->{o} C={C:5d} K={K}: {res[C]:9.2f} us per launch ({nl} launches)   {res[C] / (C * K) * 1e6:7.2f} us per 1e6 evaluations"
          f"   [{', '.join(f'{x:.2f}' for x in v)}]", flush=True)
Cs = np.array(sizes, dtype=float); T = np.array([res[c] for c in sizes])
a, b = np.polyfit(Cs, T, 1)
print(f"fit T(C) = a C + b over all sizes: a = {a:.4f} us per controller ({a * 100:.2f} us per 100 controllers = 1e6 evaluations), b = {b:.2f} us per launch")
big = Cs >= 1000
if big.sum() >= 2:
    a2 = (T[big][-1] - T[big][0]) / (Cs[big][-1] - Cs[big][0])
    print(f"slope between the two largest sizes: {a2 * 100:.2f} us per 1e6 evaluations; the 100-controller launch pays {res.get(100, float('nan')) - a2 * 100:.2f} us on top of it")
