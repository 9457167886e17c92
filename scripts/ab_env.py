#!/usr/bin/env python3
"""Same-process A/B of ONE library under different values of an environment switch that the library reads per call
(ROBCHAR_PHILOX_FUSED, ...; round 5 used it for the staggered-fill experiment's ROBCHAR_STAGGER, which is not in the tree: NOTEBOOK 12 (i)): the BASELINE shapes, launches interleaved value by value, HIP events on the launch
stream around brackets of back-to-back launches (the method of scripts/ab_rounds.py's kernel part).

usage: python3 scripts/ab_env.py --env ROBCHAR_STAGGER=0,4,6,8 [--reps 4] [--launches 400] [--out gpurun_out/x.txt]
The first value is the reference the others are compared with.
"""
import argparse, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--env", required=True, help="NAME=v0,v1,...")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--launches", type=int, default=400)
ap.add_argument("--out", default=None)
args = ap.parse_args()
name, vals = args.env.split("=")
vals = vals.split(",")
be = importlib.import_module("code-robchar_amd.backend")
from oracle import robchar_oracle as orc
fh = open(args.out, "a") if args.out else None


def log(s):
    print(s, flush=True)
    if fh:
        fh.write(s + "\n"); fh.flush()


log(f"# scripts/ab_env.py {' '.join(sys.argv[1:])}   ({time.strftime('%Y-%m-%d %H:%M:%S')}), library {importlib.import_module('code-robchar_amd._lib').LIB_PATH}")
rng = np.random.default_rng(20220714 + 3)


def uniform(C, N):
    x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C)
    return x


z = np.load(os.path.join(ROOT, "tests", "golden", "lbfgs_n7.npz"))
shipped = np.ascontiguousarray(z["ctrl_0-6"][np.arange(100) % z["ctrl_0-6"].shape[0]])
work = [("c3 uniform N=7 0->6 100x10000", 7, 0, 6, uniform(100, 7), None),
        ("c3 shipped L-BFGS N=7 0->6", 7, 0, 6, shipped, None),
        ("c4 shape N=7 0->3 100x10000", 7, 0, 3, uniform(100, 7), None),
        ("c2 N=5 0->4 100x10000", 5, 0, 4, uniform(100, 5), None),
        ("c5 N=10 XXZ 0->9 100x10000", 10, 0, 9, uniform(100, 10), np.ascontiguousarray(orc.xxz_delta(10)))]
st = torch.cuda.current_stream()
for label, N, a, b, ctrl_np, h0 in work:
    C, K = ctrl_np.shape[0], 10000
    ctrl = torch.from_numpy(ctrl_np).cuda()
    draws = [torch.from_numpy(0.05 * np.random.default_rng(100 + t).standard_normal((C, K, N, 3))).cuda() for t in range(3)]
    outs = {v: torch.empty((C, K), dtype=torch.float64, device="cuda") for v in vals}

    def run(v, n):
        os.environ[name] = v
        for j in range(n):
            be.mc_fidelity(ctrl, draws[j % 3], N, a, b, h0_diag=h0, out=outs[v])
    for v in vals:
        run(v, 30)
    run(vals[0], 1500)
    torch.cuda.synchronize()
    res = {v: [] for v in vals}
    for r in range(args.reps):
        for v in vals:
            run(v, 60)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); run(v, args.launches); e1.record(st)
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / args.launches * 1e3)
    ref = outs[vals[0]]
    sub = draws[(args.launches - 1) % 3][:8, ::97].cpu().numpy()
    err = float(np.abs(ref[:8, ::97].cpu().numpy() - orc.fidelity_eigh(ctrl_np[:8], sub, N, a, b, h0_diag=h0)).max())
    log(f"{label}  (us per launch of 1e6 evaluations, {args.launches} launches per figure; max|dF| vs oracle {err:.1e})")
    for v in vals:
        x = res[v]
        log(f"    {name}={v:>3}: " + "  ".join(f"{t:7.2f}" for t in x) + f"   median {np.median(x):7.2f}   vs {vals[0]} {np.median(x) / np.median(res[vals[0]]):.4f}"
            f"   max|dF - ref| {float((outs[v] - ref).abs().max().item()):.1e}")
