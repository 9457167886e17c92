#!/usr/bin/env python3
"""Kernel-only timing of the fidelity kernel for a few (N, C, K) shapes (development aid, not the bench).
usage: python scripts/kbench.py [--kernel auto] [--reps 20] [--shapes 5:100:10000,7:100:10000,10:100:10000]"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
be = importlib.import_module("code-robchar_amd.backend")

ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="auto")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--shapes", default="5:100:10000,7:100:10000,10:100:10000")
ap.add_argument("--sigma", type=float, default=0.05)
ap.add_argument("--out", default="end", help="end | mid | <site index>")
ap.add_argument("--device-draws", action="store_true", help="Philox draws generated on the device (shapes too large for host RNG)")
ap.add_argument("--xxz", action="store_true", help="XXZ diagonal offsets (BASELINE config 5)")
ap.add_argument("--ring", action="store_true", help="ring topology (noise_model.py:83-85)")
ap.add_argument("--shipped", action="store_true", help="N = 7 only: the reference's shipped L-BFGS controllers (tests/golden/lbfgs_n7.npz: "
                "0->6, 57 rows, or 0->3, 100 rows, tiled to C) instead of uniform random biases - near mirror-symmetric, close eigenvalue pairs")
args = ap.parse_args()
for shp in args.shapes.split(","):
    N, C, K = (int(v) for v in shp.split(":"))
    rng = np.random.default_rng(N)
    ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
    if args.shipped:
        assert N == 7, "--shipped: the fixture holds N = 7 controllers"
        rows = np.load(os.path.join(ROOT, "tests", "golden", "lbfgs_n7.npz"))["ctrl_0-6" if args.out == "end" else "ctrl_0-3"]
        ctrl = np.ascontiguousarray(rows[np.arange(C) % rows.shape[0]])
    if args.device_draws:
        draws = be.philox_normal((C, K, N, 3), seed=N, scale=args.sigma, as_torch=True)
    else:
        draws = torch.from_numpy(args.sigma * rng.standard_normal((C, K, N, 3))).cuda()
    ct = torch.from_numpy(ctrl).cuda()
    out = torch.empty((C, K), dtype=torch.float64, device="cuda")
    o = N - 1 if args.out == "end" else (N // 2 if args.out == "mid" else int(args.out))
    from oracle import robchar_oracle as orc
    h0 = orc.xxz_delta(N) if args.xxz else None
    be.general_path_tiles(reset=True)
    for _ in range(3):
        be.mc_fidelity(ct, draws, N, 0, o, h0_diag=h0, out=out, kernel=args.kernel, ring=args.ring)
    torch.cuda.synchronize()
    gen_tiles = be.general_path_tiles() / 3.0
    pol0 = be.polish_tiles(reset=True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
    for a, b in ev:
        a.record(); be.mc_fidelity(ct, draws, N, 0, o, h0_diag=h0, out=out, kernel=args.kernel, ring=args.ring); b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    pol = be.polish_tiles() / args.reps
    sel = np.arange(0, K, max(1, K // 50))
    ref = orc.fidelity_eigh(ctrl[:6], draws[:6][:, sel].cpu().numpy(), N, 0, o, h0_diag=h0, ring=args.ring)
    err = np.abs(out[:6][:, sel].cpu().numpy() - ref).max()
    print(f"N={N} C={C} K={K} kernel={args.kernel}: median {np.median(ms)*1e3:.1f} us  min {ms.min()*1e3:.1f} us  "
          f"-> {C*K/np.median(ms)/1e-3/1e9:.3f} G evals/s, {(24*N+8)*C*K/np.median(ms)/1e-3/1e9:.0f} GB/s algorithmic  max|err| {err:.1e}  general-path tiles/launch {gen_tiles:.1f}, off one-step path {pol:.0f} of {C * ((K + 63) // 64)}")
