#!/usr/bin/env python3
"""Same-box A/B of the FINAL builds of earlier rounds against the working tree (round 5).

The driver's BENCH_r02 / r03 / r04 read kernel_ms 54.0 -> 55.3 -> 56.5 us (roofline 0.407 -> 0.398 -> 0.389) while the commit
log claimed gains on the builder's boxes; this script puts the builds on ONE box:

  part 1 (`kernel`)  one process, the rounds' librobchar_hip.so loaded side by side through ctypes (the enqueue entry
                     rc_mc_fidelity_f64_async has had the same signature since ABI 2), the SAME device tensors, launches
                     interleaved build by build, HIP events on the launch stream: the kernels and nothing else.
  part 2 (`bench`)   every round's OWN tree (its bench.py, its package, its library - `git archive <round's last commit>` under
                     build/rounds/rNN, built there) under the driver's command line `python3 bench.py --gpus 1 --steps 20
                     --warmup 5` and under the 4 000-step default, interleaved --reps times: kernel_ms, ms_per_step,
                     cold_20_steps_kernel_ms, the untimed steady-state figure.

usage: python3 scripts/ab_rounds.py [--parts kernel,bench] [--reps 3] [--out gpurun_out/r05_ab_rounds.txt]
Trees: build/rounds/r02, r03, r04 (scripts/export_rounds.sh makes them) + the working tree as r05.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def trees():
    out = []
    for n in (2, 3, 4):
        d = os.path.join(ROOT, "build", "rounds", f"r0{n}")
        if os.path.exists(os.path.join(d, "code-robchar_amd", "csrc", "librobchar_hip.so")):
            out.append((f"r0{n}", d))
    out.append(("r05", ROOT))
    return out


def kernel_part(reps, launches, log, lib_paths=None, ring=False):
    """`lib_paths`: [(name, path to a librobchar_hip.so)] - default: the rounds' trees.  The LAST entry is the reference the
    others are compared with."""
    import torch
    from oracle import robchar_oracle as orc
    dev = torch.device("cuda", 0)
    libs = []
    if lib_paths is None:
        lib_paths = [(name, os.path.join(d, "code-robchar_amd", "csrc", "librobchar_hip.so")) for name, d in trees()]
    for name, path in lib_paths:
        lib = ctypes.CDLL(path)
        vp, ll, i = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
        lib.rc_mc_fidelity_f64_async.argtypes = [i, vp, i, i, i, i, vp, vp, i, vp, vp, ll, ll, vp]
        lib.rc_mc_fidelity_f64_async.restype = i
        lib.rc_version.restype = i
        libs.append((name, lib))
    log("kernel part: libraries " + ", ".join(f"{n} (ABI {l.rc_version()})" for n, l in libs))
    rng = np.random.default_rng(20220714 + 3)

    def uniform(C, N):
        x = np.empty((C, N + 1))
        x[:, :N] = rng.uniform(-10, 10, (C, N))
        x[:, N] = rng.uniform(2, 30, C)
        return x

    z = np.load(os.path.join(ROOT, "tests", "golden", "lbfgs_n7.npz"))
    shipped = np.ascontiguousarray(z["ctrl_0-6"][np.arange(100) % z["ctrl_0-6"].shape[0]])
    hf = np.load(os.path.join(ROOT, "tests", "golden", "highfid.npz"))
    work = [("c3 uniform N=7 0->6 100x10000", 7, 0, 6, uniform(100, 7), None),
            ("c3 shipped L-BFGS N=7 0->6", 7, 0, 6, shipped, None),
            ("c4 shape N=7 0->3 100x10000", 7, 0, 3, uniform(100, 7), None),
            ("c2 N=5 0->4 100x10000", 5, 0, 4, uniform(100, 5), None),
            ("c5 N=10 XXZ 0->9 100x10000", 10, 0, 9, uniform(100, 10), np.ascontiguousarray(orc.xxz_delta(10))),
            ("c5 constructed (delocalised) N=10 XXZ 0->9", 10, 0, 9, np.ascontiguousarray(hf["c5_ctrl"]), np.ascontiguousarray(hf["c5_h0_diag"]))]
    if ring:                                     # ring topology (noise_model.py:83-85), AUTO = mixed route + repair launch
        work = [(f"ring N={n} 0->{n // 2} 100x10000", n, 0, n // 2, uniform(100, n), None) for n in (5, 7, 10)]
    st = torch.cuda.current_stream(dev)
    rows = []
    for label, N, a, b, ctrl_np, h0 in work:
        C, K = ctrl_np.shape[0], 10000
        ctrl = torch.from_numpy(ctrl_np).to(dev)
        draws = [torch.from_numpy(0.05 * np.random.default_rng(100 + t).standard_normal((C, K, N, 3))).to(dev) for t in range(3)]
        outs = {n: torch.empty((C, K), dtype=torch.float64, device=dev) for n, _ in libs}

        def run(lib, out, n):
            for j in range(n):
                rc = lib.rc_mc_fidelity_f64_async(0, ctypes.c_void_p(st.cuda_stream), 0, N, a, b,
                                                  ctypes.c_void_p(h0.ctypes.data) if h0 is not None else None, None, int(ring),
                                                  ctypes.c_void_p(ctrl.data_ptr()), ctypes.c_void_p(draws[j % 3].data_ptr()), C, K,
                                                  ctypes.c_void_p(out.data_ptr()))
                assert rc == 0
        # warm every build, then a 0.1 s clock pre-roll on the first one
        for n, lib in libs:
            run(lib, outs[n], 30)
        run(libs[0][1], outs[libs[0][0]], 1500)
        torch.cuda.synchronize(dev)
        res = {n: [] for n, _ in libs}
        for r in range(reps):
            for n, lib in libs:
                run(lib, outs[n], 60)                                   # the build's own short lead-in (instruction cache, clocks)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                run(lib, outs[n], launches)
                e1.record(st)
                torch.cuda.synchronize(dev)
                res[n].append(e0.elapsed_time(e1) / launches * 1e3)
        # the builds must agree with each other on what they computed (last launch used draws[(launches - 1) % 3])
        ref = outs[libs[-1][0]]
        dmax = {n: float((outs[n] - ref).abs().max().item()) for n, _ in libs}
        sub = draws[(launches - 1) % 3][:8, ::97].cpu().numpy()
        want = orc.fidelity_eigh(ctrl_np[:8], sub, N, a, b, h0_diag=h0, ring=ring)
        err = float(np.abs(ref[:8, ::97].cpu().numpy() - want).max())
        log(f"{label}  (us per 1e6 evaluations, {launches} launches per figure; max|dF| {libs[-1][0]} vs oracle {err:.1e})")
        for n, _ in libs:
            v = res[n]
            log(f"    {n}: " + "  ".join(f"{x:7.2f}" for x in v) + f"   median {np.median(v):7.2f}   vs {libs[-1][0]} {np.median(v) / np.median(res[libs[-1][0]]):.4f}"
                f"   max|dF - {libs[-1][0]}| {dmax[n]:.1e}")
        rows.append((label, {n: float(np.median(res[n])) for n, _ in libs}))
    return rows


def bench_part(reps, log):
    def one(tree, args):
        env = dict(os.environ, ROBCHAR_TEST_NO_BUILD="1")
        env.pop("ROBCHAR_HIP_LIB", None)
        t0 = time.time()
        r = subprocess.run([sys.executable, "bench.py"] + args, cwd=tree, env=env, capture_output=True, text=True)
        line = next((l for l in r.stdout.splitlines() if l.startswith('{"metric"')), None)
        if line is None:
            return {"error": (r.stderr or r.stdout)[-400:], "wall": time.time() - t0}
        d = json.loads(line)
        also = d.get("also", {})
        cold = also.get("cold_20_steps_kernel_ms", {})
        ship = also.get("shipped_lbfgs_controllers", {})
        steady = (d["roofline"].get("steady_state_untimed") or {}).get("kernel_ms")
        return {"kernel_us": d["roofline"]["kernel_ms"] * 1e3, "step_us": d["ms_per_step"] * 1e3, "frac": d["roofline"]["frac"],
                "cold_us": cold.get("kernel_ms", float("nan")) * 1e3 if isinstance(cold, dict) and "kernel_ms" in cold else float("nan"),
                "steady_us": steady * 1e3 if steady else float("nan"),
                "shipped_us": ship.get("kernel_ms", float("nan")) * 1e3 if isinstance(ship, dict) and "kernel_ms" in ship else float("nan"),
                "wall": time.time() - t0, "rc": r.returncode}
    modes = [("driver: --gpus 1 --steps 20 --warmup 5", ["--gpus", "1", "--steps", "20", "--warmup", "5"]),
             ("long: --steps 4000 --warmup 400 --no-cpu-baseline --no-end-to-end --no-also",
              ["--steps", "4000", "--warmup", "400", "--no-cpu-baseline", "--no-end-to-end", "--no-also"])]
    for label, args in modes:
        log(f"bench part, {label}")
        acc = {}
        for r in range(reps):
            for name, d in trees():
                res = one(d, args)
                acc.setdefault(name, []).append(res)
                if "error" in res:
                    log(f"    rep {r} {name}: FAILED {res['error']!r}")
                else:
                    log(f"    rep {r} {name}: kernel {res['kernel_us']:6.2f} us  step {res['step_us']:6.2f} us  frac {res['frac']:.4f}  "
                        f"cold20 {res['cold_us']:6.2f}  steady(untimed) {res['steady_us']:6.2f}  shipped {res['shipped_us']:6.2f}  "
                        f"[{res['wall']:.0f} s, rc {res['rc']}]")
        for name, v in acc.items():
            ok = [x for x in v if "error" not in x]
            if ok:
                med = lambda k: float(np.nanmedian([x[k] for x in ok]))
                log(f"  median {name}: kernel {med('kernel_us'):6.2f} us  step {med('step_us'):6.2f} us  frac {med('frac'):.4f}  "
                    f"cold20 {med('cold_us'):6.2f}  steady {med('steady_us'):6.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="kernel,bench")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--ring", action="store_true", help="kernel part on ring-topology workloads (N = 5, 7, 10)")
    ap.add_argument("--libs", default=None, help="kernel part on these builds instead of the rounds': name=path,name=path (last = reference)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_ab_rounds.txt"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    fh = open(args.out, "a")

    def log(s):
        print(s, flush=True)
        fh.write(s + "\n")
        fh.flush()

    log(f"# scripts/ab_rounds.py {' '.join(sys.argv[1:])}   ({time.strftime('%Y-%m-%d %H:%M:%S')})")
    log("# trees: " + ", ".join(f"{n}={os.path.relpath(d, ROOT)}" for n, d in trees()))
    parts = args.parts.split(",")
    if "bench" in parts:                       # child processes first: this process has not touched the GPU yet
        bench_part(args.reps, log)
    if "kernel" in parts:
        lp = None
        if args.libs:
            lp = [(kv.split("=")[0], os.path.join(ROOT, kv.split("=")[1])) for kv in args.libs.split(",")]
        kernel_part(args.reps, args.launches, log, lp, ring=args.ring)


if __name__ == "__main__":
    main()
