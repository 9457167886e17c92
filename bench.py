#!/usr/bin/env python3
"""bench.py - MC fidelity evaluations / second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
  fidelity kernel over C x K samples of one sigma_sim level  ->  per-controller reductions (RIM_1, std,
  min, Q(0.95), Q(0.98) for centre / DKW-upper / DKW-lower)  [-> all-gather of the fidelity slabs, N > 1].

Workload (N = 1): BASELINE config 3 - the configuration the metric is quoted on - nspin=7, in=0, out=6,
100 controllers x 10 000 perturbations, sigma_sim = 0.05, complex128-equivalent fp64 arithmetic.
Inputs as SURVEY.md 8(d): controllers B ~ U(-10,10), T ~ U(2,30) from default_rng(20220714+3); draws from the
legacy stream `np.random.seed(12345)`: one burned draw, then sigma * standard_normal((C,K,N,3)).

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling - every rank owns its own block of
100 controllers (global problem = 100 N controllers x 10 000 draws).  The sample space is sharded by
CONTROLLER, so every per-controller fidelity vector is complete on its owner rank and the per-controller
reductions are rank-local; the exchange step is an RCCL all-gather of the per-controller metric rows (15 x 100
doubles per rank: RIM_1 / std / min / Q(0.95) / Q(0.98) x centre, DKW-upper, DKW-lower) so that every rank ends
each step with the full metric table (what the `.mcm` cache holds).  Pipeline: the fidelity kernels of GROUP = 16
consecutive steps run back-to-back on the main stream into one (16 C, K) block; a high-priority side stream then
reduces the block's 16 C controller rows in one launch and moves their metric rows in one collective while the main
stream fills the other block (ROBCHAR_BENCH_GROUP).  ROBCHAR_BENCH_GATHER=fid additionally all-gathers the raw
fidelity slabs (8 MB per rank per step; what `MCDataSim` does once per sigma level to write the `.mc` cache) - at
the kernel's speed that replication is xGMI-bound, so it is not part of the default timed step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fidelity kernel):
achieved = (24 N + 8) B x C x K / mean kernel time (HIP events on the launch stream), peak = 8 TB/s HBM.
`cpu_baseline` (N = 1 only) times the oracle's reference-shaped path (one dense complex expm per sample:
oracle/expm_port.c over all host cores, calibrated against the scipy.linalg.expm loop of
oracle/robchar_oracle.py:fidelity_expm_loop) on this box's host cores on a bounded sample of the workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

NSPIN, INSPIN, OUTSPIN = 7, 0, 6
NCTRL, NDRAW, SIGMA = 100, 10000, 0.05
CONFIG_ID = 3
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
BYTES_PER_EVAL = 24 * NSPIN + 8


def make_inputs(rank: int):
    rng = np.random.default_rng(20220714 + CONFIG_ID + 1000 * rank)
    ctrl = np.empty((NCTRL, NSPIN + 1))
    ctrl[:, :NSPIN] = rng.uniform(-10, 10, (NCTRL, NSPIN))
    ctrl[:, NSPIN] = rng.uniform(2, 30, NCTRL)
    np.random.seed(12345 + rank)
    np.random.normal(scale=SIGMA)                                   # the per-level burn (mcsim.py:425)
    draws = SIGMA * np.random.standard_normal((NCTRL, NDRAW, NSPIN, 3))
    return ctrl, draws


def cpu_baseline(ctrl, draws):
    """CPU baseline on this box's host cores, bounded sample of the same workload.

    Primary figure (kind "port"): oracle/expm_port.c - the reference's algorithm shape (dense complex H, one
    Pade scaling-and-squaring expm per sample, noise_model.py:98-109) in plain C, OpenMP over all host cores,
    on the FULL workload (10^6 evaluations, ~10 core-seconds).  For calibration against the Python reference
    the same path through scipy.linalg.expm (oracle.fidelity_expm_loop, one core, 20 000 evaluations) is timed
    as well and quoted in `sample`.
    """
    import ctypes
    import subprocess
    from oracle import robchar_oracle as orc
    lib_path = os.path.join(ROOT, "oracle", "librc_oracle_port.so")
    if not os.path.exists(lib_path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(lib_path)
    vp = ctypes.c_void_p
    lib.rc_oracle_expm_fidelity.argtypes = [ctypes.c_int] * 3 + [vp, vp, ctypes.c_int, vp, vp, ctypes.c_longlong,
                                                                 ctypes.c_longlong, vp, ctypes.c_int]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    out = np.empty((NCTRL, NDRAW))
    t0 = time.perf_counter()
    rc = lib.rc_oracle_expm_fidelity(NSPIN, INSPIN, OUTSPIN, None, None, 0, ctrl.ctypes.data, draws.ctypes.data,
                                     NCTRL, NDRAW, out.ctypes.data, cores)
    wall = time.perf_counter() - t0
    assert rc == 0
    # calibration: the scipy per-sample loop (what the reference executes), single core
    nc, nd = 2, 10000
    t1 = time.perf_counter()
    f_py = orc.fidelity_expm_loop(ctrl[:nc], draws[:nc, :nd], NSPIN, INSPIN, OUTSPIN)
    wall_py = time.perf_counter() - t1
    agree = float(np.abs(f_py - out[:nc, :nd]).max())
    return {"value": NCTRL * NDRAW / wall, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"full workload (100 x 10000 = 1e6 evals) through oracle/expm_port.c (dense complex expm per "
                      f"sample, Pade-13 scaling-squaring), OpenMP {cores} threads, wall {wall:.2f}s; calibration: "
                      f"scipy.linalg.expm per-sample loop (oracle.fidelity_expm_loop) {nc * nd} evals on 1 core = "
                      f"{nc * nd / wall_py:.0f} evals/s; the two agree to {agree:.1e}; the unmodified reference measured "
                      f"in the build container (SURVEY.md 6): 10.0 k evals/s per core kernel-only at N=7"}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.35 s of GPU work - the first ~10 ms after an idle period run at ramping clocks (200 steps: 82 us
    # per step, 2000+: 74 us)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", default="auto")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run "
                     "--nproc-per-node N (one rank per GPU)")
        args.gpus = world

    ctrl_np, draws_np = make_inputs(rank)
    cpu, cpu_fid = None, None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu, cpu_fid = cpu_baseline(ctrl_np, draws_np)

    import torch
    import torch.distributed as dist
    be = importlib.import_module("code-robchar_amd.backend")
    orc = importlib.import_module("oracle.robchar_oracle")

    ndev = torch.cuda.device_count()
    if ndev == 0:
        sys.exit("bench.py needs a GPU (there is no CPU path)")
    dev_index = local_rank % ndev           # one rank per GPU when launched by the driver (ndev >= world)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RCCL ("nccl") is the product path.  ROBCHAR_BENCH_BACKEND=gloo exists only to rehearse the multi-rank
    # control flow on a one-GPU box (RCCL refuses two ranks on one device): slabs then hop through host memory.
    backend = os.environ.get("ROBCHAR_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    ctrl = torch.from_numpy(ctrl_np).to(dev)
    draws = torch.from_numpy(draws_np).to(dev)          # resident in HBM before the timed region
    eps = orc.compute_dkw_error(0.05, NDRAW)            # scalar host arithmetic only
    gather_fid = os.environ.get("ROBCHAR_BENCH_GATHER", "metrics") == "fid"
    # ROBCHAR_BENCH_CDF=1 additionally sorts every controller's 10 000 fidelities (the exact ECDF) in the reduction
    # stage; the default step delivers the CDF at the reference's two thresholds (Q 0.95 / 0.98) like its `.mcm`
    with_cdf = os.environ.get("ROBCHAR_BENCH_CDF", "0") == "1"
    # Pipeline.  The fidelity kernels of GROUP consecutive steps are launched back-to-back on the main stream into
    # the GROUP slabs of one (GROUP*C, K) block; ONE event then hands the block to a high-priority side stream, which
    # reduces all GROUP*C controller rows in ONE launch (and, N > 1, moves their metric rows in ONE collective)
    # while the main stream is already filling the other block.  Every step's reductions are computed; only the
    # launches, event records and stream waits are amortised over the group: per-step hand-over cost 6-7 us of
    # the 85 us step (scripts/step_breakdown.py), and a collective costs tens of microseconds of latency whatever
    # its size (the payload is 12 KB per step per rank).
    GROUP = max(1, int(os.environ.get("ROBCHAR_BENCH_GROUP", os.environ.get("ROBCHAR_BENCH_GATHER_EVERY", "16"))))
    NBLK = 2
    GC = GROUP * NCTRL
    fid_blk = [torch.zeros((GC, NDRAW), dtype=torch.float64, device=dev) for _ in range(NBLK)]
    # metric rows of one block packed as rim1[3] std[3] min[3] q[3][2] -> (15, GROUP*C)
    packed = [torch.empty((15, GC), dtype=torch.float64, device=dev) for _ in range(NBLK)]
    views = [{"rim1": pk[0:3], "std": pk[3:6], "min": pk[6:9], "q": pk[9:15].view(3, 2, GC)} for pk in packed]
    all_metrics = [torch.empty((world * 15, GC), dtype=torch.float64, device=dev) for _ in range(NBLK)] if world > 1 else None
    gathered = [torch.empty((world * GC, NDRAW), dtype=torch.float64, device=dev) for _ in range(NBLK)] \
        if (world > 1 and gather_fid) else None
    main_stream = torch.cuda.current_stream(dev)
    # HIP events on the launch stream around every other FULL group of GROUP back-to-back fidelity launches (nothing
    # else is enqueued on that stream in between); kernel time = bracket / GROUP.  Bracketing single launches
    # perturbs them: the two markers add ~5 us to the 75 us kernel (rocprofv3 kernel trace vs events).
    n_grp = args.steps // GROUP
    k_start = {gi: torch.cuda.Event(enable_timing=True) for gi in range(0, n_grp, 2)}
    k_stop = {gi: torch.cuda.Event(enable_timing=True) for gi in k_start}
    if not k_start:                                     # fewer timed steps than one group: bracket single launches
        k_start = {("s", i): torch.cuda.Event(enable_timing=True) for i in range(args.steps)}
        k_stop = {k: torch.cuda.Event(enable_timing=True) for k in k_start}
    last = {}
    side_stream = torch.cuda.Stream(dev, priority=-1)
    blk_done = [torch.cuda.Event() for _ in range(NBLK)]
    side_done = [torch.cuda.Event() for _ in range(NBLK)]

    def step(i, timed_idx=None, final=False):
        """i counts from 0 within the current phase (warm-up / timed); a phase ends with a flush and a fence."""
        g, blk = i % GROUP, (i // GROUP) % NBLK
        if g == 0 and i >= NBLK * GROUP:
            main_stream.wait_event(side_done[blk])         # block `blk` has been reduced (and gathered): free again
        key_a = key_b = None
        if timed_idx is not None:
            if n_grp == 0:
                key_a = key_b = ("s", timed_idx)
            else:
                if g == 0 and (timed_idx // GROUP) in k_start:
                    key_a = timed_idx // GROUP
                if g == GROUP - 1 and (timed_idx // GROUP) in k_stop:
                    key_b = timed_idx // GROUP
        if key_a is not None:
            k_start[key_a].record(main_stream)
        be.mc_fidelity(ctrl, draws, NSPIN, INSPIN, OUTSPIN, out=fid_blk[blk][g * NCTRL:(g + 1) * NCTRL], kernel=args.kernel)
        if key_b is not None:
            k_stop[key_b].record(main_stream)
        last["g"], last["blk"] = g, blk
        if not (g == GROUP - 1 or final):
            return
        blk_done[blk].record(main_stream)
        with torch.cuda.stream(side_stream):
            side_stream.wait_event(blk_done[blk])
            # a final partial group reduces the whole block too (its unused slabs hold older steps' values)
            last["red"] = be.reduce_metrics(fid_blk[blk], dkw_eps=eps, out=views[blk], want_sorted=with_cdf)
            if world > 1:
                if backend == "nccl":
                    dist.all_gather_into_tensor(all_metrics[blk], packed[blk])
                    if gather_fid:
                        dist.all_gather_into_tensor(gathered[blk], fid_blk[blk])
                else:                                   # rehearsal only: hop through host memory
                    side_stream.synchronize()
                    host = torch.empty((world * 15, GC), dtype=torch.float64)
                    dist.all_gather_into_tensor(host, packed[blk].cpu())
                    all_metrics[blk].copy_(host)
                    if gather_fid:
                        host = torch.empty((world * GC, NDRAW), dtype=torch.float64)
                        dist.all_gather_into_tensor(host, fid_blk[blk].cpu())
                        gathered[blk].copy_(host)
            side_done[blk].record(side_stream)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i, final=(i == args.warmup - 1))
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, timed_idx=i, final=(i == args.steps - 1))     # the last step flushes its (possibly partial) group
    fence()
    elapsed = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    per = GROUP if n_grp else 1
    kern_ms = [k_start[k].elapsed_time(k_stop[k]) / per for k in k_start]
    kern_ms_mean = float(np.mean(kern_ms))

    # correctness of what was timed: subsample against the oracle, RIM against the tensor mean
    g, blk = last["g"], last["blk"]
    rows = slice(g * NCTRL, (g + 1) * NCTRL)
    f_host = fid_blk[blk][rows].cpu().numpy()
    sel = np.arange(0, NDRAW, 997)
    ref = orc.fidelity_eigh(ctrl_np[:8], draws_np[:8][:, sel], NSPIN, INSPIN, OUTSPIN)
    err = float(np.abs(f_host[:8][:, sel] - ref).max())
    rim_err = float(np.abs(last["red"]["rim1"][0][rows].cpu().numpy() - (1 - f_host).mean(axis=1)).max())
    if cpu_fid is not None:      # the CPU baseline computed the same 1e6 fidelities: compare all of them
        err = max(err, float(np.abs(f_host - cpu_fid).max()))
    ok = True
    if world > 1:
        ok = bool(torch.equal(all_metrics[blk].view(world, 15, GC)[rank], packed[blk]))
        # every rank must hold the same full table
        chk = torch.nan_to_num(all_metrics[blk]).sum().reshape(1).clone()
        lo, hi = chk.clone(), chk.clone()
        if backend != "nccl":
            lo, hi = lo.cpu(), hi.cpu()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ok = ok and bool((lo == hi).all())
        if gather_fid:
            ok = ok and bool(torch.equal(gathered[blk][rank * GC:(rank + 1) * GC], fid_blk[blk]))

    if rank == 0:
        evals_per_step = world * NCTRL * NDRAW
        value = evals_per_step * args.steps / elapsed
        traffic, valu = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                prof = json.load(open(tpath))
                traffic = prof.get("hbm_bytes_per_launch")
                valu = prof.get("valu_insts_per_launch")
            except Exception:
                traffic = None
        # the binding roof: fp64 VALU issue.  One wave-instruction occupies a SIMD for 4 cycles;
        # 1024 SIMDs x 2.4 GHz / 4 = 614 G wave-instructions/s at the nominal clock.
        fp64_flop = prof.get("fp64_flop_per_launch") if traffic else None
        fp64_valu = None
        if valu:
            rate = valu / (kern_ms_mean * 1e-3) / 1e9
            fp64_valu = {"valu_wave_insts_per_launch": valu, "achieved": rate, "peak": 614.4, "unit": "G wave-inst/s",
                         "frac": rate / 614.4,
                         "note": "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction; under the 1.4 kW socket "
                                 "power cap the kernel is clocked at ~1.5 GHz (384 G wave-inst/s), i.e. it fills "
                                 "essentially every issue slot the chip grants (instruction count from "
                                 "profiles/r01_pmc_sq.csv)"}
            if fp64_flop:
                # achieved FP64 FLOP/s against the vector-FP64 peak (78.6 TFLOP/s = every issue slot an
                # FMA at 2.4 GHz; the instruction mix here is 37 % FMA, 58 % add/mul, 5 % seeds)
                tf = fp64_flop / (kern_ms_mean * 1e-3) / 1e12
                fp64_valu.update({"fp64_flop_per_launch": fp64_flop, "achieved_tflops": tf, "peak_tflops": 78.6,
                                  "frac_tflops": tf / 78.6})
        achieved = BYTES_PER_EVAL * NCTRL * NDRAW / (kern_ms_mean * 1e-3) / 1e9
        line = {
            "metric": "MC fidelity evals/sec (N=7, 100 ctrls x 10k draws)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: nspin=7 in=0 out=6, 100 controllers x 10000 "
                                   "perturbations per GPU, sigma_sim=0.05, structured perturbation, chain",
                       "draws": "legacy numpy RandomState stream (seed 12345+rank), resident in HBM",
                       "step": f"fidelity kernel + per-controller RIM/std/min/Q reductions (launched once per {GROUP} steps)"
                               + (" + row sort (exact ECDF)" if with_cdf else "")
                               + (" + RCCL all-gather of the per-controller metric rows (overlapped)" if world > 1 else "")
                               + (" + all-gather of the raw fidelity slabs" if (world > 1 and gather_fid) else ""),
                       "kernel": args.kernel, "parallelism": f"controller-sharded x{world}",
                       "collective": ("none" if world == 1 else ("rccl all_gather_into_tensor" if backend == "nccl"
                                                                 else f"{backend} (rehearsal, host hop)"))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mc_fid_chain_kernel<7, 2>" if args.kernel in ("auto", "tridiag_adj") else args.kernel,
                         "kernel_ms": kern_ms_mean,
                         "kernel_ms_method": f"HIP events on the launch stream around groups of {per} back-to-back launches, / {per}",
                         "bytes_per_eval": BYTES_PER_EVAL,
                         "note": "algorithmic traffic is 176 B/eval (PMC-measured HBM traffic = 1.001x that); the kernel is "
                                 "bound by fp64 VALU instruction count at the ~1.5 GHz the chip holds under its 1.4 kW power "
                                 "cap (rocm-smi: 1.37 kW during the kernel), not by HBM (DESIGN.md 4)"},
            "fp64_valu": fp64_valu,
            "cpu_baseline": cpu,
            "check": {"max_abs_err_vs_oracle": err, "rim_err": rim_err, "gather_ok": ok},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    if err > 1e-10 or rim_err > 1e-10 or not ok:
        sys.exit("bench: parity check failed")


if __name__ == "__main__":
    main()
