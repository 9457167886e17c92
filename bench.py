#!/usr/bin/env python3
"""bench.py - MC fidelity evaluations / second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
  fidelity kernel over the rank's C x K samples of one sigma_sim level  ->  per-controller reductions (RIM_1, std,
  min, Q(0.95), Q(0.98) for centre / DKW-upper / DKW-lower)  [-> RCCL all-gather of the metric rows, N > 1].

--config 3 (default; the configuration the metric is quoted on): BASELINE config 3 - nspin=7, in=0, out=6,
  100 controllers x 10 000 perturbations, sigma_sim = 0.05, fp64.  Inputs as SURVEY.md 8(d): controllers
  B ~ U(-10,10), T ~ U(2,30) from default_rng(20220714+3); draws from the legacy stream `np.random.seed(12345+...)`:
  one burned draw, then sigma * standard_normal((C,K,N,3)).  THREE distinct draw tensors (504 MB in all, more than the
  256 MiB Infinity Cache) are rotated step by step, so the read stream cannot live in a cache.  N > 1: WEAK scaling -
  every rank owns its own 100 controllers (global problem = 100 N controllers x 10 000 draws).
--config 4: BASELINE config 4, the configuration BASELINE.json designates for 8 GPUs - nspin=7, in=0, out=3,
  1000 controllers x 100 000 perturbations = 1e8 evaluations per step, counter-based device draws (2.1e9 normals =
  16.8 GB, generated once before the timed region, resident).  N > 1: STRONG scaling - rank r owns controllers
  [r C/N, (r+1) C/N) and exactly that slice of the Philox stream (by element offset: the result does not depend on N).
  The default (config 3) run also appends a short config-4 run as `also.config4_strong` and, under N > 1, config 3's own
  1e6 evaluations split over the ranks (`--config 30`, `also.config3_strong`): one invocation per N records weak c3,
  strong c3 and strong c4.  (config 4's timed region starts with the draws resident, as the bench contract prescribes;
  `end_to_end.c4_level_api` is the product API's figure for the same workload, draw generation included.)
--config 40: config 4 with the counter-based draws generated INSIDE the fidelity kernel in every step (no resident draw tensor):
  the whole sigma level per step; appended to the default run as `also.config4_strong_fused` (same metric table as config 4's,
  bit for bit).
--config 2 / 5: the other two GPU configurations of BASELINE.json (N = 5 weak; N = 10 XXZ strong).

In both configurations the sample space is sharded by CONTROLLER, so every per-controller fidelity vector is complete
on its owner rank and the per-controller reductions are rank-local; the exchange step is an RCCL all-gather of the
per-controller metric rows (15 doubles per controller: what the `.mcm` cache holds) so that every rank ends each step
with the full metric table.  Pipeline: the fidelity kernels of GROUP consecutive steps run back-to-back on the main
stream into one block; a side stream then reduces the block's rows in one launch and moves their
metric rows in one collective while the main stream fills the other block (config 3: GROUP = 16; config 4: 1).
ROBCHAR_BENCH_GATHER=fid additionally all-gathers the raw fidelity slabs; under N > 1 the default run appends that
variant of config 4 as `also.config4_strong_gather_fid` (north_star's "reassemble per-controller fidelity vectors").

Launching: `python3 bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh
child processes, one per GPU, before this process touches the GPU) and relays rank 0's line and the ranks' exit code;
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the launcher's ranks are used as they
are.  `config.rccl` records what the communicator saw: world size, backend, and the device of every rank.
Exit codes: 0 = headline and every appended leg fine; 1 = a parity check failed; 3 = headline fine and printed, but an
appended leg raised or the extras watchdog had to end the run (`extras_failed` in the line names the leg).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fidelity kernel): achieved =
(24 N + 8) B x evaluations per launch / mean kernel time (HIP events on the launch stream), peak = 8 TB/s HBM.
`cpu_baseline` (N = 1 only) times the oracle's reference-shaped path (one dense complex expm per sample:
oracle/expm_port.c over all host cores, calibrated against the scipy.linalg.expm loop) on this box's host cores.
`end_to_end` (rank 0 reports; N = 1 runs all of it) times the PRODUCT API - `MCDataSim.get_metrics_dict()` cold, draws
and cache files included - next to the kernel-only headline.
"""
import argparse
import importlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

# dmabuf IPC is what RCCL / cross-process device memory need on this pool's hosts; HIP/HSA reads the variable ONCE at runtime
# initialisation, so it is set here - at import, before torch or any HIP call can have happened in this process - and not
# next to init_process_group (round 3 set it after torch.cuda.device_count(): too late for the torch.distributed.run path)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
SIGMA = 0.05
PARITY_TOL = 1e-10           # north_star: fidelities and RIM values within 1e-10 of the reference's CPU path


def within(err, tol=PARITY_TOL) -> bool:
    """Parity gate.  Written as `err <= tol` and negated by the callers: a NaN error (a repair launch that did not run
    leaves NaN fidelities) FAILS the gate - `err > tol` would let it pass."""
    return bool(err <= tol)

def teeth(f_all, got, want):
    """What gives a parity figure its weight: the size of the fidelities it was measured on.  `f_all`: the fidelities of the
    timed launch (their median and the share above 1e-3); `got` / `want`: the compared subsample (max relative error over the
    samples with a reference fidelity above 1e-3; None when there is none - then the absolute bound is all there is)."""
    f_all, got, want = np.asarray(f_all, dtype=np.float64), np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    fin = f_all[np.isfinite(f_all)]
    big = np.isfinite(want) & (want > 1e-3)
    rel = float((np.abs(got - want)[big] / want[big]).max()) if big.any() else None
    return {"median_fidelity": float(np.median(fin)) if fin.size else None,
            "frac_F_gt_1e-3": float((fin > 1e-3).mean()) if fin.size else None,
            "max_rel_err_F_gt_1e-3": rel, "compared_samples": int(want.size), "compared_samples_F_gt_1e-3": int(big.sum())}


REL_TOL = 1e-9               # relative bound on compared samples with F > 1e-3 (measured ~1e-13); the absolute one is PARITY_TOL


CONFIGS = {
    2: dict(N=5, inspin=0, outspin=4, C=100, K=10000, scaling="weak", group=16, rotate=3, xxz=False, draws="legacy",
            label="BASELINE config 2: nspin=5 in=0 out=4, 100 controllers x 10000 perturbations per GPU, "
                  "sigma_sim=0.05, structured perturbation, chain"),
    5: dict(N=10, inspin=0, outspin=9, C=100, K=10000, scaling="strong", group=16, rotate=3, xxz=True, draws="legacy",
            label="BASELINE config 5: nspin=10 XXZ (Z enabled: h0_diag = qnewton.py:148-150), in=0 out=9, 100 controllers x "
                  "10000 perturbations in all (controller-sharded over the GPUs), sigma_sim=0.05"),
    3: dict(N=7, inspin=0, outspin=6, C=100, K=10000, scaling="weak", group=16, rotate=3, xxz=False, draws="legacy",
            label="BASELINE config 3: nspin=7 in=0 out=6, 100 controllers x 10000 perturbations per GPU, "
                  "sigma_sim=0.05, structured perturbation, chain"),
    4: dict(N=7, inspin=0, outspin=3, C=1000, K=100000, scaling="strong", group=1, rotate=1, xxz=False, draws="philox",
            label="BASELINE config 4: nspin=7 in=0 out=3, 1000 controllers x 100000 perturbations in all "
                  "(controller-sharded over the GPUs), sigma_sim=0.05, structured perturbation, chain"),
}
# BASELINE.json's metric is quoted on config 3 "with 1/2/4/8-GPU scaling": besides the weak-scaling headline (100 x 10 000 per
# GPU) the SAME 1e6 evaluations split by controller over the ranks (13/13/13/13/12/12/12/12 at 8: ~7 us of kernel per rank
# and step - launch-bound, which is exactly what that curve shows).  `--config 30`; appended to the default run under N > 1
# as `also.config3_strong`.
# BASELINE config 4 with the counter-based draws generated INSIDE the fidelity kernel (round 4: mc_fid_chain_philox_kernel): no
# resident draw tensor, a step = one launch that also makes its 2.1e9 draws - the whole sigma level, not just its second half.
# `--config 40`; appended to the default run as `also.config4_strong_fused`.
CONFIGS[40] = dict(CONFIGS[4], draws="philox_fused", seed_id=4,
                   label="BASELINE config 4 with the draws generated inside the fidelity kernel: nspin=7 in=0 out=3, 1000 controllers x "
                         "100000 perturbations in all (controller-sharded over the GPUs), sigma_sim=0.05, counter-based stream")
CONFIGS[30] = dict(CONFIGS[3], scaling="strong", seed_id=3,
                   label="BASELINE config 3, STRONG scaling: nspin=7 in=0 out=6, 100 controllers x 10000 perturbations in "
                         "all (controller-sharded over the GPUs), sigma_sim=0.05, structured perturbation, chain")


def make_controllers(config_id: int, n_ctrl: int, nspin: int, rank: int = 0):
    rng = np.random.default_rng(20220714 + config_id + 1000 * rank)
    ctrl = np.empty((n_ctrl, nspin + 1))
    ctrl[:, :nspin] = rng.uniform(-10, 10, (n_ctrl, nspin))
    ctrl[:, nspin] = rng.uniform(2, 30, n_ctrl)
    return ctrl


def legacy_draws(seed: int, C: int, K: int, N: int):
    np.random.seed(seed)
    np.random.normal(scale=SIGMA)                                   # the per-level burn (mcsim.py:425)
    return SIGMA * np.random.standard_normal((C, K, N, 3))


def cpu_baseline(cfg, ctrl, draws):
    """CPU baseline on this box's host cores, bounded sample of the same workload (config 3: all 1e6 evaluations).

    Primary figure (kind "port"): oracle/expm_port.c - the reference's algorithm shape (dense complex H, one
    Pade scaling-and-squaring expm per sample, noise_model.py:98-109) in plain C, OpenMP over all host cores.  For
    calibration against the Python reference the same path through scipy.linalg.expm (oracle.fidelity_expm_loop, one
    core, 20 000 evaluations) is timed as well and quoted in `sample`.
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
    N, C, K = cfg["N"], ctrl.shape[0], draws.shape[1]
    out = np.empty((C, K))
    t0 = time.perf_counter()
    h0 = np.ascontiguousarray(orc.xxz_delta(N)) if cfg["xxz"] else None
    rc = lib.rc_oracle_expm_fidelity(N, cfg["inspin"], cfg["outspin"], h0.ctypes.data if h0 is not None else None, None, 0,
                                     ctrl.ctypes.data, draws.ctypes.data, C, K, out.ctypes.data, cores)
    wall = time.perf_counter() - t0
    assert rc == 0
    nc, nd = 2, min(K, 10000)
    t1 = time.perf_counter()
    f_py = orc.fidelity_expm_loop(ctrl[:nc], draws[:nc, :nd], N, cfg["inspin"], cfg["outspin"], h0_diag=h0)
    wall_py = time.perf_counter() - t1
    agree = float(np.abs(f_py - out[:nc, :nd]).max())
    # the unmodified reference measured in the build container: evaluate_noisy_fidelity per core
    ref_container = {5: 14.4e3, 7: 10.0e3, 10: 8.5e3}.get(N)
    return {"value": C * K / wall, "unit": "evals/s", "cores": cores, "kind": "port",
            "wall_s": round(wall, 3),
            "scipy_loop_evals_per_s_1core": float(f"{nc * nd / wall_py:.5g}"),
            "scipy_loop_sample_evals": nc * nd,
            "port_vs_scipy_max_abs_diff": agree,
            "port_evals_per_s_per_core": float(f"{C * K / wall / cores:.5g}"),
            "reference_in_container_evals_per_s": ref_container,
            "sample": f"{C} x {K} = {C * K:.0e} evals of the workload through oracle/expm_port.c (dense complex expm per "
                      f"sample, Pade-13 scaling-squaring), OpenMP {cores} threads; calibration fields: the same path through "
                      f"scipy.linalg.expm sample by sample (oracle.fidelity_expm_loop, SURVEY.md 8(d)(i)'s shape) on 1 core, "
                      f"{nc * nd} evals; reference_in_container = the unmodified reference's evaluate_noisy_fidelity per core "
                      f"as measured in the build container (2.1 GHz Xeon vCPU; the reference itself does not travel to this box)"}, out


class Env:
    """Process-group plumbing shared by every leg of the benchmark."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is not None     # set at import, ahead of any HIP initialisation
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        ndev = torch.cuda.device_count()
        if ndev == 0:
            sys.exit("bench.py needs a GPU (there is no CPU path)")
        self.dev_index = self.local_rank % ndev      # one rank per GPU when launched by the driver (ndev >= world)
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        # RCCL ("nccl") is the product path.  ROBCHAR_BENCH_BACKEND=gloo exists only to rehearse the multi-rank
        # control flow on a one-GPU box (RCCL refuses two ranks on one device): tensors then hop through host memory.
        self.backend = os.environ.get("ROBCHAR_BENCH_BACKEND", "nccl")
        # ROBCHAR_BENCH_FORCE_PG=1: create the process group and run every collective even with ONE rank - the only way
        # to execute the RCCL code path on a one-GPU box (communicator set-up, all_gather_into_tensor on the side stream)
        self.collective = self.world > 1 or os.environ.get("ROBCHAR_BENCH_FORCE_PG", "0") == "1"
        if self.collective:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)

    def comm_info(self):
        """What the COMMUNICATOR saw (not what the environment said): world size and backend from torch.distributed, and
        the device every rank computes on (uuid / PCI bus id, all-gathered) - N distinct entries prove N distinct GPUs."""
        torch = self.torch
        p = torch.cuda.get_device_properties(self.dev_index)
        ident = {"rank": self.rank, "index": self.dev_index, "name": p.name}
        for attr in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
            v = getattr(p, attr, None)
            if v is not None:
                ident[attr] = str(v)
        if not self.collective:
            return {"world": 1, "backend": None, "devices": [ident], "distinct_devices": 1}
        box = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(box, ident)
        keys = {(d.get("uuid"), d.get("pci_bus_id"), d.get("pci_domain_id"), d.get("index")) for d in box}
        return {"world": self.dist.get_world_size(), "backend": self.dist.get_backend(), "devices": box,
                "distinct_devices": len(keys)}

    def all_gather(self, out, shard):
        """all_gather_into_tensor on the CURRENT stream (RCCL), or through host memory (gloo rehearsal)."""
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(out, shard)
        else:
            self.torch.cuda.current_stream(self.dev).synchronize()
            host = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, shard.cpu())
            out.copy_(host)

    def fence(self):
        self.torch.cuda.synchronize(self.dev)
        if self.collective:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, x: float) -> float:
        if not self.collective:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def run_pipeline(env, be, orc, config_id, steps, warmup, kernel, preroll_s=None, gather_fid=None):
    """The timed benchmark of one configuration.  Returns (json fields, check dict).  `gather_fid`: also all-gather the
    raw per-controller fidelity slabs in every step (default: the ROBCHAR_BENCH_GATHER=fid switch)."""
    torch = env.torch
    cfg = CONFIGS[config_id]
    config_id = cfg.get("seed_id", config_id)          # (config 30 = config 3's inputs, sharded instead of replicated)
    N, a, b, K = cfg["N"], cfg["inspin"], cfg["outspin"], cfg["K"]
    world, rank, dev = env.world, env.rank, env.dev
    if gather_fid is None:
        gather_fid = os.environ.get("ROBCHAR_BENCH_GATHER", "metrics") == "fid"
    # ROBCHAR_BENCH_CDF=1 additionally sorts every controller's fidelities (the exact ECDF) in the reduction stage;
    # the default step delivers the CDF at the reference's two thresholds (Q 0.95 / 0.98) like its `.mcm`
    with_cdf = os.environ.get("ROBCHAR_BENCH_CDF", "0") == "1"
    # steps per reduction launch (a short run gains nothing from smaller groups: measured 59.6-61.3 us per step at 20 steps
    # with groups of 4 against 60.5 with 16 - the post-synchronisation transient of the kernel dominates, not the tail)
    GROUP = max(1, int(os.environ.get("ROBCHAR_BENCH_GROUP", cfg["group"]))) if cfg["group"] > 1 else cfg["group"]
    eps = orc.compute_dkw_error(0.05, K)                # scalar host arithmetic only
    h0 = orc.xxz_delta(N) if cfg["xxz"] else None        # static diagonal (XXZ): host constants of the Hamiltonian

    # ---- inputs, resident in HBM before the timed region ------------------------------------------------------------
    if cfg["scaling"] == "weak":
        C = cfg["C"]                                    # per rank
        ctrl_np = make_controllers(config_id, C, N, rank)
        draws_np = [legacy_draws(12345 + rank + 7919 * t, C, K, N) for t in range(cfg["rotate"])]
        draws = [torch.from_numpy(d).to(dev) for d in draws_np]
        evals_per_step = world * C * K
        draw_note = (f"legacy numpy RandomState streams (seeds 12345+rank+7919 t), {cfg['rotate']} distinct tensors "
                     f"({cfg['rotate'] * C * K * N * 24 / 1e6:.0f} MB) rotated step by step, resident in HBM")
    else:
        from importlib import import_module
        part = import_module("code-robchar_amd.sharding").controller_partition(cfg["C"], world)
        lo, hi = part[rank]
        C = hi - lo                                     # this rank's controllers
        ctrl_all = make_controllers(config_id, cfg["C"], N)
        ctrl_np = ctrl_all[lo:hi]
        per_ctrl = K * N * 3
        evals_per_step = cfg["C"] * K
        if cfg["draws"] == "philox_fused":
            draws, draws_np = [None], None
            draw_note = (f"counter-based draws (Philox4x32-10 + Box-Muller, stream 20220714+{config_id}, rank slice by element "
                         f"offset) generated INSIDE the fidelity kernel in every step: no draw tensor exists")
        elif cfg["draws"] == "philox":
            draws = [be.philox_normal((C, K, N, 3), seed=20220714 + config_id, scale=SIGMA, offset=lo * per_ctrl,
                                      device=dev, as_torch=True)]
            draws_np = None
            draw_note = (f"counter-based device draws (Philox4x32-10 + Box-Muller, stream 20220714+{config_id}, rank slice "
                         f"by element offset), {C * per_ctrl * 8 / 1e9:.2f} GB per rank, generated once, resident in HBM")
        else:                                           # the same legacy streams on every rank, each keeps its slice
            draws_np = [legacy_draws(12345 + 7919 * t, cfg["C"], K, N)[lo:hi] for t in range(cfg["rotate"])]
            draws = [torch.from_numpy(np.ascontiguousarray(d)).to(dev) for d in draws_np]
            draw_note = (f"legacy numpy RandomState streams (seeds 12345+7919 t), {cfg['rotate']} distinct tensors rotated "
                         f"step by step, this rank's controller slice ({cfg['rotate'] * C * per_ctrl * 8 / 1e6:.0f} MB) "
                         f"resident in HBM")
    Cmax = -(-cfg["C"] // world) if cfg["scaling"] == "strong" else C
    ctrl = torch.from_numpy(np.ascontiguousarray(ctrl_np)).to(dev)
    fused = cfg["draws"] == "philox_fused"

    def launch(d, out):
        """one fidelity launch of this rank's C x K samples into `out`"""
        if fused:
            if C:
                be.mc_fidelity_philox(ctrl, K, N, a, b, 20220714 + config_id, offset=lo * per_ctrl, sigma=SIGMA, h0_diag=h0, out=out,
                                      kernel=kernel)
        else:
            be.mc_fidelity(ctrl, d, N, a, b, h0_diag=h0, out=out, kernel=kernel)
    NBLK = 2
    GC = GROUP * C
    fid_blk = [torch.zeros((GC, K), dtype=torch.float64, device=dev) for _ in range(NBLK)]
    packed_buf, gather_buf, fid_gather_buf = {}, {}, {}

    def buffers(ngrp, blk):
        """metric rows of `ngrp` steps' controller rows packed as rim1[3] std[3] min[3] q[3][2] -> (15, ngrp x C) (+ gather
        target).  Strong scaling: every rank's table is padded to the LARGEST shard (all_gather_into_tensor needs equal
        shapes), ngrp x Cmax columns of which this rank fills the first ngrp x C; a rank without controllers (world > C)
        still owns a table of zeros and takes part in every collective."""
        key = (ngrp, blk)
        if key not in packed_buf:
            pad = ngrp * (C if cfg["scaling"] == "weak" else Cmax)
            packed_buf[key] = torch.zeros((15, pad), dtype=torch.float64, device=dev)
            if env.collective:
                gather_buf[key] = torch.empty((world * 15, pad), dtype=torch.float64, device=dev)
                if gather_fid:
                    fid_gather_buf[key] = torch.empty((world * pad, K), dtype=torch.float64, device=dev)
        return packed_buf[key]

    for blk in range(NBLK):                             # allocate outside the timed region
        for ngrp in {GROUP, warmup % GROUP, steps % GROUP} - {0}:
            buffers(ngrp, blk)
    main_stream = torch.cuda.current_stream(dev)
    # HIP events on the launch stream around EVERY timed fidelity launch, in brackets of BRK consecutive launches (between
    # two markers the stream carries those launches and, at group boundaries, event records / waits - no other kernel);
    # kernel time = sum of the brackets / launches.  Bracketing single launches of the 53 us kernel perturbs them (the two
    # markers add ~5 us), so that is done only when a step IS one long launch (config 4).
    BRK = 16 if C * K <= 4_000_000 else 1
    n_brk = (steps + BRK - 1) // BRK
    k_start = [torch.cuda.Event(enable_timing=True) for _ in range(n_brk)]
    k_stop = [torch.cuda.Event(enable_timing=True) for _ in range(n_brk)]
    last = {}
    # The reduction stream runs at the launch stream's priority (round 4; rounds 1-3: high priority, -1).  At high priority
    # a group's reduction pre-empts the fidelity launches it overlaps - 4 of the 20 launches of the driver's window; same-box A/B
    # (profiles/r04_ab_side_priority.txt): 20-step window, kernel 57.9 -> 55.4 us and 62.0 -> 59.9 us per step (means of three
    # runs each), 4 000-step run 53.05 -> 52.58 us and 54.21 -> 53.76 us per step.  ROBCHAR_BENCH_SIDE_PRIO=-1: the old setting.
    side_stream = torch.cuda.Stream(dev, priority=int(os.environ.get("ROBCHAR_BENCH_SIDE_PRIO", "0")))
    blk_done = [torch.cuda.Event() for _ in range(NBLK)]
    side_done = [torch.cuda.Event() for _ in range(NBLK)]
    # setup, like the allocations above: the side stream's FIRST use creates its hardware queue (9 ms of host time).  With
    # --warmup 0 (or fewer warm-up steps than a group) that landed at the first group boundary of the TIMED region
    # (--steps 32 --warmup 0: 228 us per step instead of 58).  One trivial operation here; no step of the path is run.
    with torch.cuda.stream(side_stream):
        torch.zeros(8, dtype=torch.float64, device=dev).add_(1.0)
    side_stream.synchronize()

    def step(i, timed_idx=None, final=False, total=None):
        """i counts from 0 within the current phase (warm-up / timed) of `total` steps; a phase ends with a flush and a fence."""
        g, blk = i % GROUP, (i // GROUP) % NBLK
        if g == 0 and i >= NBLK * GROUP:
            main_stream.wait_event(side_done[blk])         # block `blk` has been reduced (and gathered): free again
        if timed_idx is not None and timed_idx % BRK == 0:
            k_start[timed_idx // BRK].record(main_stream)
        d = draws[i % len(draws)]
        launch(d, fid_blk[blk][g * C:(g + 1) * C])
        if timed_idx is not None and (timed_idx % BRK == BRK - 1 or timed_idx == steps - 1):
            k_stop[timed_idx // BRK].record(main_stream)
        last.update(g=g, blk=blk, draws=i % len(draws))
        if not (g == GROUP - 1 or final):
            return
        # A finished group is reduced on the SIDE stream, beside the next group's launches - when there IS a next full group to
        # hide behind (remaining launches >= GROUP): the latency-bound reduction route fills issue slots they leave idle
        # (rc_reduce_ex_f64_async, flags 0).  In the TAIL of a timed phase there is not (the driver's 20-step window is all
        # tail: its 16-group would be reduced beside the last four launches, stretching them and finishing after them): such a
        # group waits and is reduced with the phase's last one, in order on the launch stream, through the dense standalone
        # route (RC_REDUCE_STANDALONE: 2x faster alone) - a cross-queue dependency alone takes 30-40 us to resolve on this stack.
        # (timed phase only: the warm-up's last group stays on the side stream, whose first use creates its hardware queue -
        # 9 ms of host time that must not land in the timed region)
        remaining = (total - (i + 1)) if total is not None else GROUP
        if timed_idx is not None and not final and remaining < GROUP:
            pending.append((blk, g))
            return
        in_order = final and timed_idx is not None
        todo = (pending + [(blk, g)]) if in_order else [(blk, g)]
        del pending[:]
        for blk_r, g_r in todo:
            reduce_group(blk_r, g_r, in_order, overlapped=(not in_order) and remaining >= GROUP)

    pending = []

    def reduce_group(blk, g, in_order, overlapped):
        rows = (g + 1) * C                               # a final partial group reduces only the slabs it filled
        red_stream = main_stream if in_order else side_stream
        if not in_order:
            blk_done[blk].record(main_stream)
        with torch.cuda.stream(red_stream):
            if not in_order:
                side_stream.wait_event(blk_done[blk])
            pk = buffers(g + 1, blk)
            view = pk if pk.shape[1] == rows else pk[:, :rows]
            if rows == 0:                                # a rank without controllers (world > C): nothing to reduce
                red = be.packed_views(view)
            elif view.is_contiguous():
                red = be.reduce_metrics(fid_blk[blk][:rows], dkw_eps=eps, out=be.packed_views(view), want_sorted=with_cdf,
                                        overlapped=overlapped)
            else:                                        # ragged strong-scaling shard: reduce, then place in the padded rows
                tmp = be.reduce_packed(fid_blk[blk][:rows], eps, overlapped=overlapped)
                view.copy_(tmp)
                red = be.packed_views(tmp)
            last.update(red=red, rows=rows, packed=pk)
            if env.collective:
                env.all_gather(gather_buf[(g + 1, blk)], pk)
                if gather_fid:
                    env.all_gather(fid_gather_buf[(g + 1, blk)], fid_blk[blk][:rows] if pk.shape[1] == rows else
                                   torch.nn.functional.pad(fid_blk[blk][:rows], (0, 0, 0, pk.shape[1] - rows)))
                    last.update(fid_gathered=fid_gather_buf[(g + 1, blk)])
                last.update(gathered=gather_buf[(g + 1, blk)])
            side_done[blk].record(red_stream)

    # clock pre-roll (untimed, not counted as warm-up steps): after an idle period the chip's power management needs
    # ~30 ms of CONTINUOUS load to settle (scripts/time_profile.py: 56 -> 72 -> 58 us over the first 10 ms, 52-53 us from
    # ~30 ms on), and every synchronisation gap inside that window restarts part of the transient; the driver's 20-step
    # run (1.2 ms) would otherwise measure the transient, not the kernel.  So: a short calibration burst, then ONE
    # uninterrupted stream of launches worth `preroll_s` seconds, and the warm-up steps follow without a gap.
    n_pre = 0
    pre_tail = None
    if preroll_s is None:
        preroll_s = float(os.environ.get("ROBCHAR_BENCH_PREROLL_S", "0.1"))
    if preroll_s > 0:
        n_cal = 64 if C * K <= 4_000_000 else 4          # (config 4's launches take 5 ms each)
        t_cal = time.perf_counter()
        for _ in range(n_cal):
            launch(draws[n_pre % len(draws)], fid_blk[0][:C])
            n_pre += 1
        torch.cuda.synchronize(dev)
        per_launch = max((time.perf_counter() - t_cal) / n_cal, 1e-6)
        n_roll = min(20000, int(preroll_s / per_launch))
        n_tail = min(256, n_roll // 2)                   # the steady-state figure: the LAST launches of the pre-roll stream
        pre_e0, pre_e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for j in range(n_roll):
            if n_tail and j == n_roll - n_tail:
                pre_e0.record(main_stream)
            launch(draws[n_pre % len(draws)], fid_blk[0][:C])
            n_pre += 1
        if n_tail:
            pre_e1.record(main_stream)
            pre_tail = (pre_e0, pre_e1, n_tail)
    for i in range(warmup):
        step(i, final=(i == warmup - 1), total=warmup)
    env.fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, timed_idx=i, final=(i == steps - 1), total=steps)      # the last step flushes its (possibly partial) group
    env.fence()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)

    brk_ms = [float(k_start[j].elapsed_time(k_stop[j])) for j in range(n_brk)]
    kern_ms_mean = float(sum(brk_ms) / steps)
    # per bracket, per launch (only worth printing for a short run: the driver's 20-step window = one bracket of 16 + one of 4)
    brk_per_launch = [round(brk_ms[j] / min(BRK, steps - j * BRK), 6) for j in range(n_brk)] if n_brk <= 8 else None
    steady = None
    if pre_tail is not None:
        steady = {"kernel_ms": pre_tail[0].elapsed_time(pre_tail[1]) / pre_tail[2], "launches": pre_tail[2],
                  "note": "untimed: the last launches of the uninterrupted clock pre-roll stream (HIP events on the launch "
                          "stream) - what the kernel takes once the power management has settled; the timed region of a "
                          "short run starts behind a synchronisation gap and reads 4-8 % above it"}

    # ---- correctness of what was timed: subsample against the oracle, RIM against the tensor mean ------------------
    g, blk = last["g"], last["blk"]
    rows = slice(g * C, (g + 1) * C)
    f_host = fid_blk[blk][rows].cpu().numpy()
    err = rim_err = 0.0
    teeth_fields = {}
    if C:                                                # (a rank without controllers has nothing to check)
        nsub = min(8, C)
        sel = np.arange(0, K, max(1, K // 11))
        if draws_np is not None:
            sub = draws_np[last["draws"]][:nsub][:, sel]
        elif fused:                                      # the first controllers' draws, regenerated by the generator kernel
            sub = be.philox_normal((nsub, K, N, 3), seed=20220714 + config_id, scale=SIGMA, offset=lo * per_ctrl, device=dev,
                                   as_torch=True)[:, torch.from_numpy(sel).to(dev)].cpu().numpy()
        else:
            sub = draws[0][:nsub][:, torch.from_numpy(sel).to(dev)].cpu().numpy()
        ref = orc.fidelity_eigh(ctrl_np[:nsub], sub, N, a, b, h0_diag=h0)
        err = float(np.abs(f_host[:nsub][:, sel] - ref).max())
        rim_err = float(np.abs(last["red"]["rim1"][0][rows].cpu().numpy() - (1 - f_host).mean(axis=1)).max())
        teeth_fields = teeth(f_host, f_host[:nsub][:, sel], ref)
    ok = True
    if env.collective:
        gathered, pk = last["gathered"], last["packed"]
        ok = bool(torch.equal(gathered.view(world, 15, -1)[rank], pk))
        chk = torch.nan_to_num(gathered).sum().reshape(1).clone()          # every rank must hold the same full table
        lo_t, hi_t = chk.clone(), chk.clone()
        if env.backend != "nccl":
            lo_t, hi_t = lo_t.cpu(), hi_t.cpu()
        env.dist.all_reduce(lo_t, op=env.dist.ReduceOp.MIN)
        env.dist.all_reduce(hi_t, op=env.dist.ReduceOp.MAX)
        ok = ok and bool((lo_t == hi_t).all())
        if gather_fid:                                   # the reassembled fidelity vectors: own slab intact, same everywhere
            fg = last["fid_gathered"].view(world, -1, K)
            nrow = last["rows"]
            ok_f = bool(torch.equal(fg[rank, :nrow], fid_blk[last["blk"]][:nrow]))
            s_f = torch.nan_to_num(fg[:, :nrow] if cfg["scaling"] == "weak" else fg).sum().reshape(1).clone()
            lo_f, hi_f = s_f.clone(), s_f.clone()
            if env.backend != "nccl":
                lo_f, hi_f = lo_f.cpu(), hi_f.cpu()
            env.dist.all_reduce(lo_f, op=env.dist.ReduceOp.MIN)
            env.dist.all_reduce(hi_f, op=env.dist.ReduceOp.MAX)
            ok = ok and ok_f and bool((lo_f == hi_f).all())
    # The metric table of the LAST timed step as every rank holds it after the exchange, padding dropped, controllers in
    # global order: (15, controllers of all ranks).  Under strong scaling it is the same table whatever the world size is
    # (same controllers, same draws by element, per-controller reductions in a fixed order) - bit for bit, so its hash is
    # what the ragged multi-rank rehearsals compare with the one-rank run (tests/test_gpu_bench.py).
    import hashlib
    if env.collective:
        gt = last["gathered"].view(world, 15, -1)
        widths = [C] * world if cfg["scaling"] == "weak" else [h - l for l, h in part]
        table = torch.cat([gt[r][:, g * w:(g + 1) * w] for r, w in enumerate(widths)], dim=1)
    else:
        table = last["packed"][:, g * C:(g + 1) * C]
    table_np = np.ascontiguousarray(table.cpu().numpy())
    check = {"max_abs_err_vs_oracle": err, **teeth_fields, "rim_err": rim_err, "gather_ok": ok,
             "metric_table_sha256": hashlib.sha256(table_np.tobytes()).hexdigest()[:16],
             "metric_table_shape": list(table_np.shape), "metric_table_finite": bool(np.isfinite(table_np).all())}

    bytes_per_eval = 8 if fused else 24 * N + 8         # ("in philox mode draws are not read: 8 B/eval")
    evals_per_launch = C * K
    achieved = bytes_per_eval * evals_per_launch / (kern_ms_mean * 1e-3) / 1e9 if C else 0.0
    mode = f"ends (<{N}, 2>)" if {a, b} == {0, N - 1} else f"adjugate (<{N}, 1>)"
    fields = {
        "value": evals_per_step * steps / elapsed, "unit": "evals/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": cfg["scaling"],
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["label"], "draws": draw_note,
                   "step": f"fidelity kernel + per-controller RIM/std/min/Q reductions (one reduction launch per {GROUP} "
                           f"step{'s' if GROUP > 1 else ''})"
                           + (" + row sort (exact ECDF)" if with_cdf else "")
                           + (" + RCCL all-gather of the per-controller metric rows (side stream, overlapped)" if env.collective else "")
                           + (" + all-gather of the raw fidelity slabs" if (env.collective and gather_fid) else ""),
                   "kernel": kernel, "parallelism": f"controller-sharded x{world}",
                   "precision": "fp64 in, fp64 out, results checked against the fp64 oracle (1e-10; measured ~1e-16); inside "
                                "the kernel the eigenvalue starting values come from fp32 QL rotations and are finished by an "
                                "fp64 Ehrlich-Aberth / Halley step on the characteristic polynomial (DESIGN.md 3)",
                   "evals_per_step": evals_per_step, "clock_preroll_launches_untimed": n_pre,
                   "collective": ("none" if not env.collective else ("rccl all_gather_into_tensor" if env.backend == "nccl"
                                                                      else f"{env.backend} (rehearsal, host hop)"))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "kernel": (f"mc_fid_chain_{'philox_' if fused else ''}kernel, weight mode {mode}" if kernel in ("auto", "tridiag_adj")
                                else kernel),
                     "kernel_ms": kern_ms_mean, "kernel_launches_timed": steps,
                     "kernel_ms_method": f"HIP events on the launch stream around every timed launch, in brackets of {BRK} "
                                         f"consecutive launch{'es' if BRK > 1 else ''}: sum of the brackets / {steps}",
                     "bytes_per_eval": bytes_per_eval, "evals_per_launch": evals_per_launch,
                     "kernel_ms_per_bracket": brk_per_launch, "steady_state_untimed": steady},
        "check": check,
    }
    return fields, (f_host, last, ctrl_np, draws_np)


def static_profile_fields(kern_ms_mean):
    """`roofline.traffic` and the fp64 instruction accounting come from committed PMC passes (separate rocprofv3 runs,
    profiles/traffic.json) - constants of the build, NOT measured in this run; labelled as such."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None, None
    try:
        prof = json.load(open(tpath))
    except Exception:
        return None, None, None
    src = "profiles/traffic.json (static: rocprofv3 --pmc passes committed with the build, not measured in this run)"
    traffic = prof.get("hbm_bytes_per_launch")
    valu, flop = prof.get("valu_insts_per_launch"), prof.get("fp64_flop_per_launch")
    fp64 = None
    if valu:
        # the binding roof: fp64 VALU issue.  One wave-instruction occupies a SIMD for 4 cycles;
        # 1024 SIMDs x 2.4 GHz / 4 = 614 G wave-instructions/s at the nominal clock.
        rate = valu / (kern_ms_mean * 1e-3) / 1e9
        fp64 = {"source": src, "valu_wave_insts_per_launch": valu, "achieved": rate, "peak": 614.4,
                "unit": "G wave-inst/s", "frac": rate / 614.4,
                "note": "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction; the instruction stream is ~35 % "
                        "fp64 (Aberth step, weights, sincos), ~50 % fp32 (QL rotations, pair sums), the rest moves / converts / "
                        "compares; the socket sits at its 1.4 kW power cap, which holds the clock below nominal"}
        if flop:
            tf = flop / (kern_ms_mean * 1e-3) / 1e12
            fp64.update({"fp64_flop_per_launch": flop, "achieved_tflops": tf, "peak_tflops": 78.6, "frac_tflops": tf / 78.6})
        if prof.get("fp32_flop_per_launch"):
            fp64.update({"fp32_flop_per_launch": prof["fp32_flop_per_launch"],
                         "achieved_tflops_fp32": prof["fp32_flop_per_launch"] / (kern_ms_mean * 1e-3) / 1e12})
        # The binding roof as DATA, so that "VALU-issue-bound" can be recomputed from the line: the kernel's instruction mix
        # (PMC, per 64-sample wave) priced with the measured issue time of each instruction class (scripts/ubench/pk_issue,
        # profiles/r03_ubench_ql32_pk.txt: wall ns per wave-instruction per SIMD with 4 resident waves, an otherwise idle
        # chip at the clock IT gets) -> the time the launch would take if every SIMD issued back to back at that rate.
        m64, m32 = prof.get("fp64_mix_wave_insts_per_launch"), prof.get("fp32_mix_wave_insts_per_launch")
        waves = prof.get("waves_per_launch")
        if m64 and m32 and waves:
            ns = {"fp64": 1.964, "fp64_trans": 7.9, "fp32": 1.2, "fp32_trans": 2.4, "other": 1.1}
            n64 = m64["add"] + m64["mul"] + m64["fma"]
            n32 = m32["add"] + m32["mul"] + m32["fma"]
            other = valu - n64 - n32 - m64["trans"] - m32["trans"]
            floor_ns = (n64 * ns["fp64"] + m64["trans"] * ns["fp64_trans"] + n32 * ns["fp32"] + m32["trans"] * ns["fp32_trans"]
                        + other * ns["other"]) / 1024.0
            fp64["valu_issue"] = {
                "insts_per_wave": round(valu / waves, 1), "waves": int(waves), "simds": 1024,
                "mix_per_wave": {"fp64": round(n64 / waves, 1), "fp64_trans": round(m64["trans"] / waves, 1),
                                 "fp32": round(n32 / waves, 1), "fp32_trans": round(m32["trans"] / waves, 1),
                                 "other": round(other / waves, 1)},
                "ns_per_inst": ns, "ns_per_inst_source": "scripts/ubench/pk_issue (profiles/r03_ubench_ql32_pk.txt): v_fma_f64 1.964, "
                "v_fma_f32 1.30 / v_mul_f32 1.11, v_rsq_f64 ~4 issue slots, v_rsq_f32 / v_rcp_f32 ~2; 4 waves per SIMD",
                "implied_floor_us": round(floor_ns * 1e-3, 2), "measured_us": round(kern_ms_mean * 1e3, 2),
                "measured_over_floor": round(kern_ms_mean * 1e6 / floor_ns, 3),
                "note": "floor = every SIMD issuing its share of the launch's VALU instructions back to back at the micro-benchmark's "
                        "rate; the kernel sits ~1.5x above it: its dependent chains (QL chase, recurrences) and the clock the "
                        "1.4 kW power cap allows under dense fp64 (1.45-1.5 GHz against the micro-benchmark's ~2 GHz)"}
    return traffic, src, fp64


def end_to_end(env, be, full: bool):
    """The PRODUCT API timed end to end (`MCDataSim.get_metrics_dict()` on a cold cache: draws, fidelity kernels,
    reductions, D2H, cache files), next to the kernel-only headline.  Legs:
      paper_*      the paper's scale (mcsim.py:202-210): 4 algorithms x 11 sigma levels x 1000 controllers x 100 draws, N=5
      c4_level_api BASELINE config 4 through the API: one sigma level, 1000 x 100 000, device draws, metrics only
      arim_scan_legacy  `MCDataSim.get_arims` at the paper's size (40 checkpoints x 100 controllers x 11 levels x 100 draws),
                   the reference's NumPy stream in the reference's nested order (N = 1 only)
    """
    torch = env.torch
    mcmod = importlib.import_module("code-robchar_amd.mc_data_sim")
    out = {}
    tmp = tempfile.mkdtemp(prefix="robchar_bench_") if env.rank == 0 else None
    if env.collective:
        box = [tmp]
        env.dist.broadcast_object_list(box, src=0)
        tmp = box[0]
    cwd = os.getcwd()
    try:
        os.chdir(tmp)
        if env.rank == 0:
            os.makedirs("experiments", exist_ok=True)

        def write_controllers(exp, N, out_spin, algos, C):
            rng = np.random.default_rng(5)
            le = {}
            for a in algos:
                x = np.empty((C, N + 1))
                x[:, :N] = rng.uniform(-10, 10, (C, N))
                x[:, N] = rng.uniform(2, 30, C)
                le[a] = {("%d" % N if a == "lbfgs" else "0.05"): {"controller": x.tolist()}}
            if env.rank == 0:
                os.makedirs(f"experiments/{exp}", exist_ok=True)
                json.dump(le, open(f"experiments/{exp}/ppo_spin_{N}_0-{out_spin}_c_{C}", "w"))
            if env.collective:
                env.dist.barrier()

        def timed(exp, N, out_spin, algos, C, K, noises, **kw):
            write_controllers(exp, N, out_spin, algos, C)
            np.random.seed(1)
            sim = mcmod.MCDataSim(experiment_name=exp, Nspin=N, inspin=0, outspin=out_spin, noises=noises, bootreps=K,
                                  training_noise=0.05, numcontrollers=C, verbose=False, **kw)
            torch.cuda.synchronize(env.dev)
            t0 = time.perf_counter()
            met = sim.get_metrics_dict()
            torch.cuda.synchronize(env.dev)
            wall = env.max_over_ranks(time.perf_counter() - t0)
            evals = len(algos) * len(noises) * C * K
            rim = np.array(met[algos[0]][r'$W(.,\delta(x-1))$'], dtype=float)
            assert rim.shape == (len(noises), C) and np.isfinite(rim).all() and (rim >= -1e-12).all() and (rim <= 1).all()
            size = sum(os.path.getsize(os.path.join(f"experiments/{exp}", f)) for f in os.listdir(f"experiments/{exp}")
                       if ".mc" in f) if env.rank == 0 else 0
            return {"wall_s": round(wall, 5), "evals": evals, "evals_per_s": float(f"{evals / wall:.4g}"),
                    "cache_MB": round(size / 1e6, 1)}

        paper = dict(N=5, out_spin=2, algos=["ppo", "snob", "nmplus", "lbfgs"], C=1000, K=100, noises=np.linspace(0, 0.1, 11))
        # first call of anything pays one-off costs (module import, hipModule load, allocator warm-up): burn a tiny run
        timed("warm", 5, 2, ["ppo"], 64, 10, np.linspace(0, 0.1, 3), rng_mode="philox", cache_format="none")
        importlib.import_module("code-robchar_amd.cache_io").warm_up()     # (the tiny run's leaves are too small to start the encoder threads)
        out["paper_philox_metrics_only"] = timed("p1", rng_mode="philox", seed=7, cache_format="none", **paper)
        out["paper_philox_json_cache"] = timed("p2", rng_mode="philox", seed=7, cache_format="json", **paper)
        if full:
            out["paper_philox_npy_cache"] = timed("p3", rng_mode="philox", seed=7, cache_format="npy", **paper)
            out["paper_legacy_json_cache"] = timed("p4", rng_mode="legacy", cache_format="json", **paper)
        out["c4_level_api"] = timed("c4", 7, 3, ["ppo"], 1000, 100000, np.array([0.05]), rng_mode="philox", seed=11,
                                    cache_format="none")
        if full:
            # the ARIM scan of gen_fig_8_arim_fcall_scaling.py:37-69 at the paper's size: 40 function-call checkpoints x
            # 100 controllers x 11 sigma levels x 100 draws for one algorithm, the reference's stream and nested order
            rng = np.random.default_rng(8)
            def ckpt():
                x = np.empty((100, 6))
                x[:, :5] = rng.uniform(-10, 10, (100, 5))
                x[:, 5] = rng.uniform(2, 30, 100)
                return x.tolist()
            cdict = {"lbfgs": {"0.01": {str(k * 10 ** 6): ckpt() for k in range(40)}}}
            os.makedirs("experiments/arim", exist_ok=True)
            sim = mcmod.MCDataSim(experiment_name="arim", Nspin=5, inspin=0, outspin=2, bootreps=100, numcontrollers=100,
                                  verbose=False)
            np.random.seed(2)
            torch.cuda.synchronize(env.dev)
            t0 = time.perf_counter()
            arims, keys = sim.get_arims("lbfgs", nlvl="0.01", marker="bench", cdict=cdict)
            torch.cuda.synchronize(env.dev)
            wall = time.perf_counter() - t0
            assert arims.shape == (40, 11) and np.isfinite(arims).all() and len(keys) == 40
            out["arim_scan_legacy"] = {"wall_s": round(wall, 5), "evals": 40 * 100 * 11 * 100,
                                       "evals_per_s": float(f"{40 * 100 * 11 * 100 / wall:.4g}"), "cache_MB": 0.0}
        out["note"] = ("cold MCDataSim.get_metrics_dict(); paper = 4 algos x 11 levels x 1000 ctrls x 100 draws, N=5; "
                       "legacy = the reference's numpy stream (on the GPU), philox = counter-based draws; c4 = 1000 x 1e5")
    finally:
        os.chdir(cwd)
        if env.collective:
            env.dist.barrier()
        if env.rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


EXIT_PARITY = 1            # the headline (or an appended leg's) parity check failed
EXIT_EXTRAS = 3            # headline fine and printed, but an appended leg raised, or the watchdog had to end the run


def launch_ranks(n: int) -> int:
    """`python3 bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) as CHILD processes with the
    environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and
    relay rank 0's ONE line.  Nothing in this parent has touched the GPU (no torch import, no HIP call) and nothing is
    exec'ed.  Exit code: rank 0's if non-zero, else the first non-zero one; a rank that dies early takes the others
    down after a grace period instead of leaving them in a collective."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                ROBCHAR_BENCH_LAUNCHER="self")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True, bufsize=1))
    got = []

    def pump():                                          # rank 0's stdout: the line is kept, anything else goes to stderr
        for out in procs[0].stdout:
            if out.startswith('{"metric"') and not got:
                got.append(out)
            else:
                sys.stderr.write(out)

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    grace = float(os.environ.get("ROBCHAR_BENCH_RANK_GRACE_S", "60"))
    first_bad = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and first_bad is None:
            first_bad = time.monotonic()
        if first_bad is not None and time.monotonic() - first_bad > grace:
            for p in procs:
                if p.poll() is None:
                    p.kill()                             # exactly the PIDs started here
    th.join(timeout=10)
    codes = [p.returncode for p in procs]
    if got:
        sys.stdout.write(got[0])
        sys.stdout.flush()
    rc = codes[0] if codes[0] != 0 else next((c for c in codes if c != 0), 0)
    if rc == 0 and not got:
        rc = 1
    if rc != 0:
        sys.stderr.write(f"bench.py launcher: rank exit codes {codes}\n")
    return rc if rc >= 0 else 128 - rc                   # (a rank ended by a signal has a negative code)


def cold_kernel_ms(env, be, cfg, ctrl_np, draws_np, kernel, h0, n=20, idle_s=1.0):
    """The figure a COLD 20-launch run gives (no clock pre-roll): idle the chip, then bracket `n` back-to-back launches
    with HIP events on the launch stream.  Sits beside the steady-state `roofline.kernel_ms`."""
    torch = env.torch
    ctrl = torch.from_numpy(np.ascontiguousarray(ctrl_np)).to(env.dev)
    d = torch.from_numpy(np.ascontiguousarray(draws_np)).to(env.dev)
    out = torch.empty((ctrl.shape[0], d.shape[1]), dtype=torch.float64, device=env.dev)
    torch.cuda.synchronize(env.dev)
    time.sleep(idle_s)
    st = torch.cuda.current_stream(env.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        be.mc_fidelity(ctrl, d, cfg["N"], cfg["inspin"], cfg["outspin"], h0_diag=h0, out=out, kernel=kernel)
    e1.record(st)
    torch.cuda.synchronize(env.dev)
    return e0.elapsed_time(e1) / n


def delocalised_controllers(config_id):
    """(N, in, out, 100 controllers, h0_diag, description) of the DELOCALISED workload of a BASELINE GPU configuration - the
    reference's own shipped L-BFGS controllers where it ships any (N = 5: first 100 `lbfgs` rows of
    experiments/pipeline_nmplus2/ppo_spin_5_0-4_c_1000.le; N = 7: noisy_analysis/lbfgs_spin_7_0-{6,3}_in, 57 / 100 rows), for
    N = 10 XXZ controllers constructed by tests/golden/make_golden.py against the reference's noiseless fidelity (it ships
    none).  Fixtures: tests/golden/highfid.npz, lbfgs_n7.npz.  None when the fixtures are absent."""
    g = os.path.join(ROOT, "tests", "golden")
    if not (os.path.exists(os.path.join(g, "highfid.npz")) and os.path.exists(os.path.join(g, "lbfgs_n7.npz"))):
        return None
    z, l7 = np.load(os.path.join(g, "highfid.npz")), np.load(os.path.join(g, "lbfgs_n7.npz"))
    if config_id == 2:
        N, a, b, rows, h0, what = 5, 0, 4, z["c2_ctrl"], None, "the reference's first 100 shipped N=5 0->4 L-BFGS controllers"
    elif config_id in (3, 30):
        N, a, b, rows, h0, what = 7, 0, 6, l7["ctrl_0-6"], None, "the reference's 57 shipped N=7 0->6 L-BFGS controllers tiled to 100"
    elif config_id in (4, 40):
        N, a, b, rows, h0, what = 7, 0, 3, l7["ctrl_0-3"], None, "the reference's 100 shipped N=7 0->3 L-BFGS controllers"
    elif config_id == 5:
        N, a, b, rows, h0, what = 10, 0, 9, z["c5_ctrl"], np.ascontiguousarray(z["c5_h0_diag"]), \
            "100 constructed N=10 XXZ 0->9 controllers (mirror-symmetric starts improved on the reference's noiseless fidelity)"
    else:
        return None
    return N, a, b, np.ascontiguousarray(rows[np.arange(100) % rows.shape[0]]), h0, what


def delocalised_leg(env, be, orc, kernel, config_id, launches=200):
    """The TIMED kernel on fidelities of O(1) (round 5).  SURVEY.md 8(d)'s uniform random controllers are Anderson-localised
    (median fidelity 2.5e-7 at N = 7, 3.9e-10 at N = 10 XXZ): an absolute 1e-10 bound on them is a loose relative one.  This leg
    runs the configuration's shape - 100 controllers x 10 000 perturbations at sigma = 0.05 - on its delocalised controller set
    (`delocalised_controllers`): kernel time by HIP events on the launch stream over `launches` back-to-back launches (the chip
    is still warm from the headline), a 2 % subsample of the fidelities against the oracle with absolute AND relative bounds,
    the workload's median fidelity / share above 1e-3, the share of tiles off the one-step path, and - where the reference
    recorded one - the noiseless fidelity against the optimiser's own `best_fid`."""
    torch = env.torch
    w = delocalised_controllers(config_id)
    if w is None:
        return {"skipped": "tests/golden/highfid.npz / lbfgs_n7.npz not present"}
    N, a, b, ctrl_np, h0, what = w
    C, K = 100, 10000
    draws_np = SIGMA * np.random.default_rng(20220714 + 30 + (0 if config_id in (3, 30) else config_id)).standard_normal((C, K, N, 3))
    ctrl = torch.from_numpy(ctrl_np).to(env.dev)
    d = torch.from_numpy(draws_np).to(env.dev)
    out = torch.empty((C, K), dtype=torch.float64, device=env.dev)
    for _ in range(5):
        be.mc_fidelity(ctrl, d, N, a, b, h0_diag=h0, out=out, kernel=kernel)
    torch.cuda.synchronize(env.dev)
    be.polish_tiles(reset=True)
    st = torch.cuda.current_stream(env.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(launches):
        be.mc_fidelity(ctrl, d, N, a, b, h0_diag=h0, out=out, kernel=kernel)
    e1.record(st)
    torch.cuda.synchronize(env.dev)
    ms = e0.elapsed_time(e1) / launches
    off = be.polish_tiles() / launches / (C * ((K + 63) // 64))
    sel = np.arange(0, K, 50)
    f_host = out.cpu().numpy()
    want = orc.fidelity_eigh(ctrl_np, draws_np[:, sel], N, a, b, h0_diag=h0)
    err = float(np.abs(f_host[:, sel] - want).max())
    res = {"workload": f"N={N} {a}->{b}{' XXZ' if h0 is not None else ''}, {what}, 100 x 10000, sigma 0.05",
           "kernel_ms": round(ms, 5), "evals_per_s": float(f"{C * K / ms * 1e3:.5g}"),
           "roofline_frac": round((24.0 * N + 8.0) * C * K / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "tiles_off_one_step_path": round(off, 4), "mean_fidelity": round(float(f_host.mean()), 6),
           "max_abs_err_vs_oracle_2pct": err, **teeth(f_host, f_host[:, sel], want)}
    if config_id in (3, 30, 4, 40):
        z = np.load(os.path.join(ROOT, "tests", "golden", "lbfgs_n7.npz"))
        key = "0-6" if config_id in (3, 30) else "0-3"
        nrow = z["ctrl_" + key].shape[0]
        noiseless = be.mc_fidelity(ctrl_np[:nrow], np.zeros((nrow, 1, N, 3)), N, a, b)
        res["max_abs_err_noiseless_vs_reference_best_fid"] = float(np.abs(np.asarray(noiseless)[:, 0] - z["best_fid_" + key]).max())
    return res


def launch_size_leg(env, be, orc, kernel, headline_kernel_ms, controllers=2000, launches=20, lead_in=40):
    """What a launch BOUNDARY costs the headline (round 5).  The benchmark's step is one launch of 1e6 evaluations = 15 700
    single-wave workgroups = 3.8 rounds of the chip's wave slots: every launch fills (the first round's 44 MB of draws arrive as
    one HBM burst before any arithmetic starts) and drains (one wave lifetime) - profiles/r05_launch_boundary.txt.  This leg runs
    the SAME kernel on config 3's shape with `controllers` x 10 000 samples per launch (SURVEY 8(d)'s controllers, counter-based
    device draws) and reports the time per 1e6 evaluations next to the headline's: the difference is the boundary."""
    torch = env.torch
    cfg = CONFIGS[3]
    N, a, b, K = cfg["N"], cfg["inspin"], cfg["outspin"], cfg["K"]
    C = controllers
    ctrl_np = make_controllers(3, C, N)
    ctrl = torch.from_numpy(ctrl_np).to(env.dev)
    d = be.philox_normal((C, K, N, 3), seed=20220714 + 33, scale=SIGMA, device=env.dev, as_torch=True)
    out = torch.empty((C, K), dtype=torch.float64, device=env.dev)
    # the legs before this one leave the chip idle (the cold-start figure sleeps for a second): `lead_in` launches = ~35 ms of
    # load bring the clocks back to where the headline's pre-roll had them, and the timed launches follow WITHOUT a gap
    st = torch.cuda.current_stream(env.dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(lead_in):
        be.mc_fidelity(ctrl, d, N, a, b, out=out, kernel=kernel)
    e0.record(st)
    for _ in range(launches):
        be.mc_fidelity(ctrl, d, N, a, b, out=out, kernel=kernel)
    e1.record(st)
    torch.cuda.synchronize(env.dev)
    ms_per_1e6 = e0.elapsed_time(e1) / launches / (C * K / 1e6)
    rows, cols = np.arange(0, C, max(1, C // 16)), np.arange(0, K, 997)
    got = out[rows][:, cols].cpu().numpy()
    want = orc.fidelity_eigh(ctrl_np[rows], d[rows][:, cols].cpu().numpy(), N, a, b)
    return {"workload": f"config 3's kernel and shape at {C} controllers x {K} samples per launch ({C * K / 1e6:.0f}e6 evaluations, "
                        f"{launches} launches back to back behind {lead_in} untimed ones)",
            "kernel_ms_per_1e6_evals": round(ms_per_1e6, 5),
            "roofline_frac": round((24.0 * N + 8.0) * 1e6 / (ms_per_1e6 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "headline_kernel_ms": round(headline_kernel_ms, 5),
            "launch_boundary_ms_per_headline_launch": round(headline_kernel_ms - ms_per_1e6, 5),
            "max_abs_err_vs_oracle": float(np.abs(got - want).max()), "compared_samples": int(got.size)}


def delocalised_ok(r):
    rel = r.get("max_rel_err_F_gt_1e-3")
    return ("skipped" in r) or (within(r.get("max_abs_err_vs_oracle_2pct", 0.0)) and (rel is None or within(rel, REL_TOL))
                                and within(r.get("max_abs_err_noiseless_vs_reference_best_fid", 0.0)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.3 s of GPU work (config 3)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=int(os.environ.get("ROBCHAR_BENCH_CONFIG", "3")), choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the appended short runs (config 4, fid-slab gather, cold)")
    ap.add_argument("--kernel", default="auto")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 40 if args.config in (4, 40) else 4000
    if args.warmup is None:
        args.warmup = 4 if args.config in (4, 40) else 400

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))                # before anything here touches the GPU
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world_env                                # under a launcher the launcher's world is the truth

    cfg = CONFIGS[args.config]
    cpu, cpu_fid, cpu_inputs = None, None, None
    if world_env == 1 and int(os.environ.get("RANK", "0")) == 0 and not args.no_cpu_baseline:
        if cfg["draws"] == "legacy":                     # configs 2, 3, 5: the whole workload (1e6 evaluations)
            ctrl0 = make_controllers(args.config, cfg["C"], cfg["N"], 0)
            d0 = legacy_draws(12345, cfg["C"], cfg["K"], cfg["N"])
        else:                                            # bounded sample of config 4: 100 controllers x 10 000 draws
            ctrl0 = make_controllers(4, cfg["C"], cfg["N"])[:100]
            d0 = SIGMA * np.random.default_rng(4).standard_normal((100, 10000, cfg["N"], 3))
        cpu, cpu_fid = cpu_baseline(cfg, ctrl0, d0)
        cpu_inputs = (ctrl0, d0)

    env = Env(args)
    be = importlib.import_module("code-robchar_amd.backend")
    orc = importlib.import_module("oracle.robchar_oracle")
    rccl = env.comm_info()

    fields, (f_host, last, ctrl_np, draws_np) = run_pipeline(env, be, orc, args.config, args.steps, args.warmup, args.kernel)
    fields["config"]["rccl"] = rccl
    libmod = importlib.import_module("code-robchar_amd._lib")
    # which build of librobchar_hip.so ran: 0 = the product build; a timing-experiment build cannot be loaded at all without
    # ROBCHAR_ALLOW_EXPERIMENT_LIB=1 (and would show here)
    fields["config"]["library"] = {"path": os.path.relpath(libmod.LIB_PATH, ROOT), "abi": int(libmod.load().rc_version()),
                                   "build_flags": libmod.build_flags(), "build_flag_names": libmod.build_flag_names(libmod.build_flags()),
                                   "legacy_normals_bit_identical_to_numpy": bool(libmod.load().rc_legacy_log_is_host_exact())}
    fields["config"]["launcher"] = ("bench.py --gpus N (self-launched child ranks)"
                                    if os.environ.get("ROBCHAR_BENCH_LAUNCHER") == "self"
                                    else ("external (torch.distributed.run)" if "WORLD_SIZE" in os.environ else "none (one process)"))
    check = fields["check"]
    if cpu_fid is not None and cfg["draws"] == "legacy":
        # the CPU baseline computed the 1e6 fidelities of draw tensor 0: compare ALL of them with the GPU's
        import torch
        got = be.mc_fidelity(torch.from_numpy(cpu_inputs[0]).to(env.dev), torch.from_numpy(cpu_inputs[1]).to(env.dev),
                             cfg["N"], cfg["inspin"], cfg["outspin"], kernel=args.kernel,
                             h0_diag=(orc.xxz_delta(cfg["N"]) if cfg["xxz"] else None)).cpu().numpy()
        check["max_abs_err_vs_cpu_baseline_all_1e6"] = float(np.abs(got - cpu_fid).max())
        check["max_abs_err_vs_oracle"] = max(check["max_abs_err_vs_oracle"], check["max_abs_err_vs_cpu_baseline_all_1e6"])
        t_all = teeth(got, got, cpu_fid)                   # relative error over ALL samples of that tensor with F > 1e-3
        check["max_rel_err_F_gt_1e-3_all_1e6"] = t_all["max_rel_err_F_gt_1e-3"]
        check["samples_F_gt_1e-3_all_1e6"] = t_all["compared_samples_F_gt_1e-3"]
        if t_all["max_rel_err_F_gt_1e-3"] is not None:
            check["max_rel_err_F_gt_1e-3"] = max(check.get("max_rel_err_F_gt_1e-3") or 0.0, t_all["max_rel_err_F_gt_1e-3"])

    import threading
    extras = {"also": {}, "e2e": None}
    state = {"leg": None, "failed": [], "emitted": False}
    emit_lock = threading.Lock()

    def emit():
        """rank 0: the ONE JSON line, from the headline fields and whatever extras exist by now (at most once)"""
        with emit_lock:
            if state["emitted"]:
                return
            state["emitted"] = True
            also, e2e = extras["also"], extras["e2e"]
            kern_ms = fields["roofline"]["kernel_ms"]
            traffic, src, fp64 = (None, None, None)
            if args.config == 3 and args.kernel == "auto":
                traffic, src, fp64 = static_profile_fields(kern_ms)
            fields["roofline"].update({
                "traffic": traffic, "traffic_source": src,
                "note": "algorithmic traffic is 176 B/eval (PMC-measured HBM traffic = 1.001x that); the kernel is bound by "
                        "VALU work (fp32 QL rotations + fp64 Aberth step / weights / sincos) at the clock the chip holds under its "
                        "1.4 kW power cap, not by HBM - `fp64_valu` is the binding roof (DESIGN.md 4)"})
            line = {"metric": {3: "MC fidelity evals/sec (N=7, 100 ctrls x 10k draws)",
                               4: "MC fidelity evals/sec (N=7, 1000 ctrls x 100k draws, strong scaling)",
                               2: "MC fidelity evals/sec (N=5, 100 ctrls x 10k draws)",
                               5: "MC fidelity evals/sec (N=10 XXZ, 100 ctrls x 10k draws, strong scaling)",
                               30: "MC fidelity evals/sec (N=7, 100 ctrls x 10k draws, strong scaling)",
                               40: "MC fidelity evals/sec (N=7, 1000 ctrls x 100k draws incl. draw generation, strong scaling)"}[args.config]}
            line.update(fields)
            # headline numbers of the extras mirrored where a condensed record of this line keeps them
            if isinstance(e2e, dict) and "error" not in e2e:
                line["config"]["end_to_end_wall_s"] = {k: v["wall_s"] for k, v in e2e.items() if isinstance(v, dict)}
            if "value" in also.get("config4_strong", {}):
                line["config"]["config4_strong_evals_per_s"] = also["config4_strong"]["value"]
            line["fp64_valu"] = fp64
            line["cpu_baseline"] = cpu
            line["check"] = line.pop("check")
            line["extras_failed"] = list(state["failed"])     # [] = every appended leg ran to its end
            if also:
                line["also"] = also
            line["end_to_end"] = e2e                        # last: the tail of the line
            print(json.dumps(line), flush=True)

    # The extras below (appended config-4 runs, cold-start figure, product-API legs) execute collectives of their own
    # under N > 1.  They must never cost the headline: if they have not finished within the deadline (a rank that died
    # or raised alone leaves the others waiting in a collective), every rank's watchdog ends its process - rank 0 after
    # printing the line with what it has - with EXIT_EXTRAS, so the hang is visible in the exit code too.
    deadline = float(os.environ.get("ROBCHAR_BENCH_EXTRAS_TIMEOUT_S", "240"))
    def parity_ok(chk):
        rel = chk.get("max_rel_err_F_gt_1e-3")
        return within(chk["max_abs_err_vs_oracle"]) and within(chk["rim_err"]) and bool(chk["gather_ok"]) \
            and bool(chk.get("metric_table_finite", True)) and (rel is None or within(rel, REL_TOL))

    headline_ok = parity_ok(check)

    def on_deadline():
        msg = f"not finished {deadline:.0f} s after the headline run (watchdog); leg in flight: {state['leg']}"
        if env.rank == 0:
            state["failed"].append(f"watchdog:{state['leg']}")
            if state["leg"] not in (None, "end_to_end"):
                extras["also"].setdefault(state["leg"], {"error": msg})
            if extras["e2e"] is None:
                extras["e2e"] = {"error": msg}
            emit()
        sys.stderr.write(f"bench.py rank {env.rank}: extras watchdog fired after {deadline:.0f} s (leg in flight: {state['leg']})\n")
        sys.stderr.flush()
        os._exit(EXIT_EXTRAS if headline_ok else EXIT_PARITY)

    dog = threading.Timer(deadline, on_deadline)
    dog.daemon = True
    dog.start()

    def compact(f4, what):
        # compact on purpose: the driver's record keeps the TAIL of the line, these extras sit there
        return {"workload": what, "value": float(f"{f4['value']:.5g}"), "unit": "evals/s", "n_gpus": f4["n_gpus"],
                "steps": f4["steps"], "warmup": f4["warmup"], "ms_per_step": round(f4["ms_per_step"], 4),
                "scaling": "strong", "kernel_ms": round(f4["roofline"]["kernel_ms"], 4),
                "roofline_frac": round(f4["roofline"]["frac"], 4), "evals_per_step": f4["config"]["evals_per_step"],
                "evals_per_launch": f4["roofline"]["evals_per_launch"], "collective": f4["config"]["collective"],
                "check": f4["check"]}

    def leg(name, fn):
        state["leg"] = name
        try:
            return fn()
        except Exception as e:                          # never lose the headline line to an appended leg
            state["failed"].append(name)
            return {"error": repr(e)}
        finally:
            state["leg"] = None

    def c4_leg(name, what, config=4, **kw):
        def run():
            f4, _ = run_pipeline(env, be, orc, config, kernel=args.kernel, preroll_s=0.02, **kw)
            if not parity_ok(f4["check"]):
                check["appended_leg_failed"] = True
            return compact(f4, what)
        extras["also"][name] = leg(name, run)

    if args.config == 3 and not args.no_also:
        c4_leg("config4_strong", "BASELINE config 4: N=7 0->3, 1000 x 100000, strong scaling; the 2.1e9 draws are generated "
               "once BEFORE the timed region (resident, as the bench contract prescribes) - a real sigma level also pays their "
               "generation: see end_to_end.c4_level_api, the product API's figure for the same workload", steps=8, warmup=2)
        c4_leg("config4_strong_fused", "BASELINE config 4 with the 2.1e9 draws of the level generated INSIDE the fidelity kernel in "
               "every step (mc_fid_chain_philox_kernel: no draw tensor; bit-identical fidelities) - the whole sigma level per step, "
               "where config4_strong's timed region starts with the draws resident; roofline_frac counts SURVEY's 8 B/eval for "
               "philox mode and only says that this path is not about memory", config=40, steps=8, warmup=2)
        if env.collective:
            # the metric's own workload under STRONG scaling: config 3's 100 x 10 000 split by controller over the ranks
            # (the headline above is weak scaling: 100 x 10 000 PER rank) - one --gpus N invocation records weak c3,
            # strong c3 and strong c4
            c4_leg("config3_strong", "BASELINE config 3 (the metric's workload), STRONG scaling: N=7 0->6, 100 x 10000 in all, "
                   "controller-sharded over the ranks", config=30, steps=160, warmup=32)
            # north_star's exchange step as written - "all-gather ... to reassemble per-controller fidelity vectors": the
            # same run with every rank's raw fidelity slab (C/N x K fp64) replicated on every rank in every step
            c4_leg("config4_strong_gather_fid", "BASELINE config 4 + all-gather of the raw fidelity slabs (every rank "
                   "ends each step with all 1000 x 100000 fidelities)", steps=4, warmup=1, gather_fid=True)
        if draws_np is not None:
            h0c = orc.xxz_delta(cfg["N"]) if cfg["xxz"] else None
            extras["also"]["cold_20_steps_kernel_ms"] = leg("cold_20_steps_kernel_ms", lambda: {
                "kernel_ms": round(cold_kernel_ms(env, be, cfg, ctrl_np, draws_np[0], args.kernel, h0c), 5),
                "note": "20 back-to-back launches after 1 s of idle, NO clock pre-roll (HIP events on the launch stream): "
                        "the power-management transient the headline's untimed pre-roll skips"})

        if env.world == 1:
            def delocalised(cid):
                def run():
                    r = delocalised_leg(env, be, orc, args.kernel, cid)
                    if not delocalised_ok(r):
                        check["appended_leg_failed"] = True     # same consequence as a parity miss of the appended config-4 run
                    return r
                return run
            # the headline shape on the reference's shipped controllers (key kept from rounds 3 / 4), then the other three GPU
            # configurations' shapes on THEIR delocalised sets: every timed kernel of BASELINE.json checked on fidelities of O(1)
            def big_launches():
                r = launch_size_leg(env, be, orc, args.kernel, fields["roofline"]["kernel_ms"])
                if not within(r["max_abs_err_vs_oracle"]):
                    check["appended_leg_failed"] = True
                return r
            extras["also"]["launch_size"] = leg("launch_size", big_launches)
            extras["also"]["shipped_lbfgs_controllers"] = leg("shipped_lbfgs_controllers", delocalised(3))
            extras["also"]["delocalised"] = {f"config{c}": leg(f"delocalised_config{c}", delocalised(c)) for c in (2, 4, 5)}
    elif not args.no_also and env.world == 1 and delocalised_controllers(args.config) is not None:
        # --config 2 / 4 / 5: the run's own shape on its delocalised controller set
        def deloc_own():
            r = delocalised_leg(env, be, orc, args.kernel, args.config)
            if not delocalised_ok(r):
                check["appended_leg_failed"] = True
            return r
        cname = {30: 3, 40: 4}.get(args.config, args.config)
        extras["also"]["delocalised"] = {f"config{cname}": leg(f"delocalised_config{cname}", deloc_own)}

    if not args.no_end_to_end:
        extras["e2e"] = leg("end_to_end", lambda: end_to_end(env, be, full=(env.world == 1)))

    dog.cancel()
    if env.rank == 0:
        emit()
    with emit_lock:                                      # a watchdog that fired meanwhile finishes its line, then exits
        pass
    if env.collective:
        env.dist.destroy_process_group()
    if not parity_ok(check) or check.get("appended_leg_failed"):
        sys.stderr.write("bench: parity check failed\n")
        sys.exit(EXIT_PARITY)
    if state["failed"]:
        sys.stderr.write(f"bench: appended legs failed: {state['failed']}\n")
        sys.exit(EXIT_EXTRAS)


if __name__ == "__main__":
    main()
