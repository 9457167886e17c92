"""`MCDataSim` - the Monte-Carlo driver and its cached-results layout, API mirror of mcsim.py:200-510.

Same constructor kwargs, attributes, method names, file names and JSON layouts as the reference so that the
reference's own cache-hit branches (mcsim.py:396-397, :504-506) and its figure scripts' data accessors interoperate:

    <exp dir>/ppo_spin_{N}_{in}-{out}_c_{C}{filemarker}                         controller file (.le)  (input)
    ..._tn{training_noise}_br_{K}_nlvl{str(noises)}.mc    {algo: [L][C][K]}     fidelity cache
    ..._tn{training_noise}_br_{K}_nlvl{str(noises)}.mcm   {algo: {metric(+" upper"|" lower"): [L][C]}}

What is different is the execution.  The reference's triple loop (noise level x controller x draw, one `expm` per
iteration, mcsim.py:424-449) becomes, per algorithm, a DEVICE-RESIDENT pipeline on one stream:

    per level:  draws (legacy stream: host -> H2D;  philox: generated on the GPU)  ->  fidelity kernel into the
                level's slab of one (L, C, K) device tensor
    once:       ONE reduction launch over all L x C rows -> (15, L x C) metric rows, D2H of those rows only

The fidelity tensor itself stays on the GPU behind a `DeviceFids` handle and crosses PCIe only if somebody looks at
it (the `.mc` writer, `np.array(fids[algo])`); `get_metrics_dict` never re-uploads what was just computed, and the
`.mc` / `.mcm` files are formatted by the native encoder (cache_io.py).  RNG consumption in "legacy" mode is identical
to the reference: one burned draw per level (the value returned by ``rng(scale=noise)``, mcsim.py:425), then 3 N draws
per sample in (controller, draw, site, slot) order, nothing for NaN-padded controllers (mcsim.py:370-374, :442-443).

With an initialised torch.distributed group (one process per GPU) the controllers of each level are sharded over the
ranks (contiguous blocks): every rank evaluates and reduces its own block on its own GPU and the ranks all-gather the
metric rows (RCCL), plus the fidelity slabs when a `.mc` cache is wanted.  philox: every rank generates exactly its
slice of the counter-based stream.  legacy, drawn on the device (the default): the reference's stream is sequential, but
re-generating it costs milliseconds, so EVERY rank runs the generator on its own GPU from the same `RandomState`
(aligned by one small broadcast before the run) and keeps the rows of its controller block - nothing moves between
ranks and every rank ends at the reference's stream position by itself.  legacy, drawn on the host
(`legacy_draws="host"`, the bit-identical mode): rank 0 alone advances the stream, scatters the slices and broadcasts
the final generator state.  Only rank 0 writes cache files.
"""
from __future__ import annotations

import json
import os
from typing import Callable, List

import numpy as np

from . import backend, cache_io, rim_metrics
from .naming import DirectoryDoesNotExistError, ExperimentNamer  # noqa: F401  (re-exported like mcsim.py:26)
from .noise import structured_perturbation
from .rim_metrics import METRIC_NAMES, compute_dkw_error


def _progress(seq):
    try:
        from tqdm import tqdm
        return tqdm(seq)
    except Exception:          # tqdm is optional here
        return seq



class DeviceFids:
    """(L, C, K) fidelity tensor of one algorithm, resident on the GPU (rows of the `nvalid` real controllers only; the
    NaN-padded controllers of mcsim.py:442-443 are added on the host).  Behaves like the reference's nested list for
    readers - `np.array(x)`, `x[j][i][k]`, `len(x)`, `x.tolist()` - and moves to the host on first such use only."""

    def __init__(self, tensor, numcontrollers: int, host: np.ndarray = None, shape=None):
        self.tensor = tensor                      # torch (L, nvalid, K) on the compute device (None: see `host`)
        self.numcontrollers = int(numcontrollers)
        self._host = host                         # (L, C, K) already on the host (single-process multi-device mode)
        self._shape = shape

    @property
    def shape(self):
        if self.tensor is None:
            return tuple(self._host.shape) if self._host is not None else tuple(self._shape)
        L, _, K = self.tensor.shape
        return (int(L), self.numcontrollers, int(K))

    def numpy(self) -> np.ndarray:
        if self._host is None and self.tensor is None:
            raise RuntimeError("the fidelities of this run were not kept (cache_format='none' with devices=...): only "
                               "the metric rows left the GPUs; use another cache_format to keep them")
        if self._host is None:
            L, C, K = self.shape
            nvalid = int(self.tensor.shape[1])
            host = np.empty((L, C, K))
            if nvalid < C:
                host[:, nvalid:] = np.nan
            if nvalid:
                got = self.tensor.cpu().numpy()
                if nvalid == C:
                    host = got
                else:
                    host[:, :nvalid] = got
            self._host = host
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, i):
        return self.numpy()[i]

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self.numpy())

    def tolist(self):
        return self.numpy().tolist()


class MCDataSim:
    "MC data generation with structured perturbations of XX-controllers, on MI355X."

    def __init__(self, experiment_name: str = "pipeline_alpha", Nspin: int = 5,
                 inspin: int = 0, outspin: int = 2,
                 noises: np.ndarray = np.linspace(0, 0.1, 11),
                 bootreps: int = 100, training_noise: float = None,
                 numcontrollers: int = 100, parallel: bool = False,
                 num_workers: int = None,
                 dkw_conflvl: float = 0.95,
                 filemarker: str = None,
                 topk: int = 100, verbose: bool = True,
                 rng_mode: str = "legacy", seed: int = 0,
                 cache_format: str = "auto", json_max_values: int = 8_000_000,
                 legacy_draws: str = "device", devices=None):
        self.global_experiments_directory = "experiments/"
        self.filemarker = filemarker
        self.experiment_name = experiment_name
        self.topk = topk
        self.args = dict(Nspin=Nspin, inspin=inspin, outspin=outspin)
        self.bootreps = bootreps
        self.alpha = 1 - dkw_conflvl
        self.training_noise = training_noise
        self.Nspin = Nspin
        self.inspin = inspin
        self.outspin = outspin
        self.noises = noises
        self.numcontrollers = numcontrollers
        self.verbose = verbose
        # "legacy": draws from numpy's global RandomState exactly as the reference consumes it (default).
        # "philox": counter-based draws generated on the GPU (rc_draws_philox_f64) - NOT the reference's stream;
        #           for sample spaces too large to draw on the host (2.1e9 draws per level in BASELINE config 4).
        if rng_mode not in ("legacy", "philox"):
            raise ValueError("rng_mode must be 'legacy' or 'philox'")
        self.rng_mode = rng_mode
        self.seed = int(seed)
        self._philox_offset = 0
        # How the fidelity tensors are cached (extension; the default keeps the reference's files):
        #   "json"  the reference's `.mc` JSON, whatever the size        "npy"  `.npy` sidecars + a small JSON index
        #   "auto"  JSON up to `json_max_values` values per file (the paper's 4 x 11 x 1000 x 100 = 4.4e6 fit), npy above
        #   "none"  metrics only: no `.mc` is written and the tensors never leave the GPU unless the caller reads them
        if cache_format not in ("auto", "json", "npy", "none"):
            raise ValueError("cache_format must be 'auto', 'json', 'npy' or 'none'")
        self.cache_format = cache_format
        self.json_max_values = int(json_max_values)
        # Where the reference's legacy stream (rng_mode="legacy") is produced: "device" = MT19937 + polar Box-Muller on
        # the GPU (`rc_draws_legacy_f64`: same uint32 stream, same generator state afterwards, and - round 5 - the SAME
        # normals bit for bit: glibc's log restated operation for operation; on a host whose libm is another one
        # (`backend.legacy_device_exact()` false) the draws are made by NumPy instead), "host" = NumPy itself (~20 ns per draw).
        if legacy_draws not in ("device", "host"):
            raise ValueError("legacy_draws must be 'device' or 'host'")
        self.legacy_draws = legacy_draws
        # devices: None = one GPU, torch's current device (one process per GPU under torch.distributed);  "all" or a list
        # of ordinals = ALL THOSE GPUs FROM THIS ONE PROCESS through the C ABI's multi-device entry
        # (`rc_mc_metrics_sharded_f64`: controller blocks, one host thread + stream per device, metric rows and - for a
        # `.mc` cache - fidelities assembled on the host; legacy draws come from NumPy on the host in this mode).
        self.devices = None if devices is None else (list(range(backend._lib.require_gpu())) if devices == "all"
                                                      else [int(d) for d in devices])
        self._mc_writers = {}            # path -> cache_io.McWriter
        self._metric_rows = {}           # algo -> (fidelity object, (15, L, C) host metric rows computed with it)

        self.get_controller_name = self.get_experiment_name(experiment_name)()
        if self.filemarker is not None:
            self.get_controller_name += self.filemarker
        self._say(self.get_controller_name)
        try:
            self.controllers = self.load_controllers()
            self.algos = self.ctrlnames(self.controllers)
        except FileNotFoundError as e:            # mcsim.py:231-237: flagged, not fatal
            self._say("flagging: ", e)
            self.controllers = None
            self.algos = None

        self.noise_model = structured_perturbation(**self.args)
        # accepted for signature compatibility; the GPU batch replaces the per-controller Pool (mcsim.py:451-455)
        self.parallel = parallel
        self.num_workers = num_workers
        self.colors = ["blue", "orange", "gold", "purple", "pink", "brown",
                       "red", "cyan", "gray", "mediumseagreen", "olive"]
        self.figlabels = ["({})".format(i) for i in "abcdefghijklmnopqrstuvwxyz"]
        self.controller = None

    # ------------------------------------------------------------------ small helpers
    def _say(self, *a):
        if self.verbose:
            print(*a)

    def set_fig_save_directory(self, cur_save_folder):
        self.cur_save_folder = cur_save_folder
        if not os.path.exists(cur_save_folder):
            os.mkdir(cur_save_folder)

    def ctrlnames(self, ctrlcontainer) -> List:
        """Algorithm names of a controller container; empty per-algo dicts are dropped (mcsim.py:337-349)."""
        if isinstance(ctrlcontainer, dict):
            for key in [k for k, v in ctrlcontainer.items() if v == {}]:
                ctrlcontainer.pop(key)
            return list(ctrlcontainer.keys())
        if isinstance(ctrlcontainer, (list, np.ndarray)):
            return ["unnamed"]
        raise TypeError("need controller container either as a list or a dict")

    def get_experiment_name(self, experiment_name: str) -> Callable[[str], ExperimentNamer]:
        return ExperimentNamer(experiment_name=experiment_name, numcontrollers=self.numcontrollers, **self.args)

    def get_mcname(self, training_noise=None, noises=None) -> str:
        """Cache file name (mcsim.py:351-356); `noises` is embedded as ``str(np.ndarray)``."""
        if training_noise is None:
            training_noise = self.training_noise
        if noises is None:
            noises = self.noises
        return self.get_controller_name + "_tn{}_br_{}_nlvl{}.mc".format(training_noise, self.bootreps, noises)

    def load_controllers(self, controllers=None):
        if controllers is None:
            return json.load(open(self.get_controller_name, "rb"))
        if isinstance(controllers, str):
            return json.load(open(controllers, "rb"))
        if isinstance(controllers, (list, np.ndarray)):
            return controllers

    def loadsimdata(self, simname: str):
        """`json.load` of a cache file (mcsim.py:366-367); an `.mc` written as an npy index resolves to arrays."""
        return cache_io.load_mc(simname)

    def get_controller_fid_dist_boot(self, x=None):
        """One noisy evaluation of `self.controller` (mcsim.py:369-374); NaN controller -> NaN, no RNG use."""
        if self.controller is not np.nan:
            return self.noise_model.evaluate_noisy_fidelity(self.controller, ham_noisy=True)
        return np.nan

    # ------------------------------------------------------------------ distributed plumbing
    @staticmethod
    def _dist():
        try:
            import torch.distributed as dist
            # ROBCHAR_FORCE_DIST=1: take the sharded code path even with ONE rank (rehearsal of the RCCL branches on a
            # one-GPU box: scatter, all-gathers, state broadcast all execute, on trivial partitions)
            if dist.is_available() and dist.is_initialized() and (
                    dist.get_world_size() > 1 or os.environ.get("ROBCHAR_FORCE_DIST", "0") == "1"):
                return dist
        except Exception:
            pass
        return None

    def _is_writer(self) -> bool:
        d = self._dist()
        return d is None or d.get_rank() == 0

    def _barrier(self) -> None:
        d = self._dist()
        if d is not None:
            d.barrier()

    # ------------------------------------------------------------------ the MC itself
    def _controller_rows(self, algoname, training_noise):
        """Controller list of one algorithm: keyed by str(training_noise), lbfgs by str(Nspin)
        (mcsim.py:427-441)."""
        key = str(self.Nspin) if algoname == "lbfgs" else str(training_noise)
        return self.controllers[algoname][key]["controller"]

    def _gather_rows(self, local, rows_max: int):
        """All-gather along dim 0 of per-rank tensors padded to `rows_max` rows -> (world, rows_max, ...) on every rank.
        RCCL moves device tensors over xGMI; gloo (CPU tests, one-GPU rehearsals) hops through host memory."""
        import torch
        d = self._dist()
        world = d.get_world_size()
        shard = local
        if local.shape[0] != rows_max:
            shard = torch.zeros((rows_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            shard[: local.shape[0]] = local
        shard = shard.contiguous()
        via_host = shard.is_cuda and d.get_backend() != "nccl"
        if via_host:
            shard = shard.cpu()
        out = torch.empty((world * rows_max,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        d.all_gather_into_tensor(out, shard)
        if via_host:
            out = out.to(local.device)
        return out.view((world, rows_max) + tuple(shard.shape[1:]))

    def _level_draws(self, nvalid: int, lo: int, hi: int, dev, buf):
        """Device tensor (hi - lo, K, N, 3) of this rank's draws for the current level (sigma is set on the noise
        model).  legacy: the reference's sequential stream - drawn by the only process, or by rank 0 which scatters
        the slices; philox: this rank's slice of the counter-based stream, generated in place."""
        import torch
        N, K = self.Nspin, self.bootreps
        per_ctrl = K * N * 3
        d = self._dist()
        if self.rng_mode == "philox":
            sigma = float(self.noise_model.rng.args.get("scale", self.noise_model.noise))
            base = self._philox_offset
            self._philox_offset += nvalid * per_ctrl
            out = buf[: (hi - lo) * per_ctrl].view(hi - lo, K, N, 3)
            if hi > lo:
                backend.philox_normal(out.shape, self.seed, scale=sigma, offset=base + lo * per_ctrl, out=out)
            return out
        if d is None:
            host = self.noise_model.draw_samples(nvalid, K)
            return torch.from_numpy(np.ascontiguousarray(host)).to(dev)
        from .sharding import controller_partition
        world, rank = d.get_world_size(), d.get_rank()
        bounds = controller_partition(nvalid, world)
        rows = max(b[1] - b[0] for b in bounds)
        on_host = d.get_backend() != "nccl"
        mine = torch.empty((rows, K, N, 3), dtype=torch.float64, device="cpu" if on_host else dev)
        if rank == 0:
            host = torch.from_numpy(np.ascontiguousarray(self.noise_model.draw_samples(nvalid, K)))
            full = host if on_host else host.to(dev)
            pieces = []
            for (a, b) in bounds:
                piece = torch.zeros((rows, K, N, 3), dtype=torch.float64, device=full.device)
                piece[: b - a] = full[a:b]
                pieces.append(piece)
            d.scatter(mine, pieces, src=0)
        else:
            d.scatter(mine, None, src=0)
        return mine[: hi - lo].to(dev)

    _LEGACY_DEVICE_MAX_DRAWS = 1 << 29           # draws generated per device call (4 GiB of fp64)
    _BATCH_LEVELS_MAX_BYTES = 1 << 30            # all levels of an algorithm in one fidelity launch up to this many bytes of draws

    def _device_legacy_levels(self, noises: np.ndarray, nvalid: int, lo: int, hi: int, dev):
        """The reference's legacy stream for ALL levels of an algorithm, produced on the GPU: per level one burned draw
        and nvalid*K*3N draws scaled by the level's sigma (`rc_draws_legacy_f64`: period = 1 + nvalid*K*3N, skip = 1).
        Returns `block(j)` -> this rank's (hi - lo, K, N, 3) draws of level j.

        Under sharding EVERY rank runs the generator on its own GPU from the same `RandomState` (`_run_algo` aligns the
        states first) and keeps the rows [lo, hi) of each level: the stream is sequential - accepted polar attempts
        decide where a value lands, so no rank can jump to its slice without counting everything before it - but
        re-generating it on the device costs milliseconds (6.6e7 normals: 7.6 ms), moves nothing between ranks, leaves no
        rank waiting for rank 0, and every rank ends at the reference's stream position by itself.  (Round 2: rank 0
        generated and scattered zero-padded slices level by level - L collectives and 7 idle GPUs per algorithm.)"""
        N, K = self.Nspin, self.bootreps
        per_level = nvalid * K * N * 3
        L = int(noises.size)
        per_call = max(1, self._LEGACY_DEVICE_MAX_DRAWS // max(per_level, 1))          # levels per generator call
        cache = {}

        def generate(j0):
            j1 = min(L, j0 + per_call)
            return backend.legacy_normal_periods(j1 - j0, 1 + per_level, 1, noises[j0:j1], device=dev)

        def block(j):
            j0 = (j // per_call) * per_call
            if cache.get("j0") != j0:
                cache.clear()
                cache.update(j0=j0, data=generate(j0))
            if not per_level:
                return None
            return cache["data"][j - j0].view(nvalid, K, N, 3)[lo:hi]

        def all_levels():
            """(L, nvalid, K, N, 3) - every level in one tensor - when one generator call covers the algorithm, else None"""
            if per_call < L or not per_level or not L:
                return None
            block(0)
            return cache["data"].view(L, nvalid, K, N, 3)

        block.all_levels = all_levels
        self.noise_model.rng.args.update(scale=noises[-1] if L else self.noise_model.rng.args.get("scale"))
        return block

    def _sync_legacy_rng(self):
        """Every rank adopts rank 0's generator state.  Host-drawn legacy mode: after the run (rank 0 alone consumed the
        reference's stream and scattered slices); device-drawn legacy mode: BEFORE the run (every rank then generates the
        same stream on its own GPU and ends at the same position).  Either way the process group as a whole is at the
        reference's stream position afterwards."""
        d = self._dist()
        if d is None or self.rng_mode != "legacy":
            return
        box = [np.random.get_state() if d.get_rank() == 0 else None]
        d.broadcast_object_list(box, src=0)
        if d.get_rank() != 0:
            np.random.set_state(box[0])

    def _run_algo(self, algoname: str, noises: np.ndarray, training_noise):
        """The device-resident MC of one algorithm: (L, nvalid, K) fidelities on the GPU and the (15, L, C) metric rows
        on the host (rows of `backend.packed_views`; NaN / -0 conventions of the reference for padded controllers)."""
        import torch
        from .sharding import controller_partition
        L, C, K, N = int(noises.size), self.numcontrollers, self.bootreps, self.Nspin
        rows_all = self._controller_rows(algoname, training_noise)
        nvalid = min(len(rows_all), C)
        ctrl = np.asarray(rows_all[:nvalid], dtype=np.float64).reshape(nvalid, N + 1)
        d = self._dist()
        if self.devices is not None and d is None:
            return self._run_algo_multi_device(algoname, noises, training_noise, rows_all, ctrl)
        dev = backend.compute_device()
        world, rank = (d.get_world_size(), d.get_rank()) if d is not None else (1, 0)
        bounds = controller_partition(nvalid, world)
        lo, hi = bounds[rank]
        nloc = hi - lo
        rows_max = max(b[1] - b[0] for b in bounds)
        ctrl_dev = torch.from_numpy(ctrl[lo:hi]).to(dev) if nloc else None
        fid_loc = torch.empty((L, nloc, K), dtype=torch.float64, device=dev)
        # philox mode on a chain of <= 13 (14) spins: the draws are generated INSIDE the fidelity kernel (round 4:
        # rc_mc_fidelity_philox_f64_async - same stream elements, bit-identical fidelities, no (C, K, N, 3) tensor: 16.8 GB per
        # level at BASELINE config 4's size, and 30 % less kernel time than generator + fidelity kernel)
        h_diag, h_off, h_ring, h_imag = self.noise_model._static_terms()
        fused = (self.rng_mode == "philox" and dev.type == "cuda" and not h_imag.any()
                 and backend.philox_fused_supported(N, h_ring) and backend.philox_fused_pays(N, self.inspin, self.outspin))
        # (philox_fused_pays asks the library - rc_philox_fused_pays, which also reads ROBCHAR_PHILOX_FUSED - so that this route
        # and the single-process multi-device one decide by ONE copy of the rule)
        if fused:
            return self._run_algo_philox_fused(algoname, noises, training_noise, rows_all, ctrl_dev, fid_loc, d, bounds, rank,
                                               (h_diag, h_off))
        buf = torch.empty((nloc * K * N * 3,), dtype=torch.float64, device=dev) if self.rng_mode == "philox" else None
        on_device = (self.rng_mode == "legacy" and self.legacy_draws == "device" and dev.type == "cuda"
                     and backend.legacy_stream_usable(self.noise_model.rng))
        if on_device and d is not None:
            self._sync_legacy_rng()                       # one stream position for the group; each rank generates from it
        level_block = self._device_legacy_levels(noises, nvalid, lo, hi, dev) if on_device else None
        # One process, draws made on the device, everything of the algorithm fits a modest buffer: the L levels go through
        # ONE fidelity launch over L * nvalid "controllers" (the rows tiled L times) instead of L launches - at the paper's
        # scale a level is 10 us of kernel behind ~130 us of Python and launch overhead (scripts/profile_paper_scale.py).
        all_bytes = L * nloc * K * N * 3 * 8
        batched = (d is None and nloc and K and L > 1 and all_bytes <= self._BATCH_LEVELS_MAX_BYTES
                   and (level_block is not None or self.rng_mode == "philox"))
        draws_all = None
        if batched and level_block is None:
            draws_all = torch.empty((L, nloc, K, N, 3), dtype=torch.float64, device=dev)
        for j, noise in enumerate(_progress(noises[:]) if self.verbose else noises[:]):
            self._say(algoname, training_noise)
            if level_block is not None:                   # burn + draws of this level are produced on the GPU
                if batched:
                    continue
                draws = level_block(j)                    # (under sharding: generated on every rank's own GPU)
                if nloc and K:
                    self.noise_model.fidelity_from_draws(ctrl_dev, draws, out=fid_loc[j])
                continue
            # sets sigma_sim AND burns one draw (mcsim.py:425); under sharding only rank 0 owns the legacy stream
            if d is None or self.rng_mode == "philox" or rank == 0:
                self.noise_model.rng(scale=noise)
            else:
                self.noise_model.rng.args.update(scale=noise)
            if nvalid and K:
                draws = self._level_draws(nvalid, lo, hi, dev, draws_all[j].view(-1) if batched else buf)
                if nloc and not batched:
                    self.noise_model.fidelity_from_draws(ctrl_dev, draws, out=fid_loc[j])
        if batched:
            if level_block is not None:
                draws_all = level_block.all_levels()
            if draws_all is None:                         # (several generator calls per algorithm: level by level after all)
                for j in range(L):
                    self.noise_model.fidelity_from_draws(ctrl_dev, level_block(j), out=fid_loc[j])
            else:
                self.noise_model.fidelity_from_draws(ctrl_dev.repeat(L, 1), draws_all.view(L * nloc, K, N, 3),
                                                     out=fid_loc.view(L * nloc, K))
        # the reference leaves the last visited controller on the instance (mcsim.py:445)
        self.controller = rows_all[C - 1] if len(rows_all) >= C else np.nan
        if level_block is None:
            self._sync_legacy_rng()
        return self._finish_algo(fid_loc, d, bounds, nvalid)

    def _run_algo_philox_fused(self, algoname, noises, training_noise, rows_all, ctrl_dev, fid_loc, d, bounds, rank, h0):
        """`_run_algo` for rng_mode='philox' with the draws generated inside the fidelity kernel: per level the same stream
        elements as `_level_draws` would have produced (level j starts at `_philox_offset`, which advances by nvalid * K * 3N;
        this rank's rows [lo, hi) start lo * K * 3N further), the same burned draw of `rng(scale=...)` (mcsim.py:425).  One
        process: ALL levels in one launch - the controller rows tiled L times, one scale per row."""
        import torch
        L, C, K, N = int(noises.size), self.numcontrollers, self.bootreps, self.Nspin
        nvalid = min(len(rows_all), C)
        lo, hi = bounds[rank]
        nloc = hi - lo
        per_ctrl = K * N * 3
        offsets = []
        for j, noise in enumerate(_progress(noises[:]) if self.verbose else noises[:]):
            self._say(algoname, training_noise)
            self.noise_model.rng(scale=noise)             # sets sigma_sim AND burns one draw (mcsim.py:425)
            offsets.append(self._philox_offset + lo * per_ctrl)
            if nvalid and K:
                self._philox_offset += nvalid * per_ctrl
        kw = dict(h0_diag=h0[0], h0_offdiag=h0[1])
        if nloc and K:
            if d is None and L > 1:
                sig = torch.as_tensor(np.repeat(np.asarray(noises, dtype=np.float64), nloc), device=fid_loc.device)
                backend.mc_fidelity_philox(ctrl_dev.repeat(L, 1), K, N, self.inspin, self.outspin, self.seed, offset=offsets[0],
                                           sigma=sig, out=fid_loc.view(L * nloc, K), **kw)
            else:
                for j in range(L):
                    backend.mc_fidelity_philox(ctrl_dev, K, N, self.inspin, self.outspin, self.seed, offset=offsets[j],
                                               sigma=float(noises[j]), out=fid_loc[j], **kw)
        self.controller = rows_all[C - 1] if len(rows_all) >= C else np.nan
        self._sync_legacy_rng()
        return self._finish_algo(fid_loc, d, bounds, nvalid)

    def _finish_algo(self, fid_loc, d, bounds, nvalid):
        """Reductions, exchange step and host rows of one algorithm's (L, nloc, K) fidelities (tail of `_run_algo`)."""
        import torch
        L, nloc, K = (int(v) for v in fid_loc.shape)
        C = self.numcontrollers
        dev = fid_loc.device
        rows_max = max(b[1] - b[0] for b in bounds)
        eps = compute_dkw_error(self.alpha, K) if K else 0.0
        # (nothing of this pipeline runs beside the reduction: the standalone route)
        packed = backend.reduce_packed(fid_loc.view(L * nloc, K), eps, overlapped=False) if (nloc and K) else \
            torch.empty((backend.PACKED_ROWS, 0), dtype=torch.float64, device=dev)
        packed = packed.view(backend.PACKED_ROWS, L, nloc)
        need_fids = self.cache_format != "none"
        if d is not None:
            # exchange step: metric rows always (15 x L doubles per controller), fidelity slabs only for a `.mc` cache
            g = self._gather_rows(packed.permute(2, 0, 1).contiguous(), rows_max)          # (world, rows_max, 15, L)
            packed = torch.cat([g[r, : b[1] - b[0]] for r, b in enumerate(bounds)]).permute(1, 2, 0).contiguous()
            if need_fids:
                g = self._gather_rows(fid_loc.permute(1, 0, 2).contiguous(), rows_max)     # (world, rows_max, L, K)
                fid_loc = torch.cat([g[r, : b[1] - b[0]] for r, b in enumerate(bounds)]).permute(1, 0, 2).contiguous()
        rows = np.full((backend.PACKED_ROWS, L, C), np.nan)
        rows[9:15, :, nvalid:] = 0.0                       # Q of a NaN row: no sample passes a threshold (mcsim.py:144-146)
        if nvalid:
            rows[:, :, :nvalid] = packed.cpu().numpy()
        return DeviceFids(fid_loc, C), rows

    def _run_algo_multi_device(self, algoname, noises, training_noise, rows_all, ctrl):
        """`_run_algo` for `devices=[...]`: every sigma level is ONE call of the C ABI's multi-device entry - the
        controllers split into contiguous blocks over the GPUs, draws -> fidelities -> reductions on each of them, the
        (15, C) metric rows (and, unless cache_format='none', the fidelities) assembled in host arrays."""
        L, C, K, N = int(noises.size), self.numcontrollers, self.bootreps, self.Nspin
        nvalid = ctrl.shape[0]
        diag, off, ring, imag = self.noise_model._static_terms()
        need_fids = self.cache_format != "none"
        eps = compute_dkw_error(self.alpha, K) if K else 0.0
        rows = np.full((backend.PACKED_ROWS, L, C), np.nan)
        rows[9:15, :, nvalid:] = 0.0
        fids = np.full((L, C, K), np.nan) if need_fids else None
        for j, noise in enumerate(_progress(noises[:]) if self.verbose else noises[:]):
            self.noise_model.rng(scale=noise)              # sets sigma_sim AND burns one draw (mcsim.py:425)
            self._say(algoname, training_noise)
            if not (nvalid and K):
                continue
            kw = dict(devices=self.devices, h0_diag=diag, h0_offdiag=off, ring=ring, dkw_eps=eps, want_fid=need_fids)
            if self.rng_mode == "philox":
                if imag.any():
                    raise NotImplementedError("complex static couplings need rng_mode='legacy' in multi-device mode")
                base = self._philox_offset
                self._philox_offset += nvalid * K * N * 3
                res = backend.mc_metrics_sharded(ctrl, K, N, self.inspin, self.outspin, seed=self.seed, offset=base,
                                                 sigma=float(noise), **kw)
            else:
                draws = np.array(self.noise_model.draw_samples(nvalid, K), dtype=np.float64)
                if imag.any():
                    draws[..., 1:, 2] += imag
                res = backend.mc_metrics_sharded(ctrl, K, N, self.inspin, self.outspin, draws=draws, **kw)
            rows[0:3, j, :nvalid], rows[3:6, j, :nvalid], rows[6:9, j, :nvalid] = res["rim1"], res["std"], res["min"]
            rows[9:15, j, :nvalid] = res["q"].reshape(6, nvalid)
            if need_fids:
                fids[j, :nvalid] = res["fid"]
        self.controller = rows_all[C - 1] if len(rows_all) >= C else np.nan
        return DeviceFids(None, C, host=fids, shape=(L, C, K)), rows

    def get_algo_fid_dist(self, algoname: str, allalgoallfids: dict, noises, training_noise):
        """(L, C, K) fidelity tensor of one algorithm, stored into `allalgoallfids` (as a `DeviceFids` handle) and the
        whole dict dumped to the `.mc` file - mcsim.py:422-460."""
        noises = np.asarray(noises)
        fids, rows = self._run_algo(algoname, noises, training_noise)
        allalgoallfids[algoname] = fids
        self._metric_rows[algoname] = (fids, rows)
        if self._is_writer() and self.cache_format != "none":
            self._dump_mc(allalgoallfids, self.get_mcname(training_noise, noises))
        self._barrier()                  # no rank reads a cache file the writer is still writing
        return allalgoallfids

    def _dump_mc(self, simdict: dict, path: str) -> None:
        """`json.dump(simdict, open(path, "w"))` after every algorithm (mcsim.py:457-459), as an append: the file is a
        complete JSON object after each call, earlier algorithms are neither re-encoded nor re-written."""
        w = self._mc_writers.get(path)
        if w is None:
            w = self._mc_writers[path] = cache_io.McWriter(path, self.json_max_values, self.cache_format)
        w.dump(simdict)

    def get_fid_dists(self, training_noise: str = None, noises: np.ndarray = None, algoname=None) -> dict:
        """Cache-aware entry (mcsim.py:382-419): load the `.mc` file when present and only compute the
        algorithms missing from it; otherwise compute every requested algorithm (from `lbfgs` onwards with
        ``training_noise=None``, as the reference does)."""
        if isinstance(algoname, str):
            algos = [algoname]
        elif algoname is None:
            algos = self.algos
        if noises is None:
            noises = self.noises
        if training_noise is None:
            training_noise = self.training_noise

        path = self.get_mcname(training_noise, noises)
        if os.path.exists(path):
            simdict = self.loadsimdata(path)
            if self._is_writer() and self.cache_format != "none" and isinstance(simdict, dict):
                # resume: what is on disk stays as it is, missing algorithms are appended (never re-dump a loaded
                # tensor - an `.npy` sidecar would be rewritten from its own memory map)
                self._mc_writers[path] = cache_io.McWriter.resume(path, simdict, self.json_max_values, self.cache_format)
            for name in algos:
                if name not in simdict:
                    self.get_algo_fid_dist(name, simdict, noises, training_noise)
        else:
            simdict = {}
            for name in algos:
                if name == "lbfgs":
                    training_noise = None
                self.get_algo_fid_dist(name, simdict, noises, training_noise)
        for name in simdict.keys():
            if name not in algos:
                raise Exception(f"Fid distribution generation for {name} was unsuccessful.")
        return simdict

    def _metric_arrays(self, algo: str, dists) -> dict:
        """{metric(+suffix): (L, C) array} of one algorithm (mcsim.py:480-500).  Rows computed together with `dists` on
        the GPU are reused; a tensor that came from a cache file is uploaded and reduced (one launch for all levels)."""
        hit = self._metric_rows.get(algo)
        if hit is not None and hit[0] is dists:
            rows = hit[1]
        else:
            import torch
            T = np.ascontiguousarray(np.asarray(dists, dtype=np.float64))
            L, C, K = T.shape
            eps = compute_dkw_error(self.alpha, self.bootreps)
            dev = backend.compute_device()
            packed = backend.reduce_packed(torch.from_numpy(T.reshape(L * C, K)).to(dev), eps, overlapped=False)
            rows = packed.cpu().numpy().reshape(backend.PACKED_ROWS, L, C)
        out = {}
        for v, suffix in enumerate(("", " upper", " lower")):
            out[METRIC_NAMES[0] + suffix] = rows[0 + v]
            out[METRIC_NAMES[1] + suffix] = -rows[9 + 2 * v]          # Q and worst-case are stored negated
            out[METRIC_NAMES[2] + suffix] = -rows[10 + 2 * v]         # (mcsim.py:148-149, :169-176)
            out[METRIC_NAMES[3] + suffix] = rows[3 + v]
            out[METRIC_NAMES[4] + suffix] = -rows[6 + v]
        # key order of the reference: metric-major, then "", " upper", " lower"
        return {name + suffix: out[name + suffix] for name in METRIC_NAMES for suffix in ("", " upper", " lower")}

    def metrics_for_tensor(self, dists_tensor) -> dict:
        """{metric(+suffix): [L][C]} of an (L, C, K) tensor (mcsim.py:480-500) on the GPU."""
        return {k: v.tolist() for k, v in self._metric_arrays("", dists_tensor).items()}

    def get_metrics_dict(self, training_noise: str = None, noises: np.ndarray = None, algoname=None):
        "dict of the 5 metrics x {centre, upper, lower} per algorithm; cached as `.mcm` (mcsim.py:463-510)"
        if training_noise is None:
            training_noise = self.training_noise
        if noises is None:
            noises = self.noises
        noises = np.asarray(noises)
        path = self.get_mcname(training_noise, noises) + "m"
        if os.path.exists(path):
            return self.loadsimdata(path)
        # cold cache: the reference always recomputes for ALL algorithms here (mcsim.py:509)
        algofiddists = self.get_fid_dists(training_noise, noises, None)
        arrays = {algo: self._metric_arrays(algo, algofiddists[algo]) for algo in self.algos}
        if self._is_writer():
            cache_io.write_json(arrays, path)
        self._barrier()
        # (writing the file on a second thread while this one builds the lists was tried: 19 -> 22 ms at the paper's scale -
        # the encoder threads and `tolist()` contend for the interpreter lock)
        return {algo: {k: v.tolist() for k, v in tab.items()} for algo, tab in arrays.items()}

    # ------------------------------------------------------------------ second caller of the kernel
    _RIMS_CHUNK_DRAWS = 1 << 27          # host draws per batch (1 GiB of fp64)

    def _rims_batch(self, conts, noises) -> np.ndarray:
        """RIM_1 = 1 - mean fidelity of every (controller, sigma level) pair: (M, L).  RNG consumption is the
        reference's nested order (gen_fig_8_arim_fcall_scaling.py:55-69 around :121-132): per controller, per level,
        ONE burned draw (`rng(scale=nlvl)`) and then K x 3N draws - realised as one `standard_normal` block per batch
        of rows (row = one (controller, level) pair, 1 + 3NK values, scaled by the row's sigma), ONE fidelity launch
        and ONE rim1 reduction per batch instead of M x L x K single-sample evaluations."""
        import torch
        conts = np.asarray(conts, dtype=np.float64).reshape(-1, self.Nspin + 1)
        noises = np.asarray(noises, dtype=np.float64).reshape(-1)
        M, L, K, N = conts.shape[0], noises.size, self.bootreps, self.Nspin
        rng = self.noise_model.rng
        batched = (rng.generator is np.random.normal and set(rng.args) <= {"scale", "loc"}
                   and float(rng.args.get("loc", 0.0)) == 0.0)
        rims = np.zeros((M, L))
        if M == 0 or L == 0:
            return rims
        dev = backend.compute_device()
        per_row = 1 + 3 * N * K
        rows_per_batch = max(L, (self._RIMS_CHUNK_DRAWS // per_row) // L * L)       # whole controllers per batch
        R = M * L
        sig_all = np.tile(noises, M)
        ctrl_rows = np.repeat(conts, L, axis=0)
        for r0 in range(0, R, rows_per_batch):
            r1 = min(R, r0 + rows_per_batch)
            if batched and self.legacy_draws == "device" and dev.type == "cuda" and backend.legacy_device_exact():
                # the same stream continued on the GPU: one period per row, its first draw burned
                draws = backend.legacy_normal_periods(r1 - r0, per_row, 1, sig_all[r0:r1], device=dev).view(r1 - r0, K, N, 3)
            elif batched:
                z = np.random.standard_normal((r1 - r0, per_row))                    # column 0: the burned draw
                draws = torch.from_numpy((z[:, 1:] * sig_all[r0:r1, None]).reshape(r1 - r0, K, N, 3)).to(dev)
            else:                                                                    # user-supplied generator
                draws = np.empty((r1 - r0, K, N, 3))
                for r in range(r0, r1):
                    rng(scale=sig_all[r])
                    draws[r - r0] = self.noise_model.draw_samples(1, K)[0]
                draws = torch.from_numpy(draws).to(dev)
            fid = self.noise_model.fidelity_from_draws(torch.from_numpy(ctrl_rows[r0:r1]).to(dev), draws)
            red = backend.reduce_metrics(fid, q_thresholds=(), overlapped=False)
            rims.reshape(-1)[r0:r1] = red["rim1"][0].cpu().numpy()
        rng.args.update(scale=noises[-1])              # sticky sigma of the last `rng(scale=...)` call
        return rims

    def get_rims(self, cont, noises=None):
        """`NStochOpt.get_rims` (gen_fig_8_arim_fcall_scaling.py:121-132): for every noise level one burned draw, K
        noisy evaluations of `cont`, returns 1 - mean fidelity per level - all levels in one launch."""
        noises = self.noises if noises is None else noises
        return self._rims_batch(np.asarray(cont, dtype=np.float64).reshape(1, -1)[:, : self.Nspin + 1], noises)[0]

    def get_arims(self, algo="lbfgs", nlvl="0.01", marker="", cdict=None):
        """`NStochOpt.get_arims` (gen_fig_8_arim_fcall_scaling.py:37-69): ARIM (mean RIM_1 over the controllers) per
        function-call checkpoint and sigma level for ``cdict[algo][nlvl] = {checkpoint: [controllers]}``; checkpoints
        with fewer than `numcontrollers` controllers are dropped FROM `cdict` (as there); the `(checkpoints, L)` array
        is pickled to ``<controller file>_arims_<algo><nlvl><marker>.pickle`` and served from it afterwards.  Returns
        ``(arims, kept_checkpoint_keys)`` - ``(arims, None)`` on a cache hit.  All (checkpoint, controller, level)
        triples go through `_rims_batch`: the legacy stream is consumed in the reference's order."""
        import pickle
        save_fname = self.get_controller_name + "_arims_" + algo + nlvl + marker + ".pickle"
        if os.path.exists(save_fname):
            return pickle.load(open(save_fname, "rb")), None
        if cdict is None or algo not in cdict:
            raise Exception("Unaccounted for case encountered.")
        fcall_dict = cdict[algo][nlvl]
        for key in list(fcall_dict.keys()):
            if len(fcall_dict[key]) < self.numcontrollers:
                fcall_dict.pop(key)
        new_keys = list(fcall_dict.keys())
        L = len(self.noises)
        arims = np.zeros((len(new_keys), L))
        counts = [len(fcall_dict[k]) for k in new_keys]
        if any(n > self.numcontrollers for n in counts):
            # rims_all has numcontrollers rows in the reference: a longer list is an IndexError there too
            raise IndexError("a checkpoint holds more controllers than numcontrollers")
        if new_keys:
            conts = np.concatenate([np.asarray(fcall_dict[k], dtype=np.float64).reshape(-1, self.Nspin + 1)
                                    for k in new_keys])
            rims = self._rims_batch(conts, self.noises)
            start = 0
            for j, n in enumerate(counts):
                arims[j] = rims[start:start + n].sum(axis=0) / n
                start += n
        if self._is_writer():
            pickle.dump(arims, open(save_fname, "wb"))
        return arims, new_keys

    # ------------------------------------------------------------------ cache / controller-file tooling
    def get_path(self, directory_exportable, of: str = "controllers"):
        """Controller file / cache files of another experiment directory (mcsim.py:571-592)."""
        import glob
        rootpath = self.global_experiments_directory + directory_exportable
        self._say(rootpath)
        if not os.path.exists(rootpath):
            raise DirectoryDoesNotExistError(self.global_experiments_directory)
        path = self.get_experiment_name(directory_exportable)()
        self._say(path)
        if self.filemarker is not None:
            path += self.filemarker
        if not os.path.exists(path):
            raise DirectoryDoesNotExistError(path)
        if of == "controllers":
            return path
        if of == "mcm":
            return glob.glob(glob.escape(path) + "**.mcm")
        if of == "mc":
            return glob.glob(glob.escape(path) + "**.mc")
        raise Exception("No such object type exists. Please specify a correct .description.")

    def load_controllers_in_dir(self, directory_exportable):
        return self.load_controllers(self.get_path(directory_exportable, of="controllers"))

    def merge_controller_files(self, directory_exportable: str) -> None:
        """Merge the same-named controller file of another experiment directory into this one: whole algorithms
        that are missing here, and for the noise-keyed algorithms the training-noise entries missing here
        (mcsim.py:629-649).  Rewrites this experiment's controller file."""
        alt = self.load_controllers_in_dir(directory_exportable)
        for algo in self.ctrlnames(alt):
            if algo not in self.controllers:
                self.controllers[algo] = alt[algo]
            elif algo != "lbfgs":
                for noise_key, entry in alt[algo].items():
                    self.controllers[algo].setdefault(noise_key, entry)
        cache_io.write_json(self.controllers, self.get_controller_name)

    def merge_mcdata(self, directory_exportable):
        """Merge the `.mc` / `.mcm` caches of another experiment directory (same file names) into this one's:
        algorithms missing here are copied over (mcsim.py:594-622).  NOTE: the reference writes the merged METRIC
        dict into the `.mc` path and the merged FIDELITY dict into the `.mcm` path (mcsim.py:619-620, swapped
        targets); here each goes back to its own file."""
        exportable = self.global_experiments_directory + directory_exportable
        fid_paths = sorted(self.get_path(self.experiment_name, of="mc"))
        met_paths = sorted(self.get_path(self.experiment_name, of="mcm"))
        for fid_path, met_path in zip(fid_paths, met_paths):
            mine_f, mine_m = self.loadsimdata(fid_path), self.loadsimdata(met_path)
            other_f = self.loadsimdata(exportable + "/" + fid_path.split("/")[-1])
            other_m = self.loadsimdata(exportable + "/" + met_path.split("/")[-1])
            for algo, val in other_f.items():
                mine_f.setdefault(algo, val)
            for algo, val in other_m.items():
                mine_m.setdefault(algo, val)
            cache_io.write_json(mine_f, fid_path)
            cache_io.write_json(mine_m, met_path)
        self._say("files successfully merged")

    @staticmethod
    def get_ranks(array):
        order = np.argsort(array)
        ranks = np.zeros_like(order)
        ranks[order] = np.arange(len(order))
        return ranks
