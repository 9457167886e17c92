"""`MCDataSim` - the Monte-Carlo driver and its cached-results layout, API mirror of mcsim.py:200-510.

Same constructor kwargs, attributes, method names, file names and JSON layouts as the reference so that the
reference's figure scripts (subclasses of `MCDataSim`) and its on-disk caches interoperate:

    <exp dir>/ppo_spin_{N}_{in}-{out}_c_{C}{filemarker}                         controller file (.le)  (input)
    ..._tn{training_noise}_br_{K}_nlvl{str(noises)}.mc    {algo: [L][C][K]}     fidelity cache
    ..._tn{training_noise}_br_{K}_nlvl{str(noises)}.mcm   {algo: {metric(+" upper"|" lower"): [L][C]}}

What is different is the execution: the reference's triple loop (noise level x controller x draw, one
`expm` per iteration, mcsim.py:424-449) becomes, per noise level, ONE batched draw of the (C, K, N, 3)
perturbation tensor from the same legacy RNG stream and ONE launch of the HIP fidelity kernel; the metric
maps (mcsim.py:480-500) become one launch of the reduction kernel per noise level.  RNG consumption is
identical to the reference: one burned draw per level (the value returned by ``rng(scale=noise)``,
mcsim.py:425), then 3 N draws per sample in (controller, draw, site, slot) order, nothing for NaN-padded
controllers (mcsim.py:370-374, :442-443).

With an initialised torch.distributed group (one process per GPU) the controllers of each level are sharded
over the ranks (sharding.py) and all-gathered; every rank draws the full tensor from its own copy of the
stream, so the result is bit-identical to the single-GPU run, and only rank 0 writes cache files.
"""
from __future__ import annotations

import json
import os
from typing import Callable, List

import numpy as np

from . import rim_metrics
from .naming import DirectoryDoesNotExistError, ExperimentNamer  # noqa: F401  (re-exported like mcsim.py:26)
from .noise import structured_perturbation
from .rim_metrics import METRIC_NAMES, compute_dkw_error


def _progress(seq):
    try:
        from tqdm import tqdm
        return tqdm(seq)
    except Exception:          # tqdm is optional here
        return seq



def _json_write(obj, path: str) -> None:
    """Same bytes as `json.dump(obj, open(path, "w"))`, through the C encoder (`json.dump` streams through the
    pure-Python chunk iterator: 4x slower on the multi-megabyte metric / fidelity dicts)."""
    with open(path, "w") as fh:
        fh.write(json.dumps(obj))


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
                 rng_mode: str = "legacy", seed: int = 0):
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
        return json.load(open(simname, "rb"))

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
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                return dist
        except Exception:
            pass
        return None

    def _is_writer(self) -> bool:
        d = self._dist()
        return d is None or d.get_rank() == 0

    # ------------------------------------------------------------------ the MC itself
    def _controller_rows(self, algoname, training_noise):
        """Controller list of one algorithm: keyed by str(training_noise), lbfgs by str(Nspin)
        (mcsim.py:427-441)."""
        key = str(self.Nspin) if algoname == "lbfgs" else str(training_noise)
        return self.controllers[algoname][key]["controller"]

    def _level_fidelities(self, ctrl: np.ndarray) -> np.ndarray:
        """(C_valid, K) fidelities of one noise level, sigma already set on the noise model."""
        nvalid = ctrl.shape[0]
        if self.rng_mode == "philox":
            return self._level_fidelities_philox(ctrl)
        draws = self.noise_model.draw_samples(nvalid, self.bootreps)     # full stream on every rank
        d = self._dist()
        if d is None:
            return np.asarray(self.noise_model.fidelity_from_draws(ctrl, draws))
        from .sharding import ShardedMC
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        sh = ShardedMC(self._sharded_compute, device=dev)
        lo, hi = sh.local_slice(nvalid)
        local = draws[lo:hi]
        if dev is not None:
            local = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
        res = sh.run_level(ctrl, local, self.Nspin, self.inspin, self.outspin, num_controllers=nvalid)
        return res.fid.cpu().numpy()

    def _level_fidelities_philox(self, ctrl: np.ndarray) -> np.ndarray:
        """Device-generated draws: element ((c*K + k)*N + i)*3 + slot of the level's block of the Philox stream,
        so the result does not depend on how the controllers are sharded.  Each rank generates its own slice."""
        from . import backend
        nvalid = ctrl.shape[0]
        per_ctrl = self.bootreps * self.Nspin * 3
        sigma = float(self.noise_model.rng.args.get("scale", self.noise_model.noise))
        base = self._philox_offset
        self._philox_offset += nvalid * per_ctrl
        d = self._dist()
        if d is None:
            draws = backend.philox_normal((nvalid, self.bootreps, self.Nspin, 3), self.seed, scale=sigma,
                                          offset=base, device=self.noise_model.device, as_torch=True)
            return self.noise_model.fidelity_from_draws(ctrl, draws).cpu().numpy()
        from .sharding import ShardedMC
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        sh = ShardedMC(self._sharded_compute, device=dev)
        lo, hi = sh.local_slice(nvalid)
        local = backend.philox_normal((hi - lo, self.bootreps, self.Nspin, 3), self.seed, scale=sigma,
                                      offset=base + lo * per_ctrl, device=dev, as_torch=True)
        res = sh.run_level(ctrl, local, self.Nspin, self.inspin, self.outspin, num_controllers=nvalid)
        return res.fid.cpu().numpy()

    def _sharded_compute(self, ctrl, draws, nspin, inspin, outspin, **kw):
        return self.noise_model.fidelity_from_draws(ctrl, draws)

    def get_algo_fid_dist(self, algoname: str, allalgoallfids: dict, noises, training_noise):
        """(L, C, K) fidelity tensor of one algorithm, stored into `allalgoallfids` and dumped (whole
        dict) to the `.mc` file - mcsim.py:422-460."""
        noises = np.asarray(noises)
        allfids = np.zeros((noises.size, self.numcontrollers, self.bootreps))
        for j, noise in enumerate(_progress(noises[:])):
            self.noise_model.rng(scale=noise)            # sets sigma_sim AND burns one draw (mcsim.py:425)
            self._say(algoname, training_noise)
            rows = self._controller_rows(algoname, training_noise)
            nvalid = min(len(rows), self.numcontrollers)
            if nvalid < self.numcontrollers:
                allfids[j, nvalid:] = np.nan             # padded controllers (mcsim.py:442-443)
            if nvalid and self.bootreps:
                ctrl = np.asarray(rows[:nvalid], dtype=np.float64)
                allfids[j, :nvalid] = self._level_fidelities(ctrl)
            # the reference leaves the last visited controller on the instance (mcsim.py:445)
            self.controller = rows[self.numcontrollers - 1] if len(rows) >= self.numcontrollers else np.nan
        allalgoallfids[algoname] = allfids.tolist()
        if self._is_writer():
            self._dump_mc(allalgoallfids, self.get_mcname(training_noise, noises))
        return allalgoallfids

    def _dump_mc(self, simdict: dict, path: str) -> None:
        """`json.dump(simdict, open(path, "w"))` (mcsim.py:459) - the whole dict after every algorithm, as the
        reference does - but each algorithm's tensor is serialised only once: the text is kept and reused by the
        later dumps of the same dict (a paper-scale tensor is 25 MB of JSON; the reference re-encodes all of them
        every time)."""
        cache = self.__dict__.setdefault("_mc_json_text", {})
        parts = []
        for algo, tensor in simdict.items():
            hit = cache.get(algo)
            if hit is None or hit[0] is not tensor:
                hit = (tensor, json.dumps(tensor))
                cache[algo] = hit
            parts.append(json.dumps(algo) + ": " + hit[1])
        with open(path, "w") as fh:
            fh.write("{" + ", ".join(parts) + "}")

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

    def metrics_for_tensor(self, dists_tensor) -> dict:
        """{metric(+suffix): [L][C]} of one algorithm's (L, C, K) tensor (mcsim.py:480-500) on the GPU."""
        T = np.asarray(dists_tensor, dtype=np.float64)
        eps = compute_dkw_error(self.alpha, self.bootreps)
        per_level = [rim_metrics.metric_table(T[lvl], dkw_eps=eps) for lvl in range(T.shape[0])]
        out = {}
        for name in METRIC_NAMES:
            for suffix in ("", " upper", " lower"):
                out[name + suffix] = [lvl[suffix][name] for lvl in per_level]
        return out

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
        allalgos = {algo: self.metrics_for_tensor(algofiddists[algo]) for algo in self.algos}
        if self._is_writer():
            _json_write(allalgos, path)
        return allalgos

    # ------------------------------------------------------------------ second caller of the kernel
    def get_rims(self, cont, noises=None):
        """`NStochOpt.get_rims` (gen_fig_8_arim_fcall_scaling.py:121-132) batched: for every noise level one
        burned draw, K noisy evaluations of `cont`, returns 1 - mean fidelity per level."""
        noises = self.noises if noises is None else noises
        x = np.asarray(cont, dtype=np.float64).reshape(1, -1)
        rims = np.zeros(len(noises))
        for i, nlvl in enumerate(noises):
            self.noise_model.rng(scale=nlvl)
            fids = self.noise_model.fidelity_batch(x, self.bootreps, ham_noisy=True)
            rims[i] = rim_metrics.backend.reduce_metrics(fids, q_thresholds=())["rim1"][0, 0]
        return rims

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
        _json_write(self.controllers, self.get_controller_name)

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
            _json_write(mine_f, fid_path)
            _json_write(mine_m, met_path)
        self._say("files successfully merged")

    @staticmethod
    def get_ranks(array):
        order = np.argsort(array)
        ranks = np.zeros_like(order)
        ranks[order] = np.arange(len(order))
        return ranks
