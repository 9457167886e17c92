"""Command line entry of the MC path:  python -m robchar_amd.cli  (or  python robchar_cli.py).

Uses the flag names the reference defines for this purpose but never wires up (`get_mcsim_args`,
parse.py:112-145, with the common flags of parse.py:92-110): --exp_name --nspin --inspin --outspin --bootreps
--num_workers --training_noise --parallel --mc_max_noise --mc_noise_res, plus the `MCDataSim` kwargs that have
no flag there (--numcontrollers --filemarker --dkw_conflvl --topk) and the RNG / cache options of this implementation.
It builds `MCDataSim`, computes (or loads) the fidelity and metric caches and prints a per-algorithm summary.
"""
from __future__ import annotations

import argparse

import numpy as np


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser("Run a cachable Monte Carlo simulation on MI355X")
    p.add_argument("--exp_name", type=str, default="pipeline_nmplus2")
    p.add_argument("--nspin", type=int, default=5, help="Spin size/len of the qc system.")
    p.add_argument("--inspin", type=int, default=0, help="Input spin")
    p.add_argument("--outspin", type=int, default=2, help="Output spin")
    p.add_argument("--bootreps", type=int, default=100, help="Number of bootstrap repetitions.")
    p.add_argument("--num_workers", type=int, default=None, help="accepted for compatibility (unused on the GPU)")
    p.add_argument("--training_noise", type=str, default="0.1", help="Relevant if algo was trained on noise else pass None")
    p.add_argument("--parallel", type=bool, default=False, help="accepted for compatibility (unused on the GPU)")
    p.add_argument("--mc_max_noise", type=float, default=0.1, help="Maximum simulation noise")
    p.add_argument("--mc_noise_res", type=float, default=11, help="MC noise resolution/steps")
    p.add_argument("--numcontrollers", type=int, default=100)
    p.add_argument("--filemarker", type=str, default=None)
    p.add_argument("--dkw_conflvl", type=float, default=0.95)
    p.add_argument("--topk", type=int, default=100)
    p.add_argument("--algo", type=str, default=None, help="restrict to one algorithm (default: all in the controller file)")
    p.add_argument("--rng", choices=("legacy", "philox"), default="legacy",
                   help="legacy = numpy's global stream as the reference; philox = counter-based draws on the GPU")
    p.add_argument("--seed", type=int, default=None, help="np.random.seed / Philox key")
    p.add_argument("--cache_format", choices=("auto", "json", "npy", "none"), default="auto",
                   help="fidelity cache: json = the reference's .mc layout, npy = sidecars + index, none = metrics only")
    p.add_argument("--legacy_draws", choices=("device", "host"), default="device",
                   help="where numpy's legacy stream is produced (device: same stream and state, normals within 2 ulp)")
    return p


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    from .mc_data_sim import MCDataSim
    tn = None if args.training_noise in ("None", "none", "") else float(args.training_noise)
    noises = np.linspace(0, args.mc_max_noise, int(args.mc_noise_res))
    if args.seed is not None:
        np.random.seed(args.seed)
    sim = MCDataSim(experiment_name=args.exp_name, Nspin=args.nspin, inspin=args.inspin, outspin=args.outspin,
                    noises=noises, bootreps=args.bootreps, training_noise=tn, numcontrollers=args.numcontrollers,
                    parallel=args.parallel, num_workers=args.num_workers, dkw_conflvl=args.dkw_conflvl,
                    filemarker=args.filemarker, topk=args.topk, rng_mode=args.rng, seed=args.seed or 0,
                    cache_format=args.cache_format, legacy_draws=args.legacy_draws)
    if sim.controllers is None:
        print("no controller file:", sim.get_controller_name)
        return 2
    metrics = sim.get_metrics_dict(algoname=args.algo)
    name = r'$W(.,\delta(x-1))$'
    for algo, table in metrics.items():
        rim = np.array(table[name], dtype=float)             # (L, C)
        print(f"{algo:8s} mean RIM per sigma_sim: " + " ".join(f"{v:.4f}" for v in np.nanmean(rim, axis=1)))
    print("caches:", sim.get_mcname(), "(+m)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
