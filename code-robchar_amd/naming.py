"""On-disk naming of controller files and MC caches (the cached-results layout of the drop-in boundary).

Mirrors `ExperimentNamer` (noise_analysis.py:33-49) and `DirectoryDoesNotExistError` (:58-61): the
controller file of an experiment is ``<global_dir>/<experiment>/ppo_spin_{N}_{in}-{out}_c_{C}`` and the
experiment directory is created on first use.  As in the reference, calling `home()` replaces the bound
method by the resulting string on that instance (single-use object).
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class ExperimentNamer:
    experiment_name: str = "alpha"
    Nspin: int = 5
    inspin: int = 0
    outspin: int = 2
    numcontrollers: int = 100
    global_dir: str = "experiments"

    def home(self):
        path = f"{self.global_dir}/{self.experiment_name}"
        if not os.path.exists(path):
            try:
                os.mkdir(path)
            except FileExistsError:      # another rank of the process group created it in between
                pass
        self.home = path            # the reference rebinds the attribute the same way (noise_analysis.py:43)
        return path

    def __call__(self):
        return f"{self.home()}/ppo_spin_{self.Nspin}_{self.inspin}-{self.outspin}_c_{self.numcontrollers}"


class DirectoryDoesNotExistError(Exception):
    def __init__(self, global_exp_path):
        self.message = "Directory not found in {}!".format(global_exp_path)
        super().__init__(self.message)
