"""robchar_amd - MI355X-native Monte-Carlo robustness characterisation (the RobChar hot path).

Host-side mirror of the reference's interface for the MC path only (names resolve lazily; `import robchar_amd` at the
repo root aliases this package, whose directory name is not a Python identifier):

    MCDataSim, DeviceFids                                        mc_data_sim.py   (reference mcsim.py:200-510)
    noise_function, noise_model_base, structured_perturbation,
    directional_perturbation                                     noise.py         (reference noise_model.py)
    wd_from_ideal, wd_from_ideal_zero, RIM_p, compute_dkw_error,
    dkw_ecdf_bounds, metric_table                                rim_metrics.py   (reference wd_sortof_fast_implementation.py,
                                                                                   mcsim.py:144-183)
    ExperimentNamer, DirectoryDoesNotExistError                  naming.py        (reference noise_analysis.py:33-61)
    backend, cache_io, sharding                                  C-ABI wrappers, cache files, controller partition

Compute goes through librobchar_hip.so (include/robchar_hip.h) - see backend.py; there is no CPU fallback.
"""
import importlib as _importlib

__version__ = "0.2.0"

_EXPORTS = {
    "MCDataSim": "mc_data_sim", "DeviceFids": "mc_data_sim",
    "noise_function": "noise", "noise_model_base": "noise", "structured_perturbation": "noise",
    "directional_perturbation": "noise",
    "wd_from_ideal": "rim_metrics", "wd_from_ideal_zero": "rim_metrics", "RIM_p": "rim_metrics",
    "compute_dkw_error": "rim_metrics", "dkw_ecdf_bounds": "rim_metrics", "metric_table": "rim_metrics", "get_cdf": "rim_metrics",
    "ExperimentNamer": "naming", "DirectoryDoesNotExistError": "naming",
}
_SUBMODULES = ("backend", "cache_io", "sharding", "mc_data_sim", "noise", "rim_metrics", "naming", "cli", "_lib")
__all__ = sorted(_EXPORTS) + ["backend", "cache_io", "sharding"]


def __getattr__(name):
    if name in _EXPORTS:
        return getattr(_importlib.import_module("." + _EXPORTS[name], __name__), name)
    if name in _SUBMODULES:
        return _importlib.import_module("." + name, __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return sorted(set(globals()) | set(__all__))
