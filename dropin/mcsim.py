"""`import mcsim` shim -> robchar_amd MC driver (MI355X).  Only the MC-path names are provided."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import importlib as _il
_m = _il.import_module("code-robchar_amd.mc_data_sim")
_r = _il.import_module("code-robchar_amd.rim_metrics")
MCDataSim = _m.MCDataSim
ExperimentNamer = _m.ExperimentNamer
DirectoryDoesNotExistError = _m.DirectoryDoesNotExistError
wd_from_ideal, compute_dkw_error = _r.wd_from_ideal, _r.compute_dkw_error
# the metric callables of mcsim.py:144-183 (lazy like the reference's map objects; one GPU reduction launch behind each)
Q, wc_fids, std_fids, Q_fids, wd_from_ideal_fids, Q_partial = _r.Q, _r.wc_fids, _r.std_fids, _r.Q_fids, _r.wd_from_ideal_fids, _r.Q_partial
__metric_name_to_metric__ = _r.__metric_name_to_metric__
get_cdf = _r.get_cdf            # mcsim.py:42-47
