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
