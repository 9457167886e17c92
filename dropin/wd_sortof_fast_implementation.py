"""`import wd_sortof_fast_implementation` shim -> robchar_amd RIM metrics (GPU reductions)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import importlib as _il
_r = _il.import_module("code-robchar_amd.rim_metrics")
wd_from_ideal, wd_from_ideal_zero, RIM_p = _r.wd_from_ideal, _r.wd_from_ideal_zero, _r.RIM_p
compute_dkw_error, dkw_ecdf_bounds = _r.compute_dkw_error, _r.dkw_ecdf_bounds
