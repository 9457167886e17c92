"""`import noise_model` shim -> robchar_amd noise models (MI355X)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import importlib as _il
_n = _il.import_module("code-robchar_amd.noise")
noise_function = _n.noise_function
noise_model_base = _n.noise_model_base
structured_perturbation = _n.structured_perturbation
directional_perturbation = _n.directional_perturbation
