"""`import noise_analysis` shim: only the naming scheme of the MC path (controller collection is out of scope)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import importlib as _il
_n = _il.import_module("code-robchar_amd.naming")
ExperimentNamer, DirectoryDoesNotExistError = _n.ExperimentNamer, _n.DirectoryDoesNotExistError
