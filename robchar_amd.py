"""Import alias: ``import robchar_amd`` -> the package in ``code-robchar_amd/`` (a directory name that is
not a Python identifier, so it is loaded through importlib)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("code-robchar_amd")
sys.modules[__name__] = _pkg
