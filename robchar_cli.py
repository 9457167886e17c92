#!/usr/bin/env python3
"""`python robchar_cli.py --exp_name ... --nspin 5 ...` - thin launcher of code-robchar_amd/cli.py."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
raise SystemExit(importlib.import_module("code-robchar_amd.cli").main())
