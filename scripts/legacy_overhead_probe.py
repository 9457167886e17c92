#!/usr/bin/env python3
"""Wall time of one `legacy_normal_periods` call against its size: the fixed cost of a call (1000 normals: no jump), one
sub-stream, a dozen, a paper-scale algorithm.  Development aid."""
import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
for n in (1000, 100000, 1500000, 16500000):
    ts=[]
    for rep in range(8):
        np.random.seed(1); torch.cuda.synchronize(); t0=time.perf_counter()
        out = be.legacy_normal_periods(1, n, 0, np.array([0.05])); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print(n, "median %.3f ms min %.3f ms" % (1e3*sorted(ts)[len(ts)//2], 1e3*min(ts)))
