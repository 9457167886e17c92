#!/usr/bin/env python3
"""The reduction of an (R, K) fidelity slab to its metric rows (`reduce_packed`: RIM_1, std, min, Q thresholds - sort-based)
over row lengths and row counts: time per call and per value; looks for cliffs between the three kernel routes (wave per row,
row in LDS, global merge sort).  Development aid."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
dev = torch.device("cuda", 0)
def timed(f, reps=30):
    for _ in range(3): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3
g = torch.Generator(device=dev); g.manual_seed(1)
for R in (100, 1000, 11000):
    for K in (16, 64, 100, 256, 1000, 2048, 2049, 4096, 4097, 8192, 8193, 10000, 10240, 10241, 16384, 16385, 50000, 100000, 300000):
        if R * K > 4e8: continue
        f = torch.rand((R, K), dtype=torch.float64, device=dev, generator=g)
        out = torch.empty((be.PACKED_ROWS, R), dtype=torch.float64, device=dev)
        # overlapped=True: the route of a pipelined caller (rc_reduce_f64_async); False: RC_REDUCE_STANDALONE - what MCDataSim,
        # the blocking entry and the multi-device entries use (round 5); they differ for 8192 < K <= 10240 only
        t = timed(lambda: be.reduce_packed(f, 0.0043, out=out, overlapped=True))
        t2 = timed(lambda: be.reduce_packed(f, 0.0043, out=out, overlapped=False))
        print(f"R={R:6d} K={K:7d}: {t:9.1f} us  {t * 1e3 / (R * K):7.3f} ns per value  {R * K * 8 / t / 1e6:6.2f} TB/s read"
              f"   | standalone hint: {t2:9.1f} us  {R * K * 8 / t2 / 1e6:6.2f} TB/s")
        del f
