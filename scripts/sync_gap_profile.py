#!/usr/bin/env python3
"""Per-launch kernel time of the headline kernel right after a synchronisation gap (what the timed region of a 20-step
bench run sees): steady stream -> torch.cuda.synchronize() [+ sleep] -> 40 launches, one HIP-event bracket per 2 launches."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
N, C, K = 7, 100, 10000
rng = np.random.default_rng(0)
ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
ct = torch.from_numpy(ctrl).cuda()
draws = [torch.from_numpy(0.05 * rng.standard_normal((C, K, N, 3))).cuda() for _ in range(3)]
fid = torch.empty((C, K), dtype=torch.float64, device="cuda")
def launch(i): be.mc_fidelity(ct, draws[i % 3], N, 0, N - 1, out=fid)
for gap in (0.0, 0.0002, 0.005, 0.1):
    rows = []
    for rep in range(5):
        for i in range(3000): launch(i)                       # ~0.16 s of steady load
        torch.cuda.synchronize()
        if gap: time.sleep(gap)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        for j in range(20):
            ev[j].record(); launch(2 * j); launch(2 * j + 1)
        ev[20].record()
        torch.cuda.synchronize()
        rows.append([ev[j].elapsed_time(ev[j + 1]) / 2 * 1e3 for j in range(20)])
    m = np.median(np.array(rows), axis=0)
    print(f"gap {gap*1e3:.1f} ms: us per launch, pairs 0..19:", " ".join(f"{v:.1f}" for v in m), f"| mean of the first 20 launches {m[:10].mean():.1f}")
# steady state for reference: 2000 launches in one bracket
for i in range(3000): launch(i)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(2000): launch(i)
b.record(); torch.cuda.synchronize()
print(f"steady: {a.elapsed_time(b) / 2000 * 1e3:.1f} us per launch")
