#!/usr/bin/env python3
"""Device / host memory over many calls of every kind of entry (round 5): free device memory (hipMemGetInfo through torch) and
the process's resident set before and after `ITER` rounds; a leak of one buffer per call would show as hundreds of MB.
usage: python3 scripts/leak_probe.py [ITER=300]"""
import importlib, os, sys, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(3)
def ctrl(C, N):
    x = np.empty((C, N + 1)); x[:, :N] = rng.uniform(-10, 10, (C, N)); x[:, N] = rng.uniform(2, 30, C); return x
N, C, K = 7, 40, 1000
c_np = ctrl(C, N); d_np = 0.05 * rng.standard_normal((C, K, N, 3))
c_t, d_t = torch.from_numpy(c_np).cuda(), torch.from_numpy(d_np).cuda()
out = torch.empty((C, K), dtype=torch.float64, device="cuda")
def rss(): return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2**20      # CURRENT resident set
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
def one_round(i):
    be.mc_fidelity(c_t, d_t, N, 0, N - 1, out=out)
    be.mc_fidelity(c_t, d_t, N, 0, 3, out=out)
    be.mc_fidelity(c_t, d_t, N, 0, 3, out=out, ring=True)
    be.reduce_metrics(out, dkw_eps=0.01, want_sorted=(i % 2 == 0))
    be.mc_fidelity_philox(c_t, K, N, 0, N - 1, 7 + i, sigma=0.05, out=out)
    be.philox_normal((C, K, N, 3), seed=i, scale=0.05, as_torch=True)
    be.mc_fidelity(c_np, d_np, N, 0, N - 1)                                   # blocking entry, host buffers
    be.mc_fidelity_sharded(c_np, d_np, N, 0, N - 1, devices=[0])
    be.mc_metrics_sharded(c_np, K, N, 0, N - 1, seed=3, sigma=0.05, devices=[0])
    if i % 10 == 0:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            be.mc_fidelity(c_t, d_t, N, 0, 3, out=out, ring=True)             # a fresh stream running the ring route
        s.synchronize(); be.release_stream(s)
for i in range(20): one_round(i)                                              # allocator pools, workspaces, lazy code loading
torch.cuda.empty_cache()
f0, r0 = free(), rss()
trace = []
for i in range(ITER):
    one_round(i)
    if (i + 1) % max(1, ITER // 6) == 0: trace.append(round(rss()))
torch.cuda.empty_cache()
f1, r1 = free(), rss()
print('resident set along the way (MiB):', trace)
with be.RcclComm(devices=[0]) as comm:
    for i in range(max(10, ITER // 10)):
        be.mc_metrics_gathered(comm, c_np, K, N, 0, N - 1, seed=3, sigma=0.05, want_fid=True)
torch.cuda.empty_cache()
f2, r2 = free(), rss()
print(f"{ITER} rounds of 10 entry kinds: free device memory {f0:.0f} -> {f1:.0f} MiB ({f1 - f0:+.0f}), resident set {r0:.0f} -> {r1:.0f} MiB ({r1 - r0:+.0f})")
print(f"+ {max(10, ITER // 10)} gathered calls on a one-device communicator: free {f2:.0f} MiB ({f2 - f1:+.0f}), resident set {r2:.0f} MiB ({r2 - r1:+.0f})")
assert f1 - f0 > -64 and f2 - f1 > -64, "device memory is leaking"
print("ok")
