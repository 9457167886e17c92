#!/usr/bin/env python3
"""Concurrency stress (round 5): T Python threads, each on its own torch stream, mixing the enqueue-only entries (chain, ring
with its per-stream repair list, reductions with the row sort, Philox) with the blocking host-buffer entries and the
single-process multi-device entry, ITER rounds each; every result must equal the single-threaded reference bit for bit
(the ring route included since its repair kernel runs per-lane sweeps: before, it was reproducible to rounding only - its
listed samples are packed into waves in ARRIVAL order, and a wave's sweep count was shared by its lanes).
usage: python3 scripts/thread_stress.py [T=6] [ITER=150]"""
import importlib, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ITER = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(11)
work = []
for t in range(T):
    N = (5, 7, 10, 12, 6, 9)[t % 6]; C, K = 12 + t, 700 + 37 * t
    c = np.empty((C, N + 1)); c[:, :N] = rng.uniform(-10, 10, (C, N)); c[:, N] = rng.uniform(2, 30, C)
    d = 0.05 * rng.standard_normal((C, K, N, 3))
    work.append((N, C, K, c, d, torch.from_numpy(c).cuda(), torch.from_numpy(d).cuda()))
ref = []
for (N, C, K, c, d, ct, dt) in work:                      # single-threaded references
    f_chain = be.mc_fidelity(ct, dt, N, 0, N - 1).clone()
    f_mid = be.mc_fidelity(ct, dt, N, 1, N // 2).clone()
    f_ring = be.mc_fidelity(ct, dt, N, 0, N // 2, ring=True).clone()
    red = be.reduce_metrics(f_chain, dkw_eps=0.01, want_sorted=True)
    f_ph = be.mc_fidelity_philox(ct, K, N, 0, N - 1, 99, sigma=0.05).clone()
    f_host = np.array(be.mc_fidelity(c, d, N, 0, N - 1))
    ref.append((f_chain, f_mid, f_ring, {k: (v.clone() if hasattr(v, "clone") else v) for k, v in red.items()}, f_ph, f_host))
torch.cuda.synchronize()
errors = []
ring_worst = [0.0]
def worker(t):
    try:
        N, C, K, c, d, ct, dt = work[t]
        r = ref[t]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(ITER):
                a = be.mc_fidelity(ct, dt, N, 0, N - 1)
                b = be.mc_fidelity(ct, dt, N, 1, N // 2)
                g = be.mc_fidelity(ct, dt, N, 0, N // 2, ring=True)
                red = be.reduce_metrics(a, dkw_eps=0.01, want_sorted=True)
                p = be.mc_fidelity_philox(ct, K, N, 0, N - 1, 99, sigma=0.05)
                s.synchronize()
                ok = (torch.equal(a, r[0]) and torch.equal(b, r[1]) and torch.equal(g, r[2]) and torch.equal(p, r[4])
                      and torch.equal(red["sorted"], r[3]["sorted"]) and all(torch.equal(red["rim1"][v], r[3]["rim1"][v]) for v in range(3)))
                ring_worst[0] = max(ring_worst[0], float((g - r[2]).abs().max()))
                if not ok:
                    names = ("chain", "mid", "ring", "philox", "sorted", "rim1")
                    pairs = ((a, r[0]), (b, r[1]), (g, r[2]), (p, r[4]), (red["sorted"], r[3]["sorted"]), (red["rim1"][0], r[3]["rim1"][0]))
                    bad = {n: float((x - y).abs().max()) for n, (x, y) in zip(names, pairs) if not torch.equal(x, y)}
                    ring_worst[0] = max(ring_worst[0], bad.get("ring", 0.0))
                    errors.append((t, i, "async results differ", N, bad)); return
                if i % 5 == t % 5:                        # the blocking entries share one library stream / workspace per device
                    h = np.array(be.mc_fidelity(c, d, N, 0, N - 1))
                    sh = np.array(be.mc_fidelity_sharded(c, d, N, 0, N - 1, devices=[0]))
                    if not (np.array_equal(h, r[5]) and np.array_equal(sh, r[5])):
                        errors.append((t, i, "blocking results differ")); return
        be.release_stream(s)
    except Exception as e:                                # noqa: BLE001
        errors.append((t, -1, repr(e)))
th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for x in th: x.start()
for x in th: x.join()
print(f"{T} threads x {ITER} rounds: errors {errors[:5]}; ring route's worst deviation from the single-threaded run {ring_worst[0]:.1e}")
assert not errors
print("ok")
