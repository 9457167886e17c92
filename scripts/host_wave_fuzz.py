#!/usr/bin/env python3
"""The adversarial configurations of scripts/fuzz_parity.py through the LOCK-STEP host emulation of a wave
(tests/host/host_wave.cpp: every lane a host thread, every wave-level vote a barrier), tile by tile, against the oracle - a
fuzz campaign for what wave-uniform decisions do to the other lanes of a tile, without a GPU (the one-sample-per-wave host
build of host_fuzz_guard.py cannot see those; the round-4 GPU campaign found one: DESIGN.md 3).  N = 3 .. 14, 16.
usage: host_wave_fuzz.py [first_seed] [nseeds] [ncfg] [max tiles per configuration]"""
import ctypes, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
from oracle import robchar_oracle as orc
from host_fuzz_guard import configs
P, PI = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
so = os.path.join(tempfile.mkdtemp(prefix="rc_hostwave_"), "lib.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-pthread", "-o", so,
                os.path.join(ROOT, "tests", "host", "host_wave.cpp")], check=True)
lib = ctypes.CDLL(so)


def tile(ctrl, draws, N, a, b, mode, h0d):
    nk = draws.shape[0]
    h0 = np.zeros(32)
    if h0d is not None:
        h0[:N] = h0d
    h0o = np.ones(32)
    fid, rep, ex = np.empty(nk), np.zeros(nk, dtype=np.int32), np.zeros(nk, dtype=np.int32)
    ctrl, draws = np.ascontiguousarray(ctrl), np.ascontiguousarray(draws)
    assert lib.rc_host_wave_chain_tile(N, ctrl.ctypes.data_as(P), h0.ctypes.data_as(P), h0o.ctypes.data_as(P), draws.ctypes.data_as(P),
                                       nk, a, b, mode, fid.ctypes.data_as(P), rep.ctypes.data_as(PI), ex.ctypes.data_as(PI)) == 0
    return fid, rep, ex


def ring_tile(ctrl, draws, N, a, b, route, h0d):
    nk = draws.shape[0]
    h0 = np.zeros(32)
    if h0d is not None:
        h0[:N] = h0d
    h0o = np.ones(32)
    fid, rep, ex = np.empty(nk), np.zeros(nk, dtype=np.int32), np.zeros(nk, dtype=np.int32)
    ctrl, draws = np.ascontiguousarray(ctrl), np.ascontiguousarray(draws)
    assert lib.rc_host_wave_ring_tile(N, ctrl.ctypes.data_as(P), h0.ctypes.data_as(P), h0o.ctypes.data_as(P), draws.ctypes.data_as(P),
                                      nk, a, b, route, fid.ctypes.data_as(P), rep.ctypes.data_as(PI), ex.ctypes.data_as(PI)) == 0
    return fid, rep, ex


first, nseeds, ncfg, maxt = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 7000), (2, 10), (3, 150), (4, 3)))
worst = {}
ntiles = nrep = nstep = 0
t0 = time.time()
for seed in range(first, first + nseeds):
    for meta, ctrl, draws, h0 in configs(seed, ncfg):
        N, a, b = meta["N"], meta["a"], meta["b"]
        if N < 3 or N == 15:
            continue
        want = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0)
        C, K = draws.shape[:2]
        want_r = orc.fidelity_eigh(ctrl, draws, N, a, b, h0_diag=h0, ring=True) if N <= 10 else None
        for c in range(C):
            for t in range(min(maxt, (K + 63) // 64)):
                sl = slice(64 * t, min(64 * t + 64, K))
                for name, mode in (("auto", 2 if {a, b} == {0, N - 1} else 1), ("tridiag_adj", 1), ("tridiag_ql", 0)):
                    fid, rep, ex = tile(ctrl[c], draws[c, sl], N, a, b, mode, h0)
                    e = float(np.abs(fid - want[c, sl]).max())
                    if e > worst.get(name, (0,))[0]:
                        worst[name] = (e, dict(seed=seed, it=meta["it"], N=N, a=a, b=b, c=c, tile=t, amp=meta.get("amp"), T=float(ctrl[c, N])))
                    if name == "auto" and want_r is not None:
                        for rname, route in (("ring:auto", 0), ("ring:ring_hh", 1)):
                            fr, rr, er = ring_tile(ctrl[c], draws[c, sl], N, a, b, route, h0)
                            e = float(np.abs(fr - want_r[c, sl]).max())
                            if e > worst.get(rname, (0,))[0]:
                                worst[rname] = (e, dict(seed=seed, it=meta["it"], N=N, a=a, b=b, c=c, tile=t, amp=meta.get("amp"), T=float(ctrl[c, N])))
                    if name == "auto":
                        ntiles += 1
                        nrep += int((rep > 0).any())
                        nstep += int((ex > 0).any())
print(f"{ntiles} tiles of {nseeds} x {ncfg} configurations in {time.time() - t0:.0f} s; tiles off the one-step path {nstep}, with a repaired lane {nrep}")
for k, v in worst.items():
    print(f"{k:12s} worst |dF| = {v[0]:.2e} at {v[1]}")
