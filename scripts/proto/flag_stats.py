#!/usr/bin/env python3
"""Who leaves the one-step path of the mixed-precision eigenvalue route?  The kernel arithmetic compiled for the host (a "wave" is
one sample there, so the tile-level flag `extra_steps` becomes a per-sample flag) on the benchmark workloads of BASELINE configs 3
and 5: share of flagged samples / tiles, flagged lanes per flagged tile, and the distribution over controllers.  Verdict
(DESIGN.md 8 xii): at N = 10 XXZ 10.6 % of the samples flag, clustered by controller (46 of 100 controllers have > 90 % of their
tiles flagged) - per-sample deferral to a second launch has nothing to win.  (The host's fp32 starts lack the extra lock-step
sweeps a device lane gets from its slower neighbours: the host over-counts - device: 41 % of the N = 10 tiles, host: 54 %.)"""
import ctypes, os, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import robchar_oracle as orc
so = os.path.join(tempfile.gettempdir(), "libfs_flag_stats.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "flag_stats.cpp")], check=True)
lib = ctypes.CDLL(so)
def run(N, ctrl, draws, a, b, h0=None):
    C,K = draws.shape[:2]
    h0d = np.zeros(16) if h0 is None else np.concatenate([h0, np.zeros(16-N)])
    h0o = np.ones(16)
    fid = np.empty((C,K)); fl = np.empty((C,K), dtype=np.int32)
    roots = np.zeros((C,K), dtype=np.uint32); maxd = np.zeros((C,K)); lam = np.zeros((C,K,N))
    vp = ctypes.c_void_p
    lib.flags(N, vp(ctrl.ctypes.data), vp(h0d.ctypes.data), vp(h0o.ctypes.data), vp(draws.ctypes.data), ctypes.c_longlong(C), ctypes.c_longlong(K), a, b, vp(fid.ctypes.data), vp(fl.ctypes.data),
              vp(roots.ctypes.data), vp(maxd.ctypes.data), vp(lam.ctypes.data))
    run.last = (roots, maxd, lam)
    return fid, fl
def report(name, fl):
    C,K = fl.shape
    f = fl > 0
    iters = np.where(fl > 0, fl - 1, 0)                  # stepping iterations of the sample alone
    T = K//64
    tiles = f[:, :T*64].reshape(C, T, 64)
    per_tile = tiles.sum(axis=2)
    print(name, "samples flagged %.3f%%" % (100*f.mean()), "tiles flagged %.1f%%" % (100*(per_tile>0).mean()))
    h = np.bincount(per_tile[per_tile>0].ravel(), minlength=8)
    print("  flagged lanes per flagged tile: 1:%d 2:%d 3:%d 4-7:%d 8-15:%d 16+:%d" % (h[1],h[2],h[3],h[4:8].sum(),h[8:16].sum(),h[16:].sum()))
    it_tile = iters[:, :T*64].reshape(C, T, 64).max(axis=2)     # the tile runs until its slowest lane is done
    hi = np.bincount(it_tile[it_tile > 0].ravel(), minlength=6)
    print("  stepping iterations per flagged tile (max over its lanes): " + " ".join(f"{i}:{hi[i]}" for i in range(1, len(hi)) if hi[i]))
    pc = f.mean(axis=1)
    print("  per controller flagged-sample fraction quantiles:", np.round(np.quantile(pc,[0,.25,.5,.75,.9,.95,1]),4))
    tf = (per_tile>0).mean(axis=1)
    print("  per controller flagged-TILE fraction quantiles:", np.round(np.quantile(tf,[0,.25,.5,.75,.9,.95,1]),3), " controllers with >90%% tiles flagged: %d, with <5%%: %d" % ((tf>0.9).sum(), (tf<0.05).sum()))
def chains(N, fl, roots, maxd, lam):
    """Halley chains a flagged TILE runs per stepping iteration = the union of its lanes' wanted eigenvalues: the shipped rule
    (own step and own gap per eigenvalue, positions as the fp32 QL left them - different from lane to lane) against a rule on
    SORTED iterates (adjacent gaps only, the largest step for every eigenvalue - positions aligned across the lanes)."""
    C, K = fl.shape
    T = K // 64
    f = fl > 0
    bits = ((roots[..., None] >> np.arange(N)) & 1).astype(bool) & f[..., None]
    uni = bits[:, :T*64].reshape(C, T, 64, N).any(axis=2)[..., :N-1].sum(axis=2)          # (the last eigenvalue takes no step)
    ls = np.sort(lam, axis=2)
    gap = np.full(lam.shape, np.inf)
    dif = np.diff(ls, axis=2)
    gap[..., :-1] = dif
    gap[..., 1:] = np.minimum(gap[..., 1:], dif)
    g = np.maximum(gap - 2 * maxd[..., None], 0)
    want = ((N - 1) * maxd[..., None] ** 3 > 1e-14 * g * g) & f[..., None]
    uni2 = want[:, :T*64].reshape(C, T, 64, N).any(axis=2)[..., :N-1].sum(axis=2)
    tf = f[:, :T*64].reshape(C, T, 64).any(axis=2)
    per = bits[f][:, :N-1].sum(axis=1)
    per2 = want[f][:, :N-1].sum(axis=1)
    print("  chains per flagged SAMPLE: shipped rule %.2f, sorted rule %.2f; per flagged TILE (union over its lanes): shipped %.2f, sorted %.2f"
          % (per.mean(), per2.mean(), uni[tf].mean(), uni2[tf].mean()))

K=2000
for (N,cid,a,b,xxz) in ((7,3,0,6,False),(10,5,0,9,True)):
    rng = np.random.default_rng(20220714 + cid)
    ctrl = np.empty((100,N+1)); ctrl[:,:N]=rng.uniform(-10,10,(100,N)); ctrl[:,N]=rng.uniform(2,30,100)
    np.random.seed(12345); np.random.normal(scale=0.05)
    draws = 0.05*np.random.standard_normal((100,K,N,3))
    h0 = orc.xxz_delta(N) if xxz else None
    fid, fl = run(N, ctrl, draws, a, b, h0)
    want = orc.fidelity_eigh(ctrl[:3], draws[:3], N, a, b, h0_diag=h0)
    print("err", np.abs(fid[:3]-want).max())
    report("N=%d config %d"%(N,cid), fl)
    chains(N, fl, *run.last)
