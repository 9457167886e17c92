#!/usr/bin/env python3
"""Round 5, review item 3: would a per-CONTROLLER route class pay?  Host statistics on the benchmark workloads (kernel arithmetic compiled
for the host, scripts/proto/flag_stats.cpp): which controllers hold the tiles that leave the one-step path ("hard" = an unperturbed eigenvalue
pair closer than G), how many of a hard controller's tiles are flagged today, and how many would pass a rule that lists the close pairs per
controller (sorted starts, Newton on the listed eigenvalues only).  Verdict: NOTEBOOK.md 12 (c) - dropped."""
import os, sys, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

src = open(os.path.join(HERE, "flag_stats.py")).read().split("K=2000")[0]
ns = {"__file__": os.path.join(HERE, "flag_stats.py")}
exec(compile(src, "fs", "exec"), ns)
run, orc = ns["run"], ns["orc"]
def unperturbed(ctrl, N, h0):
    lam0 = []
    for x in ctrl:
        H = np.diag(x[:N] + (h0 if h0 is not None else 0)) + np.diag(np.ones(N-1),1) + np.diag(np.ones(N-1),-1)
        lam0.append(np.linalg.eigvalsh(H))
    return np.array(lam0)
def analyse(name, N, ctrl, draws, a, b, h0):
    C,K = draws.shape[:2]
    fid, fl = run(N, ctrl, draws, a, b, h0)
    roots, maxd, lam = run.last
    T = K//64
    f = fl > 0
    tiles_f = f[:, :T*64].reshape(C,T,64).any(axis=2)
    print(name, "tiles flagged %.1f%%" % (100*tiles_f.mean()))
    lam0 = unperturbed(ctrl, N, h0)
    gap0 = np.diff(lam0, axis=1)                      # (C, N-1)
    ls = np.sort(lam, axis=2)
    # for unflagged samples the hook is not called: lam = 0 -> recompute sorted eigenvalues with numpy for all samples
    Hd = ctrl[:, None, :N] + (h0 if h0 is not None else 0) + draws[..., 0]
    e = np.abs(1.0 + draws[..., 1:, 1] + 1j*draws[..., 1:, 2])
    Hm = np.zeros((C,K,N,N)); idx=np.arange(N)
    Hm[..., idx, idx] = Hd
    Hm[..., idx[:-1], idx[1:]] = e; Hm[..., idx[1:], idx[:-1]] = e
    ls = np.linalg.eigvalsh(Hm)
    dif = np.diff(ls, axis=2)                         # (C,K,N-1) perturbed adjacent gaps
    scale = np.abs(Hd).max(axis=2)
    unc = 3*1.19e-7*np.maximum(scale, 1)
    # start error model: maxd measured only for flagged; for all samples assume typical step = per-sample fp32 error ~ 2.5e-7*scale*sqrt
    md = np.where(f, maxd, 0.0)
    # estimate typical maxd for unflagged from flagged distribution per controller median (fallback 3e-6)
    med = np.median(maxd[f]) if f.any() else 3e-6
    md = np.where(f, maxd, med)
    for G in (0.1, 0.2, 0.3, 0.5):
        P = gap0 < G                                   # (C, N-1) listed pairs
        hard = P.any(axis=1)
        g_rest = np.where(P[:, None, :], np.inf, dif).min(axis=2)
        ok_rest = md**3 <= 1e-14 * np.maximum(g_rest - unc, 0)**2
        g_pair = np.where(P[:, None, :], dif, np.inf).min(axis=2)
        # listed roots: after Aberth step error e1 ~ md^3 (N-1)/g_pair^2 ; Newton -> e1^2/g_pair <= 1e-14 and e1 < g_pair/10
        e1 = md**3*(N-1)/np.maximum(g_pair-unc,1e-30)**2
        ok_pair = (e1**2/np.maximum(g_pair,1e-30) <= 1e-14) & (e1 < 0.1*g_pair) | ~np.isfinite(g_pair)
        ok = ok_rest & ok_pair
        t_ok = ok[:, :T*64].reshape(C,T,64).all(axis=2)
        hard_t = np.repeat(hard[:,None], T, axis=1)
        nlisted = P.sum(axis=1)
        print(f"  G={G}: hard controllers {hard.sum()}, listed pairs per hard ctrl {nlisted[hard].mean():.2f}; of flagged tiles: in hard ctrl {100*(tiles_f&hard_t).sum()/tiles_f.sum():.1f}%, "
              f"hard&flagged tiles passing new rule {100*(tiles_f&hard_t&t_ok).sum()/max(1,(tiles_f&hard_t).sum()):.1f}%; hard tiles UNflagged today {100*(hard_t&~tiles_f).sum()/max(1,hard_t.sum()):.1f}% of hard tiles")
K=1280
for (N,cid,a,b,xxz) in ((7,3,0,6,False),(10,5,0,9,True)):
    rng = np.random.default_rng(20220714 + cid)
    ctrl = np.empty((100,N+1)); ctrl[:,:N]=rng.uniform(-10,10,(100,N)); ctrl[:,N]=rng.uniform(2,30,100)
    np.random.seed(12345); np.random.normal(scale=0.05)
    draws = 0.05*np.random.standard_normal((100,K,N,3))
    h0 = orc.xxz_delta(N) if xxz else None
    analyse("N=%d config %d uniform"%(N,cid), N, ctrl, draws, a, b, h0)
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'lbfgs_n7.npz'))
rows = z['ctrl_0-6']; ctrl = np.ascontiguousarray(rows[np.arange(100)%rows.shape[0]])
draws = 0.05*np.random.default_rng(5).standard_normal((100,K,7,3))
analyse("N=7 shipped", 7, ctrl, draws, 0, 6, None)
hz = np.load(os.path.join(ROOT, 'tests', 'golden', 'highfid.npz'))
draws = 0.05*np.random.default_rng(6).standard_normal((100,K,10,3))
analyse("N=10 XXZ constructed", 10, np.ascontiguousarray(hz['c5_ctrl']), draws, 0, 9, hz['c5_h0_diag'])
