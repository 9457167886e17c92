import importlib, numpy as np, sys, os
sys.path.insert(0, "/root/repo")
be = importlib.import_module("code-robchar_amd.backend")
from oracle import robchar_oracle as orc
for (N, xxz) in ((5, False), (7, False), (10, True)):
    rng = np.random.default_rng(N)
    C,K=100,10000
    ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
    d = 0.05*rng.standard_normal((C,K,N,3))
    be.polish_tiles(reset=True); be.general_path_tiles(reset=True)
    be.mc_fidelity(ctrl, d, N, 0, N-1, h0_diag=orc.xxz_delta(N) if xxz else None)
    print("   N", N, "polish tiles", be.polish_tiles(), "general", be.general_path_tiles(), "of 15700")
