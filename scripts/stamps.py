#!/usr/bin/env python3
"""Per-wave s_memtime stamps of the fidelity kernel (diagnostic build, ROBCHAR_HIP_LIB=scripts/ubench/lib_stamps.so)."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
lib = importlib.import_module("code-robchar_amd._lib").load()
N, C, K = int(os.environ.get("NN", "7")), 100, int(os.environ.get("KK", "10000"))
TPW = int(os.environ.get("TPW", "1"))
rng = np.random.default_rng(N)
ctrl = np.empty((C, N + 1)); ctrl[:, :N] = rng.uniform(-10, 10, (C, N)); ctrl[:, N] = rng.uniform(2, 30, C)
draws = torch.from_numpy(0.05 * rng.standard_normal((C, K, N, 3))).cuda()
ct = torch.from_numpy(ctrl).cuda()
out = torch.empty((C, K), dtype=torch.float64, device="cuda")
ntiles = (C * ((K + 63) // 64) + TPW - 1) // TPW
st = torch.zeros((ntiles, 8), dtype=torch.int64, device="cuda")
lib.rc_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
for _ in range(3):
    be.mc_fidelity(ct, draws, N, 0, N - 1, out=out)
torch.cuda.synchronize()
s = st.cpu().numpy()
life = s[:, 2] - s[:, 0]; load = s[:, 1] - s[:, 0]; comp = s[:, 2] - s[:, 1]
real = s[:, 3] / 100e6
print(f"shader clock from per-wave ticks/realtime: median {np.median(life / real) / 1e9:.3f} GHz (p10 {np.percentile(life/real,10)/1e9:.3f}, p90 {np.percentile(life/real,90)/1e9:.3f}); wave lifetime median {np.median(real)*1e6:.1f} us")
print("compute split (ticks, median): setup(gauge) %.0f | QL %.0f | weights+sincos+store %.0f" % (
    np.median(s[:, 6] - s[:, 1]), np.median(s[:, 7] - s[:, 6]), np.median(s[:, 2] - s[:, 7])))
h = len(s) // 2
print(f"second half of blocks: lifetime {np.median(life[h:]):.0f} load {np.median(load[h:]):.0f} compute {np.median(comp[h:]):.0f}")
span_ticks = s[:, 2].max() - s[:, 0].min()
span_real = (s[:, 3].max() - s[:, 3].min()) / 100e6
print(f"waves {ntiles}: lifetime ticks median {np.median(life):.0f} (load {np.median(load):.0f}, compute {np.median(comp):.0f}); "
      f"p10/p90 compute {np.percentile(comp,10):.0f}/{np.percentile(comp,90):.0f}")
print(f"kernel span {span_ticks} ticks; realtime span {span_real*1e6:.1f} us -> memtime clock ~ {span_ticks/span_real/1e9:.3f} GHz")
print(f"sum of wave lifetimes / span = {life.sum()/span_ticks:.1f} (avg resident waves chip-wide; /1024 SIMDs = {life.sum()/span_ticks/1024:.2f} per SIMD)")
if os.environ.get("DUMP"):
    os.makedirs(os.path.dirname(os.environ["DUMP"]), exist_ok=True)
    np.save(os.environ["DUMP"], s)
