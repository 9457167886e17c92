#!/usr/bin/env python3
"""Timeline of ONE launch of the fidelity kernel, per compute unit, from the per-wave stamps of a diagnostic build
(`scripts/build_variant.sh stamps -DRC_STAMPS -DRC_DEV_FEW_N`, then scripts/stamps.py with DUMP=...): how long the pipeline takes to
fill (the first round of waves all ask for their draws at once), how long a wave lives alone and in a crowd, how long the drain is.
s_memtime counts per CU (the 256 counters are not synchronised), so tiles are grouped by counter base: a group of 58 ... 64 tiles
(15 700 / 256 = 61.3) is ONE compute unit; groups whose bases happen to lie close together are skipped.  Host-only: reads the .npy.

usage: python3 scripts/stamps_timeline.py gpurun_out/<label>/stamps_n7.npy [--ghz 1.95]
columns of the dump (per tile): 0 begin, 1 staged, 2 end (s_memtime ticks), 3 realtime ticks of the wave (100 MHz), 4 ctrl row read,
5 first staging phase landed, 6 QL starts, 7 QL done
"""
import sys
import numpy as np

s = np.load(sys.argv[1]).astype(np.int64)
life_us = s[:, 3] / 100.0
tick = float(np.median((s[:, 2] - s[:, 0]) / np.maximum(life_us, 1e-9)))          # ticks per us, from the waves' own realtime
if "--ghz" in sys.argv:
    tick = float(sys.argv[sys.argv.index("--ghz") + 1]) * 1e3
idx = np.argsort(s[:, 0])
cut = np.where(np.diff(s[idx, 0]) > 300000)[0]
bounds = np.concatenate([[0], cut + 1, [len(idx)]])
cus = [idx[bounds[i]:bounds[i + 1]] for i in range(len(bounds) - 1) if 58 <= bounds[i + 1] - bounds[i] <= 64]
print(f"{len(s)} tiles, {len(bounds) - 1} counter groups, {len(cus)} of them single compute units; {tick / 1e3:.3f} GHz tick rate")
agg = []
for seg in cus:
    t0 = s[seg, 0].min()
    seg = seg[np.argsort(s[seg, 0])]
    agg.append(((s[seg, 0] - t0) / tick, (s[seg, 1] - t0) / tick, (s[seg, 2] - t0) / tick))
med = lambda f: float(np.median([f(a) for a in agg]))
print(f"per compute unit (medians over the {len(agg)}): {med(lambda a: len(a[0])):.0f} tiles, launch span {med(lambda a: a[2].max()):.1f} us")
print(f"  fill : the first 16 waves begin within {med(lambda a: a[0][15]):.2f} us; their draws are staged after {med(lambda a: np.median(a[1][:16] - a[0][:16])):.2f} us "
      f"(slowest {med(lambda a: (a[1][:16] - a[0][:16]).max()):.2f}); later waves: {med(lambda a: np.median((a[1] - a[0])[20:])):.2f} us")
print(f"  first wave computing at {med(lambda a: a[1].min()):.2f} us, 8 of 16 at {med(lambda a: np.sort(a[1][:16])[7]):.2f} us, all 16 at {med(lambda a: np.sort(a[1][:16])[15]):.2f} us")
print(f"  compute phase of a wave: first four staged (alone on their SIMDs) {med(lambda a: np.median((a[2] - a[1])[np.argsort(a[1][:16])[:4]])):.2f} us, "
      f"steady state (4 per SIMD) {med(lambda a: np.median((a[2] - a[1])[20:45])):.2f} us, last ten {med(lambda a: np.median((a[2] - a[1])[-10:])):.2f} us")
print(f"  drain: the last wave begins at {med(lambda a: a[0].max()):.1f} us, the unit ends at {med(lambda a: a[2].max()):.1f} us")
tgrid = np.arange(0.0, max(a[2].max() for a in agg) + 2.0, 2.0)
print("  t_us   resident  staging  computing   (waves per compute unit, mean over the units)")
for t in tgrid:
    res = np.mean([((a[0] <= t) & (a[2] > t)).sum() for a in agg])
    stg = np.mean([((a[0] <= t) & (a[1] > t)).sum() for a in agg])
    print(f"  {t:5.1f}  {res:8.1f}  {stg:7.1f}  {res - stg:9.1f}")
steady = np.mean([np.mean([((a[1] <= t) & (a[2] > t)).sum() for t in np.arange(14.0, 34.0, 1.0)]) for a in agg])
area = np.mean([(a[2] - a[1]).sum() for a in agg])
span = med(lambda a: a[2].max())
print(f"  computing waves in steady state {steady:.1f} per unit; compute-time integral / (span x steady) = {area / (span * steady):.3f} "
      f"-> {span * (1 - area / (span * steady)):.1f} us of the {span:.1f} us span are fill + drain")
