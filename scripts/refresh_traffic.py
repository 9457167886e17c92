#!/usr/bin/env python3
"""Regenerates profiles/traffic.json (what bench.py quotes as static `roofline.traffic` / `fp64_valu`) from the PMC passes
in profiles/<prefix>*.csv (RC_PROFILE_PREFIX: "r02_e_" for round 2's files, "r03_", "r04_", "r05_" - the default - for the later rounds')."""
import collections, csv, json, os
PRE = os.environ.get("RC_PROFILE_PREFIX", "r05_")
ROUND = {"r02_e_": 2, "r03_": 3, "r04_": 4, "r05_": 5}.get(PRE, int(PRE[1:3]) if PRE[1:3].isdigit() else 5)
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles") + "/"


def means(path, kern):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


t = json.load(open(P + "traffic.json"))
k3, k4, k2, k5 = ("mc_fid_chain_kernel<7, 2>", "mc_fid_chain_kernel<7, 1>", "mc_fid_chain_kernel<5, 2>", "mc_fid_chain_kernel<10, 2>")
fetch = means(P + PRE + "c3_pmc_fetch_size.csv", k3)["FETCH_SIZE"]
write = means(P + PRE + "c3_pmc_write_size.csv", k3)["WRITE_SIZE"]
sq, f64, f32 = (means(P + PRE + f"c3_pmc_{n}.csv", k3) for n in ("sq", "f64_mix", "f32_mix"))
rd, wr = fetch * 2048, write * 1024
flop = lambda m, s: 64 * (m[f"SQ_INSTS_VALU_ADD_{s}"] + m[f"SQ_INSTS_VALU_MUL_{s}"] + 2 * m[f"SQ_INSTS_VALU_FMA_{s}"] + m[f"SQ_INSTS_VALU_TRANS_{s}"])
mix = lambda m, s: {n.lower(): m[f"SQ_INSTS_VALU_{n}_{s}"] for n in ("ADD", "MUL", "FMA", "TRANS")}
f4 = means(P + PRE + "c4_pmc_fetch_size.csv", k4)["FETCH_SIZE"]
w4 = means(P + PRE + "c4_pmc_write_size.csv", k4)["WRITE_SIZE"]
s4, c2, c5 = means(P + PRE + "c4_pmc_sq.csv", k4), means(P + PRE + "c2_pmc_sq.csv", k2), means(P + PRE + "c5_pmc_sq.csv", k5)
c5b = means(P + PRE + "c5_pmc_fetch.csv", k5)["FETCH_SIZE"] * 2048 + means(P + PRE + "c5_pmc_write.csv", k5)["WRITE_SIZE"] * 1024
t.update({
    "build": f"final build of round {ROUND}: mixed-precision eigenvalues for N = 3..13 (fp32 QL rotations with an absolute split "
             "threshold + fp64 Ehrlich-Aberth first step (Halley on the stepping path) with the critical-point guard on every path, all-fp64 QL as the tile-wide fallback, "
             "degenerate samples repaired in registers), batched weight reciprocals, -fno-slp-vectorize",
    "collected": f"round {ROUND}, scripts/evidence.sh (rounds 2-4: scripts/gpu_call_r{ROUND}_final.sh in the history): rocprofv3 --pmc passes, ONE counter group per run, over `bench.py --steps "
                 "20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also` (the untimed clock pre-roll launches are dispatches "
                 "of the same kernel and are included in the means); files profiles/" + PRE + "c3_pmc_*.csv",
    "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
    "hbm_bytes_per_launch": rd + wr,
    "note": f"measured traffic = {(rd + wr) / 176e6:.4f} x algorithmic bytes: every draw is fetched exactly once, no re-reads, no "
            "scratch; the writes are the 8.0 MB of fidelities plus the diagnostic atomics of rc_stats_polish_tiles. With the read "
            "stream rotating over 504 MB it cannot be served from the 256 MiB Infinity Cache",
    "valu_insts_per_launch": sq["SQ_INSTS_VALU"], "waves_per_launch": sq["SQ_WAVES"],
    "valu_note": f"SQ_INSTS_VALU / SQ_WAVES = {sq['SQ_INSTS_VALU'] / sq['SQ_WAVES']:.0f} VALU wave-instructions per 64-sample wave "
                 "(all-fp64 build of round 1 / early round 2: 1698; final build of round 2: 1430)",
    "fp64_mix_wave_insts_per_launch": mix(f64, "F64"), "fp32_mix_wave_insts_per_launch": mix(f32, "F32"),
    "fp64_flop_per_launch": flop(f64, "F64"), "fp32_flop_per_launch": flop(f32, "F32"),
    "fp64_note": f"64 lanes x (add + mul + 2 fma + trans): {flop(f64, 'F64') / 1e6:.0f} fp64 flop + {flop(f32, 'F32') / 1e6:.0f} "
                 "fp32 flop per evaluation (all-fp64 build: 1973 fp64 flop)"})
t["config4"].update({"FETCH_SIZE_KiB": f4, "WRITE_SIZE_KiB": w4, "hbm_bytes_per_launch": f4 * 2048 + w4 * 1024,
                     "valu_insts_per_launch": s4["SQ_INSTS_VALU"], "valu_per_wave": s4["SQ_INSTS_VALU"] / s4["SQ_WAVES"]})
t["config2"]["valu_per_wave"] = c2["SQ_INSTS_VALU"] / c2["SQ_WAVES"]
t["config2"]["files"] = "profiles/" + PRE + "c2_*"
t["config4"]["files"] = "profiles/" + PRE + "c4_pmc_*.csv"
t["config5"] = {"kernel": k5, "workload": "BASELINE config 5 (N = 10 XXZ, 0->9, 100 x 10000), scripts/kbench.py --xxz",
                "valu_per_wave": c5["SQ_INSTS_VALU"] / c5["SQ_WAVES"], "hbm_bytes_per_launch": c5b,
                "algorithmic_bytes_per_launch": 248e6, "files": "profiles/" + PRE + "c5_*"}
json.dump(t, open(P + "traffic.json", "w"), indent=1)
w = sq["SQ_WAVES"]
print(t["valu_note"]); print(t["fp64_note"]); print(t["note"][:42])
print("per wave:", {k: round(v / w, 1) for k, v in {**t["fp64_mix_wave_insts_per_launch"], **{"f32_" + a: b for a, b in t["fp32_mix_wave_insts_per_launch"].items()}}.items()})
print("c4 valu/wave %.0f traffic %.5f | c2 valu/wave %.0f | c5 valu/wave %.0f traffic %.4f | c3 write KiB %.1f" % (
    t["config4"]["valu_per_wave"], t["config4"]["hbm_bytes_per_launch"] / 17.6e9, t["config2"]["valu_per_wave"],
    t["config5"]["valu_per_wave"], c5b / 248e6, write))
