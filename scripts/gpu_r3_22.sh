#!/bin/bash
# timeline of the driver's 20-step window: kernel trace of `bench.py --steps 20 --warmup 5` (no appended legs)
OUT=/root/repo/gpurun_out/r3w; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -o p --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also > $OUT/bench.json 2> $OUT/err.log
ls $OUT/kt | head; python3 - <<PY
import csv,glob,json
f=glob.glob("$OUT/kt/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last 20 chain-kernel dispatches + everything after the first of them
idx=[i for i,r in enumerate(rows) if "mc_fid_chain_kernel" in r["Kernel_Name"]]
first=idx[-20]
t0=int(rows[first]["Start_Timestamp"])
prev_end=int(rows[first-1]["End_Timestamp"])
print("gap before first timed launch (us): %.1f"%((t0-prev_end)/1e3), "prev kernel:", rows[first-1]["Kernel_Name"][:40])
for r in rows[first:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%8.1f %8.1f  dur %7.1f  %s"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,r["Kernel_Name"][:60]))
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("ms_per_step",d["ms_per_step"],"kernel_ms",d["roofline"]["kernel_ms"])
PY
