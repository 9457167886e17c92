#!/bin/bash
OUT=/root/repo/gpurun_out/r3x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_dir -o p --output-format csv -- python3 /root/repo/scripts/directional_bench.py > $OUT/dir.txt 2> $OUT/dir.err
grep -v amdgpu $OUT/dir.txt | head -12
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kt_dir/p_kernel_stats.csv")))
for r in rows[:22]:
    print("%6s calls %9.1f us total %8.1f us avg  %s"%(r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, r["Name"][:90]))
PY
