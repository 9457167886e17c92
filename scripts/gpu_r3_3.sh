#!/bin/bash
# round-3 GPU session 3: full GPU suite (ring mixed route, sharded legacy generation), ring timings auto (mixed) vs ring_hh
OUT=gpurun_out/r3d; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
for k in auto ring_hh; do
  python scripts/kbench.py --ring --kernel $k --reps 200 --shapes 5:100:10000,7:100:10000,10:100:10000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ring_kbench.txt
done
python scripts/kbench.py --ring --out mid --reps 100 --shapes 7:100:10000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ring_kbench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/ring_trace -o t --output-format csv -- python3 /root/repo/scripts/kbench.py --ring --reps 200 --shapes 7:100:10000 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d /root/repo/$OUT/ring_pmc -o p --output-format csv -- python3 /root/repo/scripts/kbench.py --ring --reps 10 --shapes 7:100:10000 > /dev/null 2>&1
cd /root/repo
find $OUT/ring_trace -name "*kernel_stats.csv" -exec head -8 {} \;
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/ring_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mc_fid" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print(k, {c: round(sum(v) / len(v) / w, 1) for c, v in d.items()}, "waves", w)
PY
