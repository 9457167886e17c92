#!/bin/bash
# quick VALU-instruction / cycle counters of one kernel build (development aid).  usage (on the GPU box):
#   scripts/pmc_quick.sh <tag> <lib.so> <N:C:K> [extra kbench flags]
set -e
TAG=$1; LIB=$2; SHAPE=$3; shift 3
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
export ROBCHAR_HIP_LIB=/root/repo/$LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/sq -o p --output-format csv -- python3 /root/repo/scripts/kbench.py --shapes $SHAPE --reps 10 "$@" > /dev/null 2> $OUT/sq.log
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mc_fid" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print("$TAG", k, {c: round(sum(v) / len(v) / w, 1) for c, v in d.items()}, "waves", w)
PY
