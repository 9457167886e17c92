#!/bin/bash
# The round's evidence run (on the GPU box, through scripts/gpu.sh): bench lines (defaults, the driver's arguments), rocprofv3
# kernel-trace stats of the bench defaults and of config 4, ONE PMC group per run (gpurun refuses --pmc combined with traces),
# configs 2 / 5 and the ring through kbench (scripts/collect_profiles_cfg.sh), directional pipeline, legacy stream, power probe.
#   usage: scripts/evidence.sh <tag>          -> gpurun_out/<tag>*/ ; afterwards: scripts/refresh_profiles.sh <round> <tag>
set -o pipefail
cd /root/repo
TAG=${1:-ev}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err || exit 1
cd /tmp && export TMPDIR=/tmp
# (--no-also: the appended legs launch the SAME kernel on other workloads and would blur the average)
rocprofv3 --kernel-trace --stats -d $OUT/kt_default -o p --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-also > $OUT/bench_default_under_profiler.json 2> $OUT/kt_default.log || exit 1
rocprofv3 --kernel-trace --stats -d $OUT/kt_c4 -o p --output-format csv -- python3 /root/repo/bench.py --config 4 --no-cpu-baseline --no-end-to-end --no-also > $OUT/bench_c4_under_profiler.json 2> $OUT/kt_c4.log || exit 1
echo "kernel traces done"
B="python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_$c.log || exit 1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_sq.log || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU -d $OUT/pmc_f64 -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_f64.log || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU -d $OUT/pmc_f32 -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_f32.log || exit 1
B4="python3 /root/repo/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --no-also"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc4_$c -o p --output-format csv -- $B4 > /dev/null 2> $OUT/pmc4_$c.log || exit 1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU -d $OUT/pmc4_sq -o p --output-format csv -- $B4 > /dev/null 2> $OUT/pmc4_sq.log || exit 1
echo "PMC passes done"
cd /root/repo
scripts/collect_profiles_cfg.sh ${TAG}_c2 5:100:10000 > /dev/null 2>&1 || exit 1
scripts/collect_profiles_cfg.sh ${TAG}_c5 10:100:10000 --xxz > /dev/null 2>&1 || exit 1
scripts/collect_profiles_cfg.sh ${TAG}_ring 7:100:10000 --ring > /dev/null 2>&1 || exit 1
echo "c2 / c5 / ring done"
python3 scripts/polish_rate.py 2>&1 | grep -v amdgpu.ids > $OUT/polish_rate.txt
python3 scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_bench.txt
python3 scripts/legacy_stream_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/legacy_stream_bench.txt
python3 scripts/power_probe.py > $OUT/power_probe.txt 2>&1 || true
head -8 $OUT/kt_default/p_kernel_stats.csv | cut -c1-160
head -5 $OUT/kt_c4/p_kernel_stats.csv | cut -c1-160
for f in bench_default bench_driver_args; do python3 - <<PY
import json
d=json.loads([l for l in open("$OUT/$f.json") if l.startswith("{")][-1])
print("$f", "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d.get("extras_failed"))
PY
done
