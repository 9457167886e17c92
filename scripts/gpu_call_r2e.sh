#!/bin/bash
# round-2 evidence of the mixed-precision build (profiles/r02_e_*): GPU test suite, then rocprofv3 kernel-trace stats
# of the bench defaults and of config 4, then one PMC group per run (gpurun refuses --pmc combined with traces).
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/r2e
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -5
[ $rc -ne 0 ] && exit $rc
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_default -o p --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/bench_default_under_profiler.json 2> $OUT/kt_default.log || exit 1
rocprofv3 --kernel-trace --stats -d $OUT/kt_c4 -o p --output-format csv -- python3 /root/repo/bench.py --config 4 --no-cpu-baseline --no-end-to-end > $OUT/bench_c4_under_profiler.json 2> $OUT/kt_c4.log || exit 1
B="python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_$c.log || exit 1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_sq -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_sq.log || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU -d $OUT/pmc_f64 -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_f64.log || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU -d $OUT/pmc_f32 -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_f32.log || exit 1
B4="python3 /root/repo/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc4_$c -o p --output-format csv -- $B4 > /dev/null 2> $OUT/pmc4_$c.log || exit 1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU -d $OUT/pmc4_sq -o p --output-format csv -- $B4 > /dev/null 2> $OUT/pmc4_sq.log || exit 1
cd /root/repo
scripts/collect_profiles_cfg.sh r2e_c2 5:100:10000 > /dev/null 2>&1 || exit 1
python3 scripts/power_probe.py > $OUT/power_probe.txt 2>&1 || true
head -12 $OUT/kt_default/p_kernel_stats.csv | cut -c1-160
head -8 $OUT/kt_c4/p_kernel_stats.csv | cut -c1-160
