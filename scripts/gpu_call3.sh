#!/bin/bash
# round-2 GPU call 3: RCCL single-rank rehearsal tests, then the rocprofv3 evidence of the bench defaults and config 4
set -o pipefail
cd /root/repo
mkdir -p gpurun_out/r2c3
python -m pytest tests/test_gpu_bench.py tests/test_gpu_mcsim.py -x -q > gpurun_out/r2c3/pytest.log 2>&1; rc=$?
grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r2c3/pytest.log | tail -25
[ $rc -ne 0 ] && exit $rc
OUT=/root/repo/gpurun_out/r2c3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_default -o p --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/bench_default_under_profiler.json 2> $OUT/kt_default.log || exit 1
rocprofv3 --kernel-trace --stats -d $OUT/kt_c4 -o p --output-format csv -- python3 /root/repo/bench.py --config 4 --no-cpu-baseline --no-end-to-end > $OUT/bench_c4_under_profiler.json 2> $OUT/kt_c4.log || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc_$c -o p --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also > /dev/null 2> $OUT/pmc_$c.log || exit 1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_sq -o p --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also > /dev/null 2> $OUT/pmc_sq.log || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU -d $OUT/pmc_f64 -o p --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also > /dev/null 2> $OUT/pmc_f64.log || exit 1
# config 4: traffic of the adjugate-mode kernel
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc4_$c -o p --output-format csv -- python3 /root/repo/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/pmc4_$c.log || exit 1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU -d $OUT/pmc4_sq -o p --output-format csv -- python3 /root/repo/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/pmc4_sq.log || exit 1
cd /root/repo
head -12 $OUT/kt_default/p_kernel_stats.csv | cut -c1-160
head -8 $OUT/kt_c4/p_kernel_stats.csv | cut -c1-160
