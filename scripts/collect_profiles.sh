#!/bin/bash
# Re-collects the rocprofv3 evidence of the benchmark workload into gpurun_out/$1/ (run on the GPU box via gpurun;
# copy the summaries into profiles/ afterwards).  Separate passes for the kernel trace and for every PMC group.
set -e
TAG=${1:-prof}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- python3 /root/repo/bench.py --steps 4000 --warmup 400 > $OUT/bench_under_profiler.json 2> $OUT/kt.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 5 > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 5 > /dev/null 2> $OUT/write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/sq -o p --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 5 > /dev/null 2> $OUT/sq.log
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/tcc -o p --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 5 > /dev/null 2> $OUT/tcc.log
python3 /root/repo/bench.py > $OUT/bench.json 2> $OUT/bench.log
ls -R $OUT | head -40
