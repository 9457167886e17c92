#!/bin/bash
# rocprofv3 evidence for ONE fidelity-kernel shape other than the bench default (BASELINE c2 / c5 / ...), kernel-only
# driver scripts/kbench.py.  usage (on the GPU box): scripts/collect_profiles_cfg.sh <tag> <N:C:K> [extra kbench flags]
# Separate passes for the kernel trace and for every PMC group (gpurun refuses --pmc combined with other traces).
set -e
TAG=$1; SHAPE=$2; shift 2
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- python3 /root/repo/scripts/kbench.py --shapes $SHAPE --reps 400 "$@" > $OUT/kbench_under_profiler.txt 2> $OUT/kt.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/sq -o p --output-format csv -- python3 /root/repo/scripts/kbench.py --shapes $SHAPE --reps 20 "$@" > /dev/null 2> $OUT/sq.log
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU -d $OUT/f64 -o p --output-format csv -- python3 /root/repo/scripts/kbench.py --shapes $SHAPE --reps 20 "$@" > /dev/null 2> $OUT/f64.log
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 /root/repo/scripts/kbench.py --shapes $SHAPE --reps 20 "$@" > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 /root/repo/scripts/kbench.py --shapes $SHAPE --reps 20 "$@" > /dev/null 2> $OUT/write.log
python3 /root/repo/scripts/kbench.py --shapes $SHAPE --reps 400 "$@" > $OUT/kbench.txt 2>&1
find $OUT -name "*.csv" | head -40
