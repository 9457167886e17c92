#!/bin/bash
# round 4, call 33: reduction of 10 000-value rows with 256 threads x 40 cached values - the sweep again, and the bench's step time
# (the reduction of 16 steps' rows runs on a side stream beside the fidelity kernel) against the build before, same box, alternating
R=$PWD; OUT=$R/gpurun_out/r4ar; mkdir -p $OUT
echo "== after (sweep)" | tee $OUT/reduce_ab2.txt
timeout -k 10 200 python scripts/reduce_sweep.py 2>&1 | grep -v amdgpu.ids | grep "K=   81\|K=  100\|K=  163" | tee -a $OUT/reduce_ab2.txt
B="python bench.py --no-cpu-baseline --no-also --no-end-to-end"
for rep in 1 2 3; do
  for v in before after; do
    if [ $v = before ]; then export ROBCHAR_HIP_LIB=$R/build/variants/lib_before_reduce.so; else unset ROBCHAR_HIP_LIB; fi
    $B 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'value %.4g' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'kernel_ms %.5f' % d['roofline']['kernel_ms'], d['check']['metric_table_sha256'], d['check'].get('rim_err'))" | tee -a $OUT/reduce_ab2.txt
  done
done
unset ROBCHAR_HIP_LIB
python -m pytest tests -m gpu -q -x -k "reduce or metric or rim or bench" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -2
