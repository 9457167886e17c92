#!/bin/bash
# round 4, call 4 (first call of the second session; the outputs of calls 1-3 were lost with their container): full GPU suite,
# bench lines (defaults, driver arguments), kernel-trace stats of the bench defaults -> profiles/r04_*
R=$PWD; OUT=$R/gpurun_out/r4d; mkdir -p $OUT
python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -6
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "bench driver rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_default -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-also > $OUT/bench_default_under_profiler.json 2> $OUT/kt_default.log; echo "kt rc=$?"
cd $R
head -12 $OUT/kt_default/p_kernel_stats.csv | cut -c1-160
for f in bench_default bench_driver_args; do python3 - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print("$f", "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d.get("extras_failed"))
PY
done
