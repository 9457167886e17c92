#!/bin/bash
# round 4, call 26: the whole GPU suite and the two bench lines on the round's last build (N = 15 / 16 dispatch change)
R=$PWD; OUT=$R/gpurun_out/r4af; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -3
[ $rc -eq 0 ] || exit 1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err || exit 1
for f in bench_default bench_driver_args; do python3 - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print("$f", "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d.get("extras_failed"))
PY
done
