#!/bin/bash
OUT=gpurun_out/r3h; mkdir -p $OUT
for pr in 0.1 0.5 1.5 0.1 0.5 1.5 0.1 0.5 1.5; do
 ROBCHAR_BENCH_PREROLL_S=$pr python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also > $OUT/b.json 2>/dev/null
 python3 - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
print("preroll $pr", "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d["config"]["clock_preroll_launches_untimed"])
PY
done
