#!/bin/bash
OUT=gpurun_out/r3g; mkdir -p $OUT
for i in 1 2 3 4 5; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also > $OUT/b20_$i.json 2>/dev/null; done
python bench.py --no-cpu-baseline --no-end-to-end --no-also > $OUT/b4000.json 2>/dev/null
ROBCHAR_BENCH_GROUP=16 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also > $OUT/b20_g16.json 2>/dev/null
for f in b20_1 b20_2 b20_3 b20_4 b20_5 b4000 b20_g16; do python3 - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print("$f", "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d["roofline"]["kernel_launches_timed"])
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -3
