#!/bin/bash
# last state of the round: GPU suite, smoke, bench lines (defaults, driver arguments, two self-launched gloo ranks)
OUT=/root/repo/gpurun_out/r3H; mkdir -p $OUT
cd /root/repo
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -3 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err || exit 1
ROBCHAR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_two_ranks_gloo.json 2> $OUT/bench_two_ranks_gloo.err || exit 1
for f in bench_default bench_driver_args bench_two_ranks_gloo; do python3 - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print("$f", "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d.get("extras_failed"))
print("   e2e", {k: v.get("wall_s") for k, v in (d.get("end_to_end") or {}).items() if isinstance(v, dict)})
print("   also", {k: (v.get("value") or v.get("kernel_ms")) for k, v in d["also"].items() if isinstance(v, dict)})
PY
done
