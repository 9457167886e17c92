#!/bin/bash
# round 4, call 19: full GPU suite with MCDataSim's philox mode on the fused kernel; bench default (end_to_end legs)
R=$PWD; OUT=$R/gpurun_out/r4t; mkdir -p $OUT
python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -5
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
ROBCHAR_PHILOX_FUSED=0 python bench.py --no-cpu-baseline --no-also > $OUT/bench_default_unfused.json 2> /dev/null; echo "bench rc=$?"
python3 - <<PY
import json
for f in ("bench_default", "bench_default_unfused"):
    d=json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "value %.4g" % d["value"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], {k: v["wall_s"] for k, v in d["end_to_end"].items() if isinstance(v, dict)}, d.get("extras_failed"))
PY
timeout -k 10 200 python scripts/profile_c4_api.py 2>&1 | grep -v amdgpu.ids | grep "plain\|warm\|alloc" | tee $OUT/profile_c4_api.txt
