#!/bin/bash
# round-2 GPU call 2: full GPU test suite on the new code, default bench line, explicit config-4 line
set -o pipefail
cd /root/repo
mkdir -p gpurun_out/r2c2
python -m pytest tests -m gpu -x -q > gpurun_out/r2c2/pytest.log 2>&1; rc=$?
tail -25 gpurun_out/r2c2/pytest.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/r2c2/bench_driver_args.json 2> gpurun_out/r2c2/bench_driver_args.err || { tail -20 gpurun_out/r2c2/bench_driver_args.err; exit 1; }
python bench.py > gpurun_out/r2c2/bench_default.json 2> gpurun_out/r2c2/bench_default.err || { tail -20 gpurun_out/r2c2/bench_default.err; exit 1; }
python bench.py --config 4 > gpurun_out/r2c2/bench_c4.json 2> gpurun_out/r2c2/bench_c4.err || { tail -20 gpurun_out/r2c2/bench_c4.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_driver_args","bench_default","bench_c4"):
    d=json.load(open(f"gpurun_out/r2c2/{f}.json"))
    print(f, "value %.3e"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel_ms %.4f"%d["roofline"]["kernel_ms"], "frac %.3f"%d["roofline"]["frac"], d["check"])
    if d.get("also"): print("  also c4:", {k:d["also"]["config4_strong"].get(k) for k in ("value","ms_per_step","error")})
    if d.get("end_to_end"): print("  e2e:", {k:(round(v["wall_s"],4), "%.2e"%v["evals_per_s"]) for k,v in d["end_to_end"].items() if isinstance(v,dict)} if "error" not in d["end_to_end"] else d["end_to_end"])
    if d.get("cpu_baseline"): print("  cpu:", "%.3e"%d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
