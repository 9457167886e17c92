#!/bin/bash
# round-3 GPU session 4: directional device pipeline (tests + bench + kernel split), ring tests again, bench.py default
OUT=gpurun_out/r3e; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_mcsim.py tests/test_gpu_parity.py -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
python scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/directional_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/dir_trace -o t --output-format csv -- python3 /root/repo/scripts/directional_bench.py > /dev/null 2>&1
cd /root/repo
find $OUT/dir_trace -name "*kernel_stats.csv" -exec head -14 {} \; | cut -c1-200
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_driver_args.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","n_gpus")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["also"].get("cold_20_steps_kernel_ms"), d["extras_failed"], d["check"])
print({k:v.get("wall_s") for k,v in d["end_to_end"].items() if isinstance(v,dict)})
PY
