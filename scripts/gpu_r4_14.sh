#!/bin/bash
# round 4, call 14: full GPU suite on the round's last build; one fuzz block with the directional-entry block (200 random cases)
R=$PWD; OUT=$R/gpurun_out/r4n; mkdir -p $OUT/dump
python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -4
FUZZ_DIR=200 FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1.2e-11 SEED=4700:4749 NCFG=150 timeout -k 10 600 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_driver_args.json").read().strip().splitlines()[-1])
print("driver args", "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d.get("extras_failed"))
PY
