#!/bin/bash
OUT=gpurun_out/r3i; mkdir -p $OUT
for seed in 31 32 33 34 35 36; do
  SEED=$seed NCFG=2500 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
  echo "seed $seed rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz.txt
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>$OUT/bench.err; echo bench rc=$?
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_driver_args.json").read().strip().splitlines()[-1])
print("value %.4g" % d["value"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d["roofline"]["steady_state_untimed"]["kernel_ms"], d["extras_failed"])
PY
