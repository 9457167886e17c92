#!/bin/bash
# round 4, call 10: priority of the bench's reduction stream (A/B, the driver's 20-step window and the default run); two more
# fuzz blocks on the final build; GPU tests touched since the evidence run
R=$PWD; OUT=$R/gpurun_out/r4j; mkdir -p $OUT/dump
python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_mcsim.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest.log
line() { python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"])
PY
}
for round in 1 2 3; do
  for prio in -1 0; do
    ROBCHAR_BENCH_SIDE_PRIO=$prio python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also > $OUT/drv_p${prio}_$round.json 2>/dev/null; line $OUT/drv_p${prio}_$round.json
  done
done
for prio in -1 0; do
  ROBCHAR_BENCH_SIDE_PRIO=$prio python bench.py --no-cpu-baseline --no-end-to-end --no-also > $OUT/def_p${prio}.json 2>/dev/null; line $OUT/def_p${prio}.json
done
for r in 4200:4299 4300:4399; do
  FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1e-11 SEED=$r NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
done
ls $OUT/dump
