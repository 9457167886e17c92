#!/bin/bash
# bench.py under unusual --steps / --warmup combinations (group boundaries of the reduction pipeline: 16-launch groups, deferred tail):
# every run must end with rc 0, one JSON line and a passing parity check.   usage: scripts/bench_args_sweep.sh <outdir>
OUT=${1:-gpurun_out/bench_args}; mkdir -p $OUT
for sw in "1 0" "1 1" "2 0" "15 0" "16 0" "17 0" "31 1" "32 0" "33 16" "48 17" "20 5" "64 3" "100 0"; do
  set -- $sw
  python3 bench.py --steps $1 --warmup $2 --no-also --no-end-to-end --no-cpu-baseline > $OUT/s$1_w$2.json 2> $OUT/s$1_w$2.err; rc=$?
  python3 - "$OUT/s$1_w$2.json" $rc $1 $2 <<'PY'
import json, sys
f, rc, s, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
try:
    d = json.loads([l for l in open(f) if l.startswith('{"metric"')][0])
    print(f"steps {s:3d} warmup {w:2d}: rc {rc}  value {d['value']:.4g}  step {d['ms_per_step']*1e3:7.2f} us  kernel {d['roofline']['kernel_ms']*1e3:6.2f} us  err {d['check']['max_abs_err_vs_oracle']:.1e}  rim_err {d['check']['rim_err']:.1e}  steps_ok {d['steps'] == s and d['warmup'] == w}")
except Exception as e:
    print(f"steps {s} warmup {w}: rc {rc}  NO LINE ({e!r})")
PY
done
