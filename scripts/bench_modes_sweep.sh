#!/bin/bash
# bench.py across its configurations / kernels at a few (steps, warm-up) combinations: every run must end with rc 0, one JSON
# line, parity green.   usage: scripts/bench_modes_sweep.sh <outdir>
OUT=${1:-gpurun_out/bench_modes}; mkdir -p $OUT
run() {
  tag=$(echo "$*" | tr ' -' '__'); python3 bench.py "$@" --no-also --no-end-to-end --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err; rc=$?
  python3 - "$OUT/$tag.json" $rc "$*" <<'PY'
import json, sys
f, rc, args = sys.argv[1], int(sys.argv[2]), sys.argv[3]
try:
    d = json.loads([l for l in open(f) if l.startswith('{"metric"')][0])
    c = d['check']
    print(f"[{args}] rc {rc}  value {d['value']:.4g}  step {d['ms_per_step']*1e3:9.2f} us  kernel {d['roofline']['kernel_ms']*1e3:9.2f} us  frac {d['roofline']['frac']:.3f}  err {c['max_abs_err_vs_oracle']:.1e}  rim_err {c['rim_err']:.1e}  {d['roofline']['kernel'][:40]}")
except Exception as e:
    print(f"[{args}] rc {rc}  NO LINE ({e!r})")
PY
}
for c in 2 5 30; do for sw in "20 5" "16 0" "33 0" "1 0"; do set -- $sw; run --config $c --steps $1 --warmup $2; done; done
for c in 4 40; do for sw in "1 0" "2 1" "3 0"; do set -- $sw; run --config $c --steps $1 --warmup $2; done; done
for k in tridiag_ql tridiag_adj; do run --kernel $k --steps 20 --warmup 5; done
run --config 5 --kernel tridiag_ql --steps 8 --warmup 2
