#!/bin/bash
# round 4, call 11: bench tests with the reduction stream at normal priority; config 4 A/B of that priority; directional walk
# test incl. the 9e6-sample case (two passes); structured / directional scalar API timing
R=$PWD; OUT=$R/gpurun_out/r4k; mkdir -p $OUT
python -m pytest tests/test_gpu_bench.py tests/test_gpu_round4.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -4
line() { python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"])
PY
}
for round in 1 2; do
  for prio in -1 0; do
    ROBCHAR_BENCH_SIDE_PRIO=$prio python bench.py --config 4 --no-cpu-baseline --no-end-to-end --no-also > $OUT/c4_p${prio}_$round.json 2>/dev/null; line $OUT/c4_p${prio}_$round.json
    ROBCHAR_BENCH_SIDE_PRIO=$prio python bench.py --config 5 --no-cpu-baseline --no-end-to-end --no-also > $OUT/c5_p${prio}_$round.json 2>/dev/null; line $OUT/c5_p${prio}_$round.json
  done
done
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import importlib, time, numpy as np
noise = importlib.import_module("code-robchar_amd.noise")
rng = np.random.default_rng(0)
N = 7
x = np.concatenate([rng.uniform(-10, 10, N), [11.0]])
for cls in ("structured_perturbation", "directional_perturbation"):
    nm = getattr(noise, cls)(Nspin=N, inspin=0, outspin=6, noise=0.05)
    np.random.seed(3)
    for n in (200, 4000):
        t0 = time.perf_counter()
        s = 0.0
        for _ in range(n):
            s += nm.evaluate_noisy_fidelity(x, ham_noisy=True)
        dt = time.perf_counter() - t0
        print(f"scalar API {cls}: {n} calls, {dt / n * 1e6:.1f} us per call, mean fidelity {s / n:.6f}")
PY
