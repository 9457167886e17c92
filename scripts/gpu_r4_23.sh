#!/bin/bash
# round 4, call 23: the fuzz script's RNG blocks (legacy stream: 40 cases, directional draws: 25 cases per seed; generator state
# bit-identical to NumPy's) under eight further auxiliary seeds on the final build (rebuilt jump / count / emit kernels), beside a
# token main block; then the legacy stream at BASELINE-config-4 size (2.1e9 normals, 40 segments) against the host generator's state
R=$PWD; OUT=$R/gpurun_out/r4ac; mkdir -p $OUT
for a in 21 22 23 24 25 26 27 28; do
  FUZZ_AUX_SEED=$a FUZZ_FUSED=20 FUZZ_DIR=10 SEED=$((5700 + a)) NCFG=20 timeout -k 10 200 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "legacy stream\|directional draws\|fused Philox" | tee -a $OUT/fuzz_rng.txt
  echo "aux seed $a rc=${PIPESTATUS[0]}" | tee -a $OUT/fuzz_rng.txt
done
timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/legacy_big.txt
import importlib, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
be = importlib.import_module("code-robchar_amd.backend")
# 2.1e8 normals (a tenth of config 4: 100 x 1e5 x 21) in ONE call: several 2^27-word segments; state against NumPy's own
n = 100 * 100000 * 21
for rep in range(2):
    np.random.seed(7)
    np.random.standard_normal(3)                       # odd position: a cached normal is pending
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = be.legacy_normal_periods(1, n, 0, np.array([0.05]))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st_dev = np.random.get_state()
    print(f"{n:.2e} legacy normals on the device in {dt * 1e3:.1f} ms ({n / dt:.3e} per s)")
np.random.seed(7); np.random.standard_normal(3)
t0 = time.perf_counter(); ref = np.random.normal(scale=0.05, size=n); dt = time.perf_counter() - t0
st_host = np.random.get_state()
print(f"NumPy on the host: {dt * 1e3:.0f} ms")
assert np.array_equal(st_dev[1], st_host[1]) and st_dev[2:] == st_host[2:], "state differs"
got = out.cpu().numpy().reshape(-1)
print("state identical; max |value difference| / ulp-scale:", float(np.abs(got - ref).max() / (2.2e-16 * np.abs(ref).max())), " bit-identical fraction:", float((got == ref).mean()))
PY
