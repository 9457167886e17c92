#!/bin/bash
# jump-ahead kernel geometry: 8 workgroups x 78 words x 3 term groups (lib_jump8) vs 24 x 26 x 9
for lib in scripts/ubench/lib_jump8.so code-robchar_amd/csrc/librobchar_hip.so; do
  echo "== $lib"; ROBCHAR_HIP_LIB=$PWD/$lib python scripts/legacy_stream_bench.py 2>&1 | grep -v amdgpu.ids
  ROBCHAR_HIP_LIB=$PWD/$lib python scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
done
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_mcsim.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -3
SEED=77 NCFG=10 timeout -k 10 300 python scripts/fuzz_parity.py 2>&1 | grep "legacy stream\|directional draws"
