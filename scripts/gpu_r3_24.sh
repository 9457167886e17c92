#!/bin/bash
# sub-stream length of the device MT19937 stream: 2048 state blocks (lib_b2048) vs 512 + a 64 B jump polynomial
for lib in scripts/ubench/lib_b2048.so code-robchar_amd/csrc/librobchar_hip.so; do
  echo "== $lib"; ROBCHAR_HIP_LIB=$PWD/$lib python scripts/legacy_stream_bench.py 2>&1 | grep -v amdgpu.ids
  ROBCHAR_HIP_LIB=$PWD/$lib python scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids | tail -4
done
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_mcsim.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -3
SEED=77 NCFG=10 timeout -k 10 300 python scripts/fuzz_parity.py 2>&1 | grep "legacy stream\|directional draws"
