#!/bin/bash
# Builds a VARIANT of librobchar_hip.so for same-box A/B timing: scripts/build_variant.sh <name> <extra compiler flags...>
# -> build/variants/lib_<name>.so (git-ignored, travels to the GPU box with the snapshot); select it with ROBCHAR_HIP_LIB.
# RC_DEV_FEW_N (chain kernels for N = 5, 7, 10 only) keeps the compile at ~25 s.
set -e
cd "$(dirname "$0")/../code-robchar_amd/csrc"
name=$1; shift
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 "$@" -shared \
    -o ../../build/variants/lib_${name}.so robchar_hip.hip robchar_large.hip cache_io.cpp legacy_host.cpp -lpthread
echo "built build/variants/lib_${name}.so ($*)"
