#!/bin/bash
# usage: SHAPES=... scripts/run_variants.sh <lib1.so> <lib2.so> ...   (kernel-only timing of each build)
SHAPES=${SHAPES:-5:100:10000,7:100:10000,10:100:10000}
for lib in "$@"; do
  echo "== $lib"
  ROBCHAR_HIP_LIB=$PWD/$lib python scripts/kbench.py --reps ${REPS:-40} --shapes $SHAPES $KBENCH_ARGS 2>&1 | grep -v amdgpu.ids
done
