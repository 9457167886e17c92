#!/bin/bash
# the reference's shipped L-BFGS controllers (near mirror-symmetric: close eigenvalue pairs) vs uniform random ones:
# split tolerance of the fp32 QL (1e-4 today, 3e-5, 1e-5) and the one-step acceptance constant (1e-14 today, 1e-13), same box
OUT=gpurun_out/r3v; mkdir -p $OUT
LIBS="code-robchar_amd/csrc/librobchar_hip.so scripts/ubench/lib_eps3e5.so scripts/ubench/lib_eps1e5.so scripts/ubench/lib_acc1e13.so"
for rep in 1 2; do
  REPS=300 SHAPES=7:100:10000 KBENCH_ARGS=--shipped scripts/run_variants.sh $LIBS
  REPS=300 SHAPES=7:100:10000 KBENCH_ARGS="--shipped --out mid" scripts/run_variants.sh $LIBS
  REPS=300 SHAPES=7:100:10000,10:100:10000 scripts/run_variants.sh $LIBS
done 2>&1 | tee $OUT/ab.txt
