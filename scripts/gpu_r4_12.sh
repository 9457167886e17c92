#!/bin/bash
# round 4, call 12: full GPU suite on the build with the Halley polish of the all-fp64 QL's eigenvalues at N >= 14 (general
# adjugate mode); its cost (same-box A/B, N = 14 / 16, 0 -> 3); the fuzz block that held the N = 16 worst case, again, and two
# new blocks
R=$PWD; OUT=$R/gpurun_out/r4l; mkdir -p $OUT/dump
python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -4
kb() { timeout -k 10 200 python scripts/kbench.py --reps 200 "$@" 2>&1 | grep "N="; }
for round in 1 2; do
  for v in nopolish polish; do
    export ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so
    echo "== $v" >> $OUT/ab_polish.txt
    kb --shapes 14:100:10000,16:100:10000 --out 3 >> $OUT/ab_polish.txt
    kb --shapes 13:100:10000,16:100:10000 >> $OUT/ab_polish.txt
  done
done
unset ROBCHAR_HIP_LIB
cat $OUT/ab_polish.txt
for r in 4200:4299 4400:4499 4500:4599; do
  FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1e-11 SEED=$r NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
done
ls $OUT/dump
