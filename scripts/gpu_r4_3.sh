#!/bin/bash
# round 4, call 3: GPU suite on the slimmed stepping path (own-step root selection, distinct-roots check skipped when it cannot
# fail) + guard m = 0; same-box A/B against the previous stepping path; stage timing of the directional pipeline; one fuzz block
R=$PWD; OUT=$R/gpurun_out/r4c; mkdir -p $OUT
python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
tail -4 $OUT/pytest.log
kb() { timeout -k 10 200 python scripts/kbench.py --reps 300 "$@" 2>&1 | grep "N="; }
for round in 1 2; do
  for v in oldstep few oldsel; do
    export ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so
    echo "== $v" >> $OUT/ab_step.txt
    kb --shapes 5:100:10000,7:100:10000 >> $OUT/ab_step.txt
    kb --shapes 10:100:10000 --xxz >> $OUT/ab_step.txt
    kb --shapes 7:100:10000 --shipped >> $OUT/ab_step.txt
    kb --shapes 7:100:10000 --out 3 >> $OUT/ab_step.txt
    kb --shapes 7:100:10000,10:100:10000 --ring >> $OUT/ab_step.txt
  done
done
unset ROBCHAR_HIP_LIB
cat $OUT/ab_step.txt
timeout -k 10 200 python scripts/directional_profile.py 2>&1 | grep -v amdgpu.ids | tee $OUT/directional_profile.txt
SEED=3100:3199 NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz.txt
