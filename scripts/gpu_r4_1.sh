#!/bin/bash
# round 4, call 1: full GPU suite (incl. the ragged 3-rank / 4-rank bench rehearsals), same-box A/B of the a-posteriori
# sum-rule guard (build/variants/lib_noguard.so vs lib_guard_few.so), one adversarial fuzz block
OUT=gpurun_out/r4a; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
tail -5 $OUT/pytest.log
for v in noguard guard_few noguard guard_few; do
  export ROBCHAR_HIP_LIB=$PWD/build/variants/lib_$v.so
  echo "== $v" >> $OUT/ab.txt
  timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 5:100:10000,7:100:10000 2>&1 | grep "N=" >> $OUT/ab.txt
  timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 10:100:10000 --xxz 2>&1 | grep "N=" >> $OUT/ab.txt
  timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 7:100:10000 --out 3 2>&1 | grep "N=" >> $OUT/ab.txt
  timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 7:100:10000 --shipped 2>&1 | grep "N=" >> $OUT/ab.txt
  timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 5:100:10000,7:100:10000,10:100:10000 --ring 2>&1 | grep "N=" >> $OUT/ab.txt
done
unset ROBCHAR_HIP_LIB
cat $OUT/ab.txt
SEED=3000:3099 NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz.txt
