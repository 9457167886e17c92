#!/bin/bash
# round 4, call 8: full GPU suite on the build with (a) the device-side walk of the directional sample chain, (b) settled
# lanes keeping their polished eigenvalues when their tile takes the fp64 QL; the fuzz blocks that found (b), again
R=$PWD; OUT=$R/gpurun_out/r4h; mkdir -p $OUT/dump
python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -6
for r in 4000:4099 4100:4199; do
  FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1e-11 SEED=$r NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
done
ls $OUT/dump
timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 5:100:10000,7:100:10000 2>&1 | grep "N=" | tee $OUT/kbench.txt
timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 10:100:10000 --xxz 2>&1 | grep "N=" | tee -a $OUT/kbench.txt
timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 7:100:10000 --out 3 2>&1 | grep "N=" | tee -a $OUT/kbench.txt
