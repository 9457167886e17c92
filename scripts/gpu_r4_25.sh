#!/bin/bash
# round 4, call 25: N = 14 / 15 / 16 end-to-end pairs (0 -> N-1) through the end-to-end instantiation (build before: one wave
# per SIMD at N = 15, 16) and through the general adjugate instantiation (this build); same box, alternating; then the tests
# that cover these sizes
R=$PWD; OUT=$R/gpurun_out/r4ae; mkdir -p $OUT
for rep in 1 2; do
  echo "== before (end-to-end instantiation at every N)" | tee -a $OUT/ab_n15.txt
  ROBCHAR_HIP_LIB=$R/build/variants/lib_before_n15.so timeout -k 10 200 python scripts/kbench.py --reps 200 --shapes 14:100:10000,15:100:10000,16:100:10000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_n15.txt
  echo "== after (general adjugate instantiation at N = 15, 16)" | tee -a $OUT/ab_n15.txt
  timeout -k 10 200 python scripts/kbench.py --reps 200 --shapes 14:100:10000,15:100:10000,16:100:10000 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_n15.txt
done
python -m pytest tests -m gpu -q -x -k "parity or property or philox or fuzz or golden or guard" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -3
