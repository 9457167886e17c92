#!/bin/bash
# round 4, call 16: Philox rounds with v_bitop3_b32 (three-input xor): parity tests of the counter-based draws, kernel time A/B
R=$PWD; OUT=$R/gpurun_out/r4p; mkdir -p $OUT
python -m pytest tests -m gpu -q -x -k "philox or config4 or whole_problem or single_process_multi_device or bench_config4" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -3
for i in 1 2 3; do
  ROBCHAR_HIP_LIB=$R/build/variants/lib_few.so timeout -k 10 120 python scripts/philox_bench.py 2>&1 | grep philox | tee -a $OUT/philox.txt
  timeout -k 10 120 python scripts/philox_bench.py 2>&1 | grep philox | tee -a $OUT/philox.txt
done
