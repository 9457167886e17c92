#!/bin/bash
# round 4, call 32: the reduction with narrower workgroups for rows of 2049 .. 8192 values - same-box A/B of scripts/reduce_sweep.py
# against the build before, the tests that use the reduction, a block of the product-level fuzz (8 % of its rows are that long)
R=$PWD; OUT=$R/gpurun_out/r4aq; mkdir -p $OUT
echo "== before" | tee $OUT/reduce_ab.txt
ROBCHAR_HIP_LIB=$R/build/variants/lib_before_reduce.so timeout -k 10 200 python scripts/reduce_sweep.py 2>&1 | grep -v amdgpu.ids | grep "K=   20\|K=   40\|K=   81\|K=  100\|K=  163" | tee -a $OUT/reduce_ab.txt
echo "== after" | tee -a $OUT/reduce_ab.txt
timeout -k 10 200 python scripts/reduce_sweep.py 2>&1 | grep -v amdgpu.ids | grep "K=   20\|K=   40\|K=   81\|K=  100\|K=  163" | tee -a $OUT/reduce_ab.txt
python -m pytest tests -m gpu -q -x -k "reduce or metric or mcdatasim or rim or config or property or bench" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -3
export RC_ALLOW_DUPLICATE_DEVICES=1
SEED=300:309 NCFG=20 timeout -k 10 420 python scripts/fuzz_mcdatasim.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_mcdatasim.txt
