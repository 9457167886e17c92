#!/bin/bash
# round 4, call 9: directional pipeline with eight-sample hops inside the blocks and 64-block superblocks (tests, bench, stage
# timing, kernel trace); guard level A/B again (the first session's record was lost with its container): none / m = 0 /
# three moments / m = 0 on the end-to-end weights too; where config 4 through the product API spends its 13 ms
R=$PWD; OUT=$R/gpurun_out/r4i; mkdir -p $OUT
python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -x -k "directional" > $OUT/pytest_dir.log 2>&1; echo "pytest directional rc=$?"
tail -3 $OUT/pytest_dir.log
timeout -k 10 200 python scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_bench.txt
timeout -k 10 200 python scripts/directional_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_profile.txt
cat $OUT/directional_bench.txt $OUT/directional_profile.txt
kb() { timeout -k 10 200 python scripts/kbench.py --reps 300 "$@" 2>&1 | grep "N="; }
for round in 1 2; do
  for v in noguard few m3 endsguard; do
    export ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so
    echo "== $v" >> $OUT/ab_guard.txt
    kb --shapes 5:100:10000,7:100:10000 >> $OUT/ab_guard.txt
    kb --shapes 10:100:10000 --xxz >> $OUT/ab_guard.txt
    kb --shapes 7:100:10000 --out 3 >> $OUT/ab_guard.txt
    kb --shapes 5:100:10000,7:100:10000,10:100:10000 --ring >> $OUT/ab_guard.txt
  done
done
unset ROBCHAR_HIP_LIB
cat $OUT/ab_guard.txt
timeout -k 10 300 python scripts/profile_c4_api.py 2>&1 | grep -v amdgpu.ids > $OUT/profile_c4_api.txt; head -70 $OUT/profile_c4_api.txt | cut -c1-200; tail -4 $OUT/profile_c4_api.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/dir_kt -o p --output-format csv -- python3 $R/scripts/directional_bench.py > $OUT/dir_kt.log 2>&1
head -24 $OUT/dir_kt/p_kernel_stats.csv | cut -c1-150
