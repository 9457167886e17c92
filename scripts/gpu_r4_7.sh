#!/bin/bash
# round 4, call 7: directional pipeline with the sample chain walked on the device (tests, bench, stage timing); branch-weight
# hints A/B (cold paths moved behind the hot path); VALU instructions per wave with / without the stepping path's code
R=$PWD; OUT=$R/gpurun_out/r4g; mkdir -p $OUT
python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -x -k "directional" > $OUT/pytest_dir.log 2>&1; echo "pytest directional rc=$?"
tail -3 $OUT/pytest_dir.log
timeout -k 10 200 python scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_bench.txt
timeout -k 10 200 python scripts/directional_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_profile.txt
RC_DIR_WALK=host timeout -k 10 200 python scripts/directional_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_profile_hostwalk.txt
cat $OUT/directional_bench.txt $OUT/directional_profile.txt $OUT/directional_profile_hostwalk.txt
kb() { timeout -k 10 200 python scripts/kbench.py --reps 300 "$@" 2>&1 | grep "N="; }
for round in 1 2; do
  for v in noexpect expect; do
    export ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so
    echo "== $v" >> $OUT/ab_expect.txt
    kb --shapes 5:100:10000,7:100:10000 >> $OUT/ab_expect.txt
    kb --shapes 10:100:10000 --xxz >> $OUT/ab_expect.txt
    kb --shapes 7:100:10000 --shipped >> $OUT/ab_expect.txt
    kb --shapes 7:100:10000 --out 3 >> $OUT/ab_expect.txt
    kb --shapes 7:100:10000,10:100:10000 --ring >> $OUT/ab_expect.txt
  done
done
unset ROBCHAR_HIP_LIB
cat $OUT/ab_expect.txt
for v in few nostep stepnotrun expect; do
  bash scripts/pmc_quick.sh r4g_pmc_$v build/variants/lib_$v.so 7:100:10000 2>&1 | tail -2
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/dir_kt -o p --output-format csv -- python3 $R/scripts/directional_bench.py > $OUT/dir_kt.log 2>&1
head -24 $OUT/dir_kt/p_kernel_stats.csv | cut -c1-150
