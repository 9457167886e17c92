#!/bin/bash
# round 4, call 5: the measurements whose outputs were lost with the first session's container - upper bound of what deferring
# the stepping path could save (lib_nostep), Philox generator with / without its HBM write, ring repair grid width, the
# directional pipeline (stage timing, bench, kernel trace), one adversarial fuzz campaign block per run
R=$PWD; OUT=$R/gpurun_out/r4e; mkdir -p $OUT
kb() { timeout -k 10 200 python scripts/kbench.py --reps 300 "$@" 2>&1 | grep "N="; }
for round in 1 2; do
  for v in few nostep; do
    export ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so
    echo "== $v" >> $OUT/ab_nostep.txt
    kb --shapes 7:100:10000 >> $OUT/ab_nostep.txt
    kb --shapes 10:100:10000 --xxz >> $OUT/ab_nostep.txt
    kb --shapes 7:100:10000 --shipped >> $OUT/ab_nostep.txt
    kb --shapes 7:100:10000 --out 3 >> $OUT/ab_nostep.txt
  done
done
for v in few philox_nostore few philox_nostore; do
  ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so timeout -k 10 120 python scripts/philox_bench.py 2>&1 | grep philox >> $OUT/philox.txt
done
unset ROBCHAR_HIP_LIB
for g in 1024 64 8 1024 64 8; do
  echo "== RC_RING_REPAIR_GRID=$g" >> $OUT/ring_grid.txt
  RC_RING_REPAIR_GRID=$g timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 7:100:10000 --ring 2>&1 | grep "N=" >> $OUT/ring_grid.txt
done
timeout -k 10 200 python scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_bench.txt
timeout -k 10 200 python scripts/directional_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_profile.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/dir_kt -o p --output-format csv -- python3 $R/scripts/directional_bench.py > $OUT/dir_kt.log 2>&1
cd $R
cat $OUT/ab_nostep.txt $OUT/philox.txt $OUT/ring_grid.txt $OUT/directional_bench.txt $OUT/directional_profile.txt
head -30 $OUT/dir_kt/p_kernel_stats.csv | cut -c1-150
for r in 4000:4099 4100:4199; do
  FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=1e-11 SEED=$r NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/fuzz.txt
done
