#!/bin/bash
# round 4, call 2: full GPU suite; guard level A/B (three moments / m = 0 only / none); upper bound of what deferring the
# stepping path could save (lib_nostep); the Philox generator with and without its HBM write; ring repair grid width;
# the directional pipeline (bench + kernel trace)
R=$PWD; OUT=$R/gpurun_out/r4b; mkdir -p $OUT
python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc
tail -8 $OUT/pytest.log
kb() { timeout -k 10 200 python scripts/kbench.py --reps 300 "$@" 2>&1 | grep "N="; }
for round in 1 2; do
  for v in noguard few m0; do
    export ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so
    echo "== $v" >> $OUT/ab_guard.txt
    kb --shapes 7:100:10000 --out 3 >> $OUT/ab_guard.txt
    kb --shapes 5:100:10000,7:100:10000,10:100:10000 --ring >> $OUT/ab_guard.txt
  done
  for v in few nostep; do
    export ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so
    echo "== $v" >> $OUT/ab_nostep.txt
    kb --shapes 7:100:10000 >> $OUT/ab_nostep.txt
    kb --shapes 10:100:10000 --xxz >> $OUT/ab_nostep.txt
    kb --shapes 7:100:10000 --shipped >> $OUT/ab_nostep.txt
  done
done
for v in few philox_nostore few philox_nostore; do
  ROBCHAR_HIP_LIB=$R/build/variants/lib_$v.so timeout -k 10 120 python scripts/philox_bench.py 2>&1 | grep philox >> $OUT/philox.txt
done
unset ROBCHAR_HIP_LIB
for g in 1024 64 1024 64; do
  echo "== RC_RING_REPAIR_GRID=$g" >> $OUT/ring_grid.txt
  RC_RING_REPAIR_GRID=$g timeout -k 10 200 python scripts/kbench.py --reps 300 --shapes 7:100:10000 --ring 2>&1 | grep "N=" >> $OUT/ring_grid.txt
done
timeout -k 10 200 python scripts/directional_bench.py 2>&1 | grep directional > $OUT/directional_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/dir_kt -o p --output-format csv -- python3 $R/scripts/directional_bench.py > $OUT/dir_kt.log 2>&1
cd $R
cat $OUT/ab_guard.txt $OUT/ab_nostep.txt $OUT/philox.txt $OUT/ring_grid.txt $OUT/directional_bench.txt
ls $OUT/dir_kt/* | head
