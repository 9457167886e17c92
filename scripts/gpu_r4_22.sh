#!/bin/bash
# round 4, call 22: legacy stream kernels - jump-ahead XOR phase with scalar index loads (lane = word, wave = term slice), polar
# attempts lane-contiguous in the count / emit kernels.  Tests of the legacy stream and the directional draws (state bit-identical
# to NumPy's), same-box A/B of the stream timing against the build before (build/variants/lib_before_rng.so), kernel trace,
# the fuzz script's RNG blocks under further auxiliary seeds
R=$PWD; OUT=$R/gpurun_out/r4aa; mkdir -p $OUT
python -m pytest tests -m gpu -q -x -k "legacy or directional or mcdatasim or arim or stream" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -3
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
  echo "== before" | tee -a $OUT/legacy_stream_bench.txt
  ROBCHAR_HIP_LIB=$R/build/variants/lib_before_rng.so timeout -k 10 200 python scripts/legacy_stream_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/legacy_stream_bench.txt
  echo "== after" | tee -a $OUT/legacy_stream_bench.txt
  timeout -k 10 200 python scripts/legacy_stream_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/legacy_stream_bench.txt
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/leg_kt -o p --output-format csv -- python3 $R/scripts/legacy_stream_bench.py > $OUT/leg_kt.log 2>&1
head -9 $OUT/leg_kt/p_kernel_stats.csv | cut -c1-170
cd $R
for a in 11 12; do
  FUZZ_AUX_SEED=$a FUZZ_FUSED=0 FUZZ_DIR=0 SEED=$((5600 + a)) NCFG=50 timeout -k 10 200 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "legacy stream\|directional draws" | tee -a $OUT/fuzz_rng.txt
done
