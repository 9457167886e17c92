#!/bin/bash
# round 4, call 13: MT19937 raw-word kernel with the register-carried chunk recurrence (tests of the legacy stream and of the
# directional draws: state bit-identical to NumPy's; timing of the stream and of the directional pipeline; kernel traces)
R=$PWD; OUT=$R/gpurun_out/r4m; mkdir -p $OUT
python -m pytest tests -m gpu -q -x -k "legacy or directional or mcdatasim or arim or stream" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -3
timeout -k 10 200 python scripts/legacy_stream_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/legacy_stream_bench.txt
timeout -k 10 200 python scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_bench.txt
timeout -k 10 200 python scripts/directional_profile.py 2>&1 | grep -v amdgpu.ids >> $OUT/directional_bench.txt
cat $OUT/directional_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/dir_kt -o p --output-format csv -- python3 $R/scripts/directional_bench.py > $OUT/dir_kt.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/leg_kt -o p --output-format csv -- python3 $R/scripts/legacy_stream_bench.py > $OUT/leg_kt.log 2>&1
head -8 $OUT/dir_kt/p_kernel_stats.csv | cut -c1-150
head -8 $OUT/leg_kt/p_kernel_stats.csv | cut -c1-150
cd $R
SEED=4600:4619 NCFG=150 timeout -k 10 420 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | grep "legacy stream\|directional draws" | tee $OUT/fuzz_rng.txt
