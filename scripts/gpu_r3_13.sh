#!/bin/bash
# final delta evidence of round 3: full GPU suite + the directional pipeline after the complex symmetric QL kernel
OUT=gpurun_out/r3z2; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "amdgpu.ids\|socket.cpp\|Gloo" $OUT/pytest.log | tail -4
python3 scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/directional_bench.txt; cat $OUT/directional_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/kt_directional -o p --output-format csv -- python3 /root/repo/scripts/directional_bench.py > /dev/null 2> /root/repo/$OUT/kt_directional.log
cd /root/repo
head -14 $OUT/kt_directional/p_kernel_stats.csv | cut -c1-150
SEED=61 NCFG=2500 timeout -k 10 400 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tail -9
