#!/bin/bash
OUT=gpurun_out/r3n; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_mcsim.py -x -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -6
python scripts/directional_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/directional_bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/dir_trace -o t --output-format csv -- python3 /root/repo/scripts/directional_bench.py > /dev/null 2>&1
cd /root/repo
find $OUT/dir_trace -name "*kernel_stats.csv" -exec head -16 {} \; | cut -c1-170
