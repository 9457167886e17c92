#!/bin/bash
# round-2 GPU call 1: regression tests, wave-residency A/B for the spill fix, c2/c5 profiles
set -e -o pipefail
cd /root/repo
mkdir -p gpurun_out/r2c1
python -m pytest tests -m gpu -x -q > gpurun_out/r2c1/pytest.log 2>&1 || { tail -30 gpurun_out/r2c1/pytest.log; exit 1; }
tail -3 gpurun_out/r2c1/pytest.log
{
echo "== default build"; python scripts/kbench.py --shapes 5:100:10000,7:100:10000,8:100:10000 --reps 200 2>&1 | grep -v amdgpu.ids
python scripts/kbench.py --shapes 10:100:10000 --xxz --reps 200 2>&1 | grep -v amdgpu.ids
for v in n10_w4 n10_w2; do echo "== $v"; ROBCHAR_HIP_LIB=$PWD/scripts/ubench/variants/lib_$v.so python scripts/kbench.py --shapes 10:100:10000 --xxz --reps 200 2>&1 | grep -v amdgpu.ids; done
for v in n8_w5 n8_w3; do echo "== $v"; ROBCHAR_HIP_LIB=$PWD/scripts/ubench/variants/lib_$v.so python scripts/kbench.py --shapes 8:100:10000 --reps 200 2>&1 | grep -v amdgpu.ids; done
for v in n5_w4 n5_w6; do echo "== $v"; ROBCHAR_HIP_LIB=$PWD/scripts/ubench/variants/lib_$v.so python scripts/kbench.py --shapes 5:100:10000 --reps 200 2>&1 | grep -v amdgpu.ids; done
for v in n7_w4; do echo "== $v"; ROBCHAR_HIP_LIB=$PWD/scripts/ubench/variants/lib_$v.so python scripts/kbench.py --shapes 7:100:10000 --reps 200 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r2c1/variants.txt 2>&1
cat gpurun_out/r2c1/variants.txt
scripts/collect_profiles_cfg.sh r2c1/c5 10:100:10000 --xxz > gpurun_out/r2c1/c5_collect.log 2>&1
scripts/collect_profiles_cfg.sh r2c1/c2 5:100:10000 > gpurun_out/r2c1/c2_collect.log 2>&1
cat gpurun_out/r2c1/c5/kbench.txt gpurun_out/r2c1/c2/kbench.txt
