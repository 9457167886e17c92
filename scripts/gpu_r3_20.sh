#!/bin/bash
# long adversarial fuzz campaign on the final build: 200 seeds x 150 configurations, every chain / ring kernel vs the oracle
OUT=gpurun_out/r3u; mkdir -p $OUT/dump
FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=5e-11 SEED=1000:1099 NCFG=150 timeout -k 10 500 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_a.txt
FUZZ_DUMP=$OUT/dump FUZZ_DUMP_ABOVE=5e-11 SEED=1100:1199 NCFG=150 timeout -k 10 500 python scripts/fuzz_parity.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_b.txt
ls $OUT/dump | head
