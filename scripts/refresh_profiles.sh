#!/bin/bash
# copies the summaries of gpurun_out/r2e* (scripts/gpu_call_r2e.sh + collect_profiles_cfg.sh r2e_c5) into profiles/r02_e_*
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/r2e
cp $R/kt_default/p_kernel_stats.csv profiles/r02_e_kernel_stats_bench_default.csv
cp $R/kt_c4/p_kernel_stats.csv profiles/r02_e_kernel_stats_bench_config4.csv
cp $R/bench_default_under_profiler.json profiles/r02_e_bench_default_under_profiler.json
cp $R/bench_c4_under_profiler.json profiles/r02_e_bench_config4_under_profiler.json
cp $R/bench_default.json profiles/r02_e_bench_default.json
cp $R/bench_driver_args.json profiles/r02_e_bench_driver_args.json
cp $R/power_probe.txt profiles/r02_e_power_probe.txt
for t in FETCH_SIZE:fetch_size WRITE_SIZE:write_size sq:sq f64:f64_mix f32:f32_mix; do a=${t%%:*}; b=${t##*:}; cp $R/pmc_$a/p_counter_collection.csv profiles/r02_e_c3_pmc_$b.csv; done
for t in FETCH_SIZE:fetch_size WRITE_SIZE:write_size sq:sq; do a=${t%%:*}; b=${t##*:}; cp $R/pmc4_$a/p_counter_collection.csv profiles/r02_e_c4_pmc_$b.csv; done
R=gpurun_out/r2e_c2; cp $R/kt/p_kernel_stats.csv profiles/r02_e_c2_kernel_stats.csv; cp $R/sq/p_counter_collection.csv profiles/r02_e_c2_pmc_sq.csv; cp $R/kbench.txt profiles/r02_e_c2_kbench.txt
R=gpurun_out/r2e_c5; cp $R/kt/p_kernel_stats.csv profiles/r02_e_c5_kernel_stats.csv; cp $R/sq/p_counter_collection.csv profiles/r02_e_c5_pmc_sq.csv; cp $R/fetch/p_counter_collection.csv profiles/r02_e_c5_pmc_fetch.csv; cp $R/write/p_counter_collection.csv profiles/r02_e_c5_pmc_write.csv; cp $R/kbench.txt profiles/r02_e_c5_kbench.txt
python3 scripts/refresh_traffic.py
