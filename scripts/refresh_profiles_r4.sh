#!/bin/bash
# copies the summaries of gpurun_out/<tag>* (scripts/gpu_call_r4_final.sh <tag>) into profiles/r04_* and regenerates traffic.json
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r4z}
R=gpurun_out/$TAG
cp $R/kt_default/p_kernel_stats.csv profiles/r04_kernel_stats_bench_default.csv
cp $R/kt_c4/p_kernel_stats.csv profiles/r04_kernel_stats_bench_config4.csv
cp $R/bench_default_under_profiler.json profiles/r04_bench_default_under_profiler.json
cp $R/bench_c4_under_profiler.json profiles/r04_bench_config4_under_profiler.json
cp $R/bench_default.json profiles/r04_bench_default.json
cp $R/bench_driver_args.json profiles/r04_bench_driver_args.json
cp $R/bench_four_ranks_gloo.json profiles/r04_bench_four_ranks_gloo_self_launched.json
cp $R/power_probe.txt profiles/r04_power_probe.txt
cp $R/ring_kbench.txt profiles/r04_ring_kbench.txt
cp $R/polish_rate.txt profiles/r04_polish_rate.txt
cp $R/directional_bench.txt profiles/r04_directional_bench.txt
cp $R/kt_directional/p_kernel_stats.csv profiles/r04_directional_kernel_stats.csv
for t in FETCH_SIZE:fetch_size WRITE_SIZE:write_size sq:sq f64:f64_mix f32:f32_mix; do a=${t%%:*}; b=${t##*:}; cp $R/pmc_$a/p_counter_collection.csv profiles/r04_c3_pmc_$b.csv; done
for t in FETCH_SIZE:fetch_size WRITE_SIZE:write_size sq:sq; do a=${t%%:*}; b=${t##*:}; cp $R/pmc4_$a/p_counter_collection.csv profiles/r04_c4_pmc_$b.csv; done
for c in c2 c5 ring; do
  Q=gpurun_out/${TAG}_$c
  cp $Q/kt/p_kernel_stats.csv profiles/r04_${c}_kernel_stats.csv
  cp $Q/sq/p_counter_collection.csv profiles/r04_${c}_pmc_sq.csv
  cp $Q/kbench.txt profiles/r04_${c}_kbench.txt
done
cp gpurun_out/${TAG}_c5/fetch/p_counter_collection.csv profiles/r04_c5_pmc_fetch.csv
cp gpurun_out/${TAG}_c5/write/p_counter_collection.csv profiles/r04_c5_pmc_write.csv
RC_PROFILE_PREFIX=r04_ python3 scripts/refresh_traffic.py
