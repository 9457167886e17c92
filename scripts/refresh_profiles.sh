#!/bin/bash
# Copies the summaries of gpurun_out/<tag>* (scripts/evidence.sh <tag>) into profiles/r<NN>_* and regenerates traffic.json.
#   usage: scripts/refresh_profiles.sh <round number> <tag>
set -e
cd "$(dirname "$0")/.."
RN=$(printf "r%02d" "$1"); TAG=$2
R=gpurun_out/$TAG
cp $R/kt_default/p_kernel_stats.csv profiles/${RN}_kernel_stats_bench_default.csv
cp $R/kt_c4/p_kernel_stats.csv profiles/${RN}_kernel_stats_bench_config4.csv
cp $R/bench_default_under_profiler.json profiles/${RN}_bench_default_under_profiler.json
cp $R/bench_c4_under_profiler.json profiles/${RN}_bench_config4_under_profiler.json
cp $R/bench_default.json profiles/${RN}_bench_default.json
cp $R/bench_driver_args.json profiles/${RN}_bench_driver_args.json
for f in power_probe polish_rate directional_bench legacy_stream_bench; do [ -f $R/$f.txt ] && cp $R/$f.txt profiles/${RN}_$f.txt; done
for t in FETCH_SIZE:fetch_size WRITE_SIZE:write_size sq:sq f64:f64_mix f32:f32_mix; do a=${t%%:*}; b=${t##*:}; cp $R/pmc_$a/p_counter_collection.csv profiles/${RN}_c3_pmc_$b.csv; done
for t in FETCH_SIZE:fetch_size WRITE_SIZE:write_size sq:sq; do a=${t%%:*}; b=${t##*:}; cp $R/pmc4_$a/p_counter_collection.csv profiles/${RN}_c4_pmc_$b.csv; done
for c in c2 c5 ring; do
  Q=gpurun_out/${TAG}_$c
  cp $Q/kt/p_kernel_stats.csv profiles/${RN}_${c}_kernel_stats.csv
  cp $Q/sq/p_counter_collection.csv profiles/${RN}_${c}_pmc_sq.csv
  cp $Q/kbench.txt profiles/${RN}_${c}_kbench.txt
done
cp gpurun_out/${TAG}_c5/fetch/p_counter_collection.csv profiles/${RN}_c5_pmc_fetch.csv
cp gpurun_out/${TAG}_c5/write/p_counter_collection.csv profiles/${RN}_c5_pmc_write.csv
RC_PROFILE_PREFIX=${RN}_ python3 scripts/refresh_traffic.py
