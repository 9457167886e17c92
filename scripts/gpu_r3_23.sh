#!/bin/bash
# re-take of the headline kernel-trace profile without the appended legs (same tag directory as the full evidence run)
OUT=/root/repo/gpurun_out/${1:-r3G}; mkdir -p $OUT; rm -rf $OUT/kt_default
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt_default -o p --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-also > $OUT/bench_default_under_profiler.json 2> $OUT/kt_default.log || exit 1
head -4 $OUT/kt_default/p_kernel_stats.csv | cut -c1-170
python3 -c "
import json; d=json.loads(open('$OUT/bench_default_under_profiler.json').read().strip().splitlines()[-1]); print('kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'], d['end_to_end']['paper_philox_metrics_only'], d['end_to_end']['paper_legacy_json_cache'])"
cd /root/repo && python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err && python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
for f in bench_default bench_driver_args; do python3 -c "
import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', 'value %.4g'%d['value'], 'ms/step %.5f'%d['ms_per_step'], 'kernel %.5f'%d['roofline']['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], d['roofline']['steady_state_untimed']['kernel_ms'], d['also']['cold_20_steps_kernel_ms']['kernel_ms'], d['also']['shipped_lbfgs_controllers']['kernel_ms'], d['cpu_baseline']['value'])"; done
