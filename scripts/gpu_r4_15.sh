#!/bin/bash
# round 4, call 15: the driver's 20-step window bracket by bracket (roofline.kernel_ms_per_bracket), four runs; GROUP = 20 for comparison
for i in 1 2 3 4; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('kernel_ms %.5f'%r['kernel_ms'], 'brackets', r['kernel_ms_per_bracket'], 'steady', '%.5f'%r['steady_state_untimed']['kernel_ms'], 'ms/step %.5f'%d['ms_per_step'])"; done
for i in 1 2; do ROBCHAR_BENCH_GROUP=20 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('GROUP=20: kernel_ms %.5f'%r['kernel_ms'], 'brackets', r['kernel_ms_per_bracket'], 'ms/step %.5f'%d['ms_per_step'])"; done
