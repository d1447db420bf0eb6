#!/bin/bash
# bench (config 2) and configs 4, 3, 1 with a batch's passes numbered in groups of 2^n (RT_HIP_PASS_GROUP=n; 0 = pass-major).  GPU box only.
cd "$(dirname "$0")/.."
for n in ${GROUPS_LOG2:-0 2 4 6}; do
  RT_HIP_PASS_GROUP=$n timeout -k 10 240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-exact --no-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('pass group 2^$n bench', round(d['value'],1), 'Msamples/s', d.get('kernel_ms'), flush=True)"
  RT_HIP_PASS_GROUP=$n timeout -k 10 300 python scripts/run_configs.py ${CFGS:-c4 c3 c1} --short 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('pass group 2^$n', d['config'], round(d['msamples_per_s'],1), 'Msamples/s', d['kernel_ms_one_batch'], flush=True)"
done
