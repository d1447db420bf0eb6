#!/bin/bash
# bench (config 2) and config 4 with the slots of a pass in row order / 8 x 8 tiles / 32 x 32 tiles (RT_HIP_TILES=0 / 8 / 32).  GPU box only.
cd "$(dirname "$0")/.."
for n in ${TILES:-0 8 32}; do
  RT_HIP_TILES=$n timeout -k 10 240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-exact --no-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('tiles $n bench', round(d['value'],1), 'Msamples/s', d.get('kernel_ms'), flush=True)"
  RT_HIP_TILES=$n timeout -k 10 300 python scripts/run_configs.py ${CFGS:-c4} --short 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('tiles $n', d['config'], round(d['msamples_per_s'],1), 'Msamples/s', d['kernel_ms_one_batch'], flush=True)"
done
