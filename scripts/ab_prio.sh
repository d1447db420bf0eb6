#!/bin/bash
# bench.py (config 2) with the second batch slot's stream at default / highest / lowest priority.  GPU box only.
cd "$(dirname "$0")/.."
for rep in 1 2; do for p in "" 1 -1; do
  RT_HIP_SLOT_PRIORITY=$p timeout -k 10 240 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-exact --no-strong --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('slot priority [$p]', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms per step', flush=True)"
done; done
