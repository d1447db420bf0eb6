#!/bin/bash
# bench.py (short) once per "NAME=VALUE" argument (and once with none): A/B of run-time knobs.  GPU box only.   usage: scripts/ab_env.sh RT_HIP_SHADOW_RAYS=1 RT_HIP_LDS_WIDE=2 ...
cd "$(dirname "$0")/.."
for kv in "" "$@"; do
  env $kv timeout -k 10 240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-exact --no-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('${kv:-default}', round(d['value'],1), 'Msamples/s', d.get('kernel_ms'), flush=True)" || echo "$kv FAILED"
done
