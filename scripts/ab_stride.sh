#!/bin/bash
# config 4 (and 5) with the 8-wide tree's node records at 80 / 96 / 128-B stride (RT_HIP_WIDE_STRIDE).  GPU box only.
cd "$(dirname "$0")/.."
for st in ${STRIDES:-80 96 128}; do
  RT_HIP_WIDE_STRIDE=$st timeout -k 10 300 python scripts/run_configs.py ${CFGS:-c4} --short 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('node stride $st', d['config'], round(d['msamples_per_s'],1), 'Msamples/s', d['kernel_ms_one_batch'], flush=True)"
done
