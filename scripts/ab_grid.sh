#!/bin/bash
# configs 4 and 5 with 8 / 16 / 32 / 64 workgroups per CU in the grids of the traversal kernels (RT_HIP_TRACE_BLOCKS_PER_CU).  GPU box only.
cd "$(dirname "$0")/.."
for g in ${GRIDS:-8 16 32 64}; do
  RT_HIP_TRACE_BLOCKS_PER_CU=$g timeout -k 10 300 python scripts/run_configs.py ${CFGS:-c5 c4} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('blocks per CU $g', d['config'], round(d['msamples_per_s'],1), 'Msamples/s', d['kernel_ms_one_batch'], flush=True)"
done
