#!/bin/bash
# config 5 (and whatever $CFGS names) with the any-hit kernels through the 8-wide quantised tree (RT_HIP_QUAD=0) and through the 4-wide float tree (=1), per library.  GPU box only.
cd "$(dirname "$0")/.."
for lib in raytracer_amd/librt_hip.so build_ab/librt_hip_*.so; do
  [ -f "$lib" ] || continue
  for w in ${MODES:-0 1}; do
    RT_HIP_QUAD=$w RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python scripts/run_configs.py ${CFGS:-c5} --short 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('$lib', 'quad=$w', d['config'], round(d['msamples_per_s'],1), 'Msamples/s', d['kernel_ms_one_batch'], 'box/ray', round(d['box_tests_per_closest_ray'],1), round(d['box_tests_per_shadow_ray'],1), 'prims/shadow ray', round(d['prim_tests_per_shadow_ray'],2), flush=True)" || echo "$lib quad=$w FAILED"
  done
done
