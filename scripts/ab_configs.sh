#!/bin/bash
# Runs scripts/run_configs.py (configs in $CFGS, default "c4 c5", --short) once per library in build_ab/librt_hip_*.so and the in-tree one.  GPU box only.
cd "$(dirname "$0")/.."
for lib in raytracer_amd/librt_hip.so build_ab/librt_hip_*.so; do
  [ -f "$lib" ] || continue
  RT_HIP_LIB=$PWD/$lib timeout -k 10 300 python scripts/run_configs.py ${CFGS:-c4 c5} --short 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('$lib', d['config'], round(d['msamples_per_s'],1), 'Msamples/s', d['kernel_ms_one_batch'], 'box/ray', round(d['box_tests_per_closest_ray'],1), round(d['box_tests_per_shadow_ray'],1), flush=True)" || echo "$lib FAILED"
done
