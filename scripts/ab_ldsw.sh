#!/bin/bash
# bench.py (config 2) through the binary LDS tree and through the W-wide one, per library in build_ab/ and the in-tree one.  GPU box only.
cd "$(dirname "$0")/.."
for lib in raytracer_amd/librt_hip.so build_ab/librt_hip_*.so; do
  [ -f "$lib" ] || continue
  for w in ${MODES:-0 1 2}; do
    RT_HIP_LDS_WIDE=$w RT_HIP_LIB=$PWD/$lib timeout -k 10 240 python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-configs --no-exact --no-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib', 'wide=$w', round(d['value'],1), 'Msamples/s', d.get('kernel_ms'), 'tree', d['scene_info']['wide_nodes'], d['scene_info']['wide_levels'], flush=True)" || echo "$lib wide=$w FAILED"
  done
done
