#!/bin/bash
# Runs bench.py once per library in build_ab/librt_hip_*.so (and the in-tree one) and prints value + per-kernel ms.  GPU box only.
cd "$(dirname "$0")/.."
for lib in raytracer_amd/librt_hip.so build_ab/librt_hip_*.so; do
  [ -f "$lib" ] || continue
  RT_HIP_LIB=$PWD/$lib timeout -k 10 240 python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-configs --no-exact --no-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib', round(d['value'],1), 'Msamples/s', d.get('kernel_ms'), 'frac', round(d['roofline']['frac'],4) if 'roofline' in d else None, flush=True)" || echo "$lib FAILED"
done
