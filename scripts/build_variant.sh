#!/bin/bash
# Builds build_ab/librt_hip_<name>.so with extra hipcc flags, for A/B runs with RT_HIP_LIB (the variants compile side by side).
# usage: scripts/build_variant.sh <name> [-DRT_...=...]...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=build_ab/$name; mkdir -p $out
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Wno-unused-function $*"
pids=()
for f in rt_lib rt_lbvh rt_ploc rt_sah rt_wide rt_variant_spheres rt_variant_mesh rt_variant_boxes rt_variant_all; do
  /opt/rocm/bin/hipcc $FLAGS -c -o $out/$f.o raytracer_amd/csrc/$f.hip & pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -c -o $out/rt_bvh.o raytracer_amd/csrc/rt_bvh.cpp & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc $FLAGS -shared -o build_ab/librt_hip_$name.so $out/*.o
rm -rf $out
echo built build_ab/librt_hip_$name.so
