#!/bin/bash
# Usage: tools/build_variant.sh <name> [-DMACRO=..]...   -> variants/librt_hip_<name>.so (not shipped; experiments only)
set -e
NAME=$1; shift
mkdir -p variants
C=ray-tracer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-function "$@" -Iinclude -I$C \
  $C/rt_api.hip $C/rt_shade_fast.hip $C/scene_pack.cpp $C/bvh_build.cpp $C/bvh_sbvh.cpp $C/bvh_sweep.cpp -o variants/librt_hip_$NAME.so -lpthread
echo built variants/librt_hip_$NAME.so
