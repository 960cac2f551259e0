#!/bin/bash
# A/B on the GPU box over sphere scenes: tools/ab_scenes.sh "N SUB" ... runs the fused pipeline at 1080p, 256 spp on each scene with
# libpt_hip.so and with every rust-pathtracer_amd/libpt_var_*.so present; DEFER=0|1 forces the shadow-ray form (default: per scene).
cd "$(dirname "$0")/.."
for sc in "$@"; do
  for lib in rust-pathtracer_amd/libpt_hip.so rust-pathtracer_amd/libpt_var_*.so; do
    [ -f "$lib" ] || continue
    echo -n "scene $sc $(basename $lib): "
    PT_LIB_PATH=$PWD/$lib PIPES=1 timeout -k 5 100 python3 -u tools/bigscene_spp.py $sc 1920 1080 256 2>&1 | grep "^pipe" | sed 's/; .*//' || exit 1
  done
done
