#!/bin/bash
# A/B on the GPU box: tools/ab.sh NAME... runs the headline render (1080p, 1024 spp, deferred kernel) with each variant library
# (libpt_var_NAME.so; "base" = libpt_hip.so), twice each, interleaved.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for n in "$@"; do
    lib=rust-pathtracer_amd/libpt_var_$n.so; [ "$n" = base ] && lib=rust-pathtracer_amd/libpt_hip.so
    echo -n "$n: "; PT_LIB_PATH=$PWD/$lib timeout -k 5 100 python3 -u tools/adaptive_one.py 1 0.0 || exit 1
  done
done
