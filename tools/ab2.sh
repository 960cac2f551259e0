#!/bin/bash
# A/B on the GPU box: tools/ab2.sh NAME... runs a fixed-count and an adaptive 1080p render with each variant library
# (libpt_var_NAME.so; "base" = libpt_hip.so), one process per variant.
cd "$(dirname "$0")/.."
for n in "$@"; do
  lib=rust-pathtracer_amd/libpt_var_$n.so; [ "$n" = base ] && lib=rust-pathtracer_amd/libpt_hip.so
  echo "== $n"; PT_LIB_PATH=$PWD/$lib timeout -k 5 100 python3 -u tools/adaptive_repeat.py 0.0 0.01 0.0 || exit 1
done
