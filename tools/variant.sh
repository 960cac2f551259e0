#!/bin/bash
# Build a variant of the library for an A/B measurement: tools/variant.sh NAME "-DFLAG ..." -> rust-pathtracer_amd/libpt_var_NAME.so
# (same objects as libpt_hip.so except pt_kernels.o, compiled with the extra flags); run with PT_LIB_PATH=that file.
set -e
cd "$(dirname "$0")/../rust-pathtracer_amd"
NAME=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -I../include \
  -ffp-contract=on -fno-slp-vectorize "$@" -c csrc/pt_kernels.hip -o build/var_$NAME.o
OBJS=$(ls build/*.o | grep -v "build/var_\|build/prof_\|build/pt_kernels.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/var_$NAME.o $OBJS -o libpt_var_$NAME.so -lz
echo built libpt_var_$NAME.so
