#!/bin/bash
# Variant of the library with pt_demosaic.hip compiled with extra flags: tools/variant_dm.sh NAME "-DFLAG ..." ->
# rust-pathtracer_amd/libpt_var_NAME.so (run with PT_LIB_PATH=that file, e.g. tools/demosaic_time.py).
set -e
cd "$(dirname "$0")/../rust-pathtracer_amd"
NAME=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -I../include \
  -ffp-contract=off "$@" -c csrc/pt_demosaic.hip -o build/var_$NAME.o
OBJS=$(ls build/*.o | grep -v "build/var_\|build/prof_\|build/pt_demosaic.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/var_$NAME.o $OBJS -o libpt_var_$NAME.so -lz
echo built libpt_var_$NAME.so
