#!/bin/bash
# a build-time variant of one translation unit linked into a library of its own (A/B runs with EU_HIP_LIB):
#   tools/mkvariant.sh NAME SRC "-DFLAG=..."   ->  envutil_amd/build/libeu_hip_NAME.so ; ISA in /tmp/isa/var_NAME/
set -e
cd "$(dirname "$0")/../envutil_amd"
NAME=$1; SRC=${2:-eu_render4}; FLAGS=$3
T=/tmp/isa/var_$NAME; mkdir -p $T
/opt/rocm/bin/hipcc $FLAGS -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-pass-failed -save-temps=obj -c csrc/$SRC.hip -o $T/$SRC.o
OBJS=$(ls build/eu_*.o | grep -v "build/${SRC}.o" | grep -v "_ballot\|_stamps\|_o[0-9]")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libeu_hip_$NAME.so $OBJS $T/$SRC.o
echo "built envutil_amd/build/libeu_hip_$NAME.so"
