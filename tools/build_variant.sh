#!/bin/bash
# Developer tool: build multipoint_amd/libmultipoint_hip_exp_<name>.so with extra hipcc flags for ONE source file
# (default conv_wino.hip); every other object comes from the regular build.   tools/build_variant.sh <name> "<flags>" [src]
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2; SRC=${3:-conv_wino43.hip}
B=multipoint_amd/csrc/_build
mkdir -p $B/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c multipoint_amd/csrc/$SRC -o $B/exp/${SRC%.hip}_$NAME.o
OBJS=$(ls $B/*.o | grep -v "/${SRC%.hip}.o" | grep -v "amdgcn")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o multipoint_amd/libmultipoint_hip_exp_$NAME.so $OBJS $B/exp/${SRC%.hip}_$NAME.o
echo built multipoint_amd/libmultipoint_hip_exp_$NAME.so
