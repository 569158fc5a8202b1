#!/bin/bash
# Builds a variant of libydet.so for A/B timing (tools/ab_conv.py):
#   tools/build_variant.sh <name> [<git rev of conv.hip> | -] [extra hipcc flags ...]
# -> tools/bin/libydet_<name>.so (conv.hip recompiled, the other objects taken from build/csrc as they are).
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:--}; shift; shift || true
mkdir -p build/ab tools/bin
src=yolo_amd/csrc/conv.hip
if [ "$rev" != "-" ]; then git show "$rev:yolo_amd/csrc/conv.hip" > build/ab/conv_$name.hip; src=build/ab/conv_$name.hip; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iyolo_amd/csrc -Wno-unused-function "$@" -c $src -o build/ab/conv_$name.o
objs=$(ls build/csrc/*.o | grep -v conv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libydet_$name.so build/ab/conv_$name.o $objs
echo built tools/bin/libydet_$name.so
