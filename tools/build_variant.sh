#!/bin/bash
# Builds a variant of libydet.so for A/B timing (tools/ab_conv.py):
#   tools/build_variant.sh <name> <conv|conv_pw|...> [<git rev> | -] [extra hipcc flags ...]
# -> tools/bin/libydet_<name>.so (that one source recompiled, the other objects taken from build/csrc as they are).
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; rev=${3:--}; shift; shift; shift || true
mkdir -p build/ab tools/bin
src=yolo_amd/csrc/$unit.hip
if [ "$rev" != "-" ]; then git show "$rev:yolo_amd/csrc/$unit.hip" > build/ab/${unit}_$name.hip; src=build/ab/${unit}_$name.hip; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iyolo_amd/csrc -Wno-unused-function "$@" -c $src -o build/ab/${unit}_$name.o
objs=$(ls build/csrc/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libydet_$name.so build/ab/${unit}_$name.o $objs
echo built tools/bin/libydet_$name.so
