#!/bin/bash
# Builds tools/exp/libnst_<name>.so with extra -D flags for conv_h2.hip (timing experiments; load with NST_LIB=...).
set -e
name=$1; shift
cd "$(dirname "$0")/../artstyletransfer_amd/csrc"
mkdir -p ../../tools/exp build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c conv_h2.hip -o build/conv_h2_$name.o
objs=$(ls build/*.hip.o build/*.cpp.o | grep -v conv_h2.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libnst_$name.so build/conv_h2_$name.o $objs
echo built tools/exp/libnst_$name.so
