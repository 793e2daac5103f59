#!/bin/bash
# Timing-only builds of the f16x2 conv kernel with one memory stream dropped each (conv_h2.hip: NST_H2_ABLATE), linked
# into tools/exp/libnst_hip_abl<bits>.so (tools/exp/ is git-ignored and outside the package: nothing there is ever loaded by
# default); then on the GPU box:
#   for b in 0 1 2 4 8 16; do NST_LIB=$PWD/tools/exp/libnst_hip_abl$b.so python tools/layer_times.py; done
# The results of such a build are WRONG by construction.
set -e
cd "$(dirname "$0")/../artstyletransfer_amd/csrc"
make -j4 >/dev/null
mkdir -p ../../tools/exp
for bits in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNST_H2_ABLATE=$bits -c conv_h2.hip -o /tmp/conv_h2_abl$bits.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libnst_hip_abl$bits.so $(ls build/*.o | grep -v conv_h2) /tmp/conv_h2_abl$bits.o -ldl
  echo built libnst_hip_abl$bits.so
done
