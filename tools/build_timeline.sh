#!/bin/bash
# Debug build with in-kernel s_memtime stamps (see fused_chain.h, LNRF_TIMELINE): lib/liblnrf_timeline.so
set -e
cd "$(dirname "$0")/../learn-nerf_amd/csrc"
mkdir -p ../lib/obj_tl
for f in nerf_mlp ngp_mlp nerf_bwd_ls; do
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLNRF_TIMELINE -c $f.hip -o ../lib/obj_tl/$f.o
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/liblnrf_timeline.so ../lib/obj_tl/nerf_mlp.o ../lib/obj_tl/ngp_mlp.o ../lib/obj_tl/nerf_bwd_ls.o \
  $(ls ../lib/obj/*.o | grep -v -e nerf_mlp.o -e ngp_mlp.o -e nerf_bwd_ls.o) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../lib/liblnrf_timeline.so
