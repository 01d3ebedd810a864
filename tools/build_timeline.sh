#!/bin/bash
# Debug build with in-kernel s_memtime stamps (see fused_chain.h, LNRF_TIMELINE): lib/liblnrf_timeline.so
set -e
cd "$(dirname "$0")/../learn-nerf_amd/csrc"
mkdir -p ../lib/obj_tl
hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLNRF_TIMELINE -c nerf_mlp.hip -o ../lib/obj_tl/nerf_mlp.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/liblnrf_timeline.so ../lib/obj_tl/nerf_mlp.o $(ls ../lib/obj/*.o | grep -v nerf_mlp.o)
echo built ../lib/liblnrf_timeline.so
