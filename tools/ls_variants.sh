#!/bin/bash
# Builds variants of the layer-stationary backward (extra -D flags) as lib/liblnrf_ls_<name>.so for A/B runs:
#   tools/ls_variants.sh name1:"-DFLAG1 -DFLAG2" name2:"-DFLAG3" ...
set -e
cd "$(dirname "$0")/../learn-nerf_amd/csrc"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  mkdir -p ../lib/obj_var
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c nerf_bwd_ls.hip -o ../lib/obj_var/nerf_bwd_ls_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/liblnrf_ls_$name.so ../lib/obj_var/nerf_bwd_ls_$name.o \
    $(ls ../lib/obj/*.o | grep -v -e nerf_bwd_ls.o) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  echo built liblnrf_ls_$name.so
done
