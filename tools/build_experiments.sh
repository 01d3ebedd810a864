#!/bin/bash
# Debug build with the A/B switches of past experiments compiled in (-DLNRF_EXPERIMENTS, csrc/common.h):
# lib/liblnrf_experiments.so reads LNRF_WGRAD_INTERLEAVE, LNRF_WGRAD_BLOCK_SCALE, LNRF_WGRAD_ATOMICS, LNRF_WGRAD_PLAIN_TILES,
# LNRF_HASHGRID_LDS, LNRF_HASHGRID_XCD and LNRF_NGP_WGRAD from the environment; the product library reads none of them.
# Use it with LNRF_LIB=<path> (learn_nerf/_lib.py).
set -e
cd "$(dirname "$0")/../learn-nerf_amd/csrc"
mkdir -p ../lib/obj_exp
for f in *.hip; do
  hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DLNRF_EXPERIMENTS -c "$f" -o "../lib/obj_exp/${f%.hip}.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/liblnrf_experiments.so ../lib/obj_exp/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../lib/liblnrf_experiments.so
