#!/bin/bash
# Builds a variant of the engine next to the product library, for A/B runs on one box (boxes differ by a few per cent):
#   bash profiles/microbench/build_variant.sh NAME [-DFLAG=VALUE ...]   ->   gpurun_variants/librr_NAME.so
# The variant is selected at run time with RR_LIB_PATH=$PWD/gpurun_variants/librr_NAME.so (river_route_amd/_lib.py).
set -e
name=$1; shift
mkdir -p gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-result "$@" \
    river_route_amd/csrc/rr_plan.cpp river_route_amd/csrc/rr_engine.hip -o gpurun_variants/librr_$name.so
echo built gpurun_variants/librr_$name.so "$@"
