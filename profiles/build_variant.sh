#!/bin/bash
# builds a variant of libddp_amd.so with extra compiler flags for ONE translation unit (A/B timing through DDP_AMD_LIB):
#   bash profiles/build_variant.sh <name> <file.hip> "<flags>"   ->  differentialdynamicprogramming.jl_amd/build/libddp_<name>.so
set -e
cd "$(dirname "$0")/../differentialdynamicprogramming.jl_amd"
name=$1; src=$2; flags=$3
extra=""
case $src in back_pass_mx*|back_pass_sh*|back_pass_q4*|back_pass_mfma*|back_pass_mf2*) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $extra $flags -c csrc/$src -o build/var_${name}.o
objs=$(ls build/*.o | grep -v "/var_" | grep -v "/${src%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/libddp_${name}.so $objs build/var_${name}.o -ldl
echo build/libddp_${name}.so
