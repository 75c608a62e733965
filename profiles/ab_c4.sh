#!/bin/bash
# A/B of C4 backward-pass variants inside ONE gpurun call (box-to-box variance is larger than the differences):
#   bash profiles/ab_c4.sh <variant> [<variant> ...]   — variant = name of build/libddp_<name>.so, "main" = the shipped library
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_$v.so
  [ "$v" = main ] && lib=$PWD/differentialdynamicprogramming.jl_amd/libddp_amd.so
  for rep in 1 2; do
    echo -n "$v: "
    DDP_C4_SOLVE=0 DDP_BC_WARMUP=10 DDP_AMD_LIB=$lib timeout 300 python profiles/bench_configs.py c4 2>&1 | grep -o '"back_pass_ms": [0-9.]*, "forward_ms": [0-9.]*'
  done
done
