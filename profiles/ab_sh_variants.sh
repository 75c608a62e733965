#!/bin/bash
# the shared-LTI backward pass under different tile / writer-wave counts, ONE gpurun call (profiles/build_variant.sh builds the libraries)
for v in "" t16w3 t16w6 t32w3 t24w8; do
  if [ -n "$v" ]; then export DDP_AMD_LIB=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_$v.so; else unset DDP_AMD_LIB; fi
  echo "== ${v:-default (t32w6)}"
  python profiles/ab_sh.py "$@" 2>/dev/null | grep shared
done
