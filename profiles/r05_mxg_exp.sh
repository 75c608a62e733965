#!/bin/bash
# removal experiments of back_pass_mxg_kernel (wrong results, timing only): libddp_mxg{1,2,3,4}.so = profiles/build_variant.sh mxgE back_pass_mxg.hip "-DMXG_EXP=E"
#   1 no result stores, 2 no operand refills, 3 both, 4 no transpose round trip, 9 the results as two contiguous 16-byte-per-lane stores (garbage).   bash profiles/r05_mxg_exp.sh
B=$PWD/differentialdynamicprogramming.jl_amd/build
export DDP_BC_STEPS=20 DDP_BC_WARMUP=5 DDP_OFFC_T=500 DDP_BACKPASS=wtile
for v in ${VARIANTS:-"" mxg1 mxg2 mxg3 mxg4 mxg9}; do
  if [ -n "$v" ]; then export DDP_AMD_LIB=$B/libddp_$v.so; else unset DDP_AMD_LIB; fi
  for nm in "12 3" "8 4"; do set -- $nm; for Bs in 1024 2048; do
    DDP_OFFC_N=$1 DDP_OFFC_M=$2 DDP_OFFC_B=$Bs python profiles/bench_configs.py offC 2>&1 | grep -E "config|rror" | python profiles/fmt_line.py "B=$Bs variant=${v:-production}"
  done; done
done
