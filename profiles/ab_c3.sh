#!/bin/bash
# A/B of two builds of libddp_amd.so on the C3 pass, alternating, clocks settled:  bash profiles/ab_c3.sh old.so [c3|c4] [ENV=VAL ...]
OLD=$1; CFG=${2:-c3}; shift 2
for kv in "$@"; do export "$kv"; done
export DDP_BC_STEPS=${DDP_BC_STEPS:-300} DDP_BC_WARMUP=50 DDP_C4_SOLVE=0
for i in 1 2 3; do
  for lib in "$OLD" ""; do
    if [ -n "$lib" ]; then export DDP_AMD_LIB=$lib; else unset DDP_AMD_LIB; fi
    python profiles/bench_configs.py $CFG | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('${lib:-new}', d['back_pass_ms'], d['back_pass_ms_median'], d['back_pass_ms_min'], d['forward_ms'])"
  done
done
