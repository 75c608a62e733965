#!/bin/bash
# A/B of two builds of libddp_amd.so on the headline pass (C2, B=1024), alternating, clocks settled:  bash profiles/ab_c2.sh old.so [ENV=VAL ...]
OLD=$1; shift
for kv in "$@"; do export "$kv"; done
for i in 1 2 3; do
  for lib in "$OLD" ""; do
    if [ -n "$lib" ]; then export DDP_AMD_LIB=$lib; else unset DDP_AMD_LIB; fi
    python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-other-configs --fill-batch 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('${lib:-new}', 'ms_per_step', d['ms_per_step'], 'back', r['avg_launch_ms'], 'forward', r['forward_kernels']['avg_launch_ms'], 'Mit/s', round(d['value']/1e6,4))"
  done
done
