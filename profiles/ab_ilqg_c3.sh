#!/bin/bash
# A/B of two builds on whole C3 solves:  bash profiles/ab_ilqg_c3.sh old.so
OLD=$1
for i in 1 2; do
  for lib in "$OLD" ""; do
    if [ -n "$lib" ]; then export DDP_AMD_LIB=$lib; else unset DDP_AMD_LIB; fi
    echo "== ${lib:-new}"; python profiles/ilqg_c3.py 2>&1 | grep -E "GPU phases|inside the C call" | tail -3
  done
done
