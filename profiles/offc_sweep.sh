#!/bin/bash
# usage: offc_sweep.sh "n m" ... ; env FORCES="tile row", BS="1024 2048", DDP_OFFC_T, DDP_OFFC_LTI: backward kernels side by side at the offC bench shape
export DDP_BC_STEPS=${DDP_BC_STEPS:-20} DDP_BC_WARMUP=${DDP_BC_WARMUP:-5}
for nm in "$@"; do set -- $nm
  for B in ${BS:-1024}; do for f in ${FORCES:-default}; do
    [ "$f" = default ] && unset DDP_BACKPASS || export DDP_BACKPASS=$f
    DDP_OFFC_N=$1 DDP_OFFC_M=$2 DDP_OFFC_B=$B python profiles/bench_configs.py offC 2>&1 | grep -E "config|rror" | python profiles/fmt_line.py "B=$B force=$f"
  done; done
done
