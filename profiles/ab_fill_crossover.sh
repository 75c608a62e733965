#!/bin/bash
# Cross-over of the LTI n=10/m=2 backward kernels with the batch size: mx (one MFMA tile per trajectory), dpp (16-lane rows), dppw (rows +
# write-back waves).  Prints back_pass_ms of bench.py's machine_filling block for --fill-batch B.
for B in 2048 4096 6144 8192 12288 16384 32768 65536; do
  for v in "x 0" "dpp 0" "dpp 1"; do set -- $v
    DDP_BACKPASS=$1 DDP_DPPW=$2 python bench.py --no-traffic --no-other-configs --no-cpu-baseline --steps 5 --warmup 2 --preheat 20 --fill-batch $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        mf = json.loads(l).get('machine_filling'); print('B=$B DDP_BACKPASS=$1 DDP_DPPW=$2', mf['back_pass_kernel'], 'back %.3f ms' % mf['back_pass_ms'], 'fwd %.3f' % mf['forward_ms'], '%.0f it/s' % mf['value'])
"
  done
done
