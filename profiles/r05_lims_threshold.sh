# cross-over of back_pass_mxg<LIMS> (DDP_BACKPASS=w) and the row kernels (=r) with control limits, by shape and batch
for cfg in ${CFGS:-"10 2 1000 2048 0 1" "10 2 1000 3072 0 1" "6 2 1000 1536 0 1" "6 2 1000 2048 0 1" "6 2 1000 3072 0 1" "4 2 600 2048 0 1" "8 4 500 1024 1 1" "8 4 500 2048 1 1" "12 3 500 2048 1 1" "12 3 500 3072 1 1" "3 1 600 2048 0 1"}; do
 for f in w r; do
  DDP_BACKPASS=$f DDP_OFFX="$cfg" DDP_BC_STEPS=30 DDP_BC_WARMUP=10 python profiles/bench_configs.py offX 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-24s force=%s back %.3f ms  %s' % ('$cfg', '$f', d['back_pass_ms'], d['back_pass_kernel']))"
 done
done
