for cfg in "10 2 1000 1024 0 1" "10 2 1000 1024 0 0" "10 2 1000 1024 1 1" "12 3 500 2048 1 1" "6 2 1000 1024 0 1" "6 2 1000 1024 0 0" "4 2 600 4096 1 1" "24 4 300 1024 1 1" "32 8 300 1024 1 1"; do
  DDP_OFFX="$cfg" DDP_BC_STEPS=40 DDP_BC_WARMUP=10 python profiles/bench_configs.py offX 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-26s back %.3f ms (%.3f of HBM) fwd %.3f ms  %s %s' % ('$cfg', d['back_pass_ms'], d['back_pass_frac_of_8TBs'], d['forward_ms'], d['back_pass_kernel'], d['forward_kernel']))"
done
