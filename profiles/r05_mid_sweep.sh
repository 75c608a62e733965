# back_pass_mid_kernel / forward_mid_kernel at a few shapes (N = 300, B = 1 024, per-trajectory time-varying dynamics): HIP-event pass times
for nm in "24 4" "32 8" "16 2" "20 6"; do set -- $nm
  DDP_OFFC_N=$1 DDP_OFFC_M=$2 DDP_BC_STEPS=${STEPS:-60} DDP_BC_WARMUP=20 python profiles/bench_configs.py offC 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('n=%d m=%d back %.3f ms (%.3f of HBM) fwd %.3f ms  %s %s' % (d['n'], d['m'], d['back_pass_ms'], d['back_pass_frac_of_8TBs'], d['forward_ms'], d['back_pass_kernel'], d['forward_kernel']))"
done
