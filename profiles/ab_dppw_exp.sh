#!/bin/bash
# Removal experiments on back_pass_dppw (B = 32 768): DDP_DPPW_EXP 0 = the product kernel, 1 = no result stores, 2 = only Vxx stored,
# 3 = the chain wave does not load its gradient entries [cx; cu] (a constant instead)
for e in ${EXPS:-0 1 2 3 0 1 2 3}; do
  DDP_DPPW_EXP=$e python bench.py --no-traffic --no-other-configs --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        mf = json.loads(l).get('machine_filling'); print('DDP_DPPW_EXP=$e back %.3f ms' % mf['back_pass_ms'], 'fwd %.3f' % mf['forward_ms'], '%.0f it/s' % mf['value'])
"
done
