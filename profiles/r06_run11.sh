cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for rep in 1 2; do for v in old new; do echo "$v $(DDP_BACKPASS=$v DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=24 timeout 600 python profiles/bench_configs.py c4tv 2>&1 | grep -o '"back_pass_ms": [0-9.]*\|"back_pass_kernel": "[a-z_0-9]*"' | paste - -)"; done; done
