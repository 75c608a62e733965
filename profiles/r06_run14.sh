cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for PAD in 0 40000 60000 76000; do for B in 3072 4096 8192; do echo "pad=$PAD B=$B $(DDP_PEND_LDS_PAD=$PAD DDP_C3_B=$B DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 300 python profiles/bench_configs.py c3 2>&1 | grep -o '"back_pass_ms": [0-9.]*\|"forward_ms": [0-9.]*' | paste - -)"; done; done
