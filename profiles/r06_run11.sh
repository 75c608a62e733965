cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_large_state.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | grep "passed\|failed\|error" | cut -c1-200
DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 c4tv 2>&1 | grep -o '"config": "[^"]*"\|"back_pass_ms": [0-9.]*\|"back_pass_kernel": "[a-z_0-9]*"' | paste - - -
