cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_diff_fun.py tests/test_gpu_edge_cases.py tests/test_gpu_pipeline_kernels.py tests/test_gpu_scheduler.py tests/test_gpu_kl.py -x -q -m gpu 2>&1 | grep "passed\|failed\|error" | cut -c1-200
for B in 1024 2048 3072 4096 8192; do echo "B=$B $(DDP_C3_B=$B DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 300 python profiles/bench_configs.py c3 2>&1 | grep -o '"back_pass_ms": [0-9.]*\|"forward_ms": [0-9.]*' | paste - -)"; done
