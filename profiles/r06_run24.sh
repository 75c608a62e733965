cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pend_chunk.py -x -q -m gpu 2>&1 | grep "passed\|failed\|rror\|assert" | tail -8 | cut -c1-300
for B in 1024 2048 2560 3072 4096 8192; do echo "B=$B $(DDP_C3_B=$B DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 300 python profiles/bench_configs.py c3 2>&1 | grep -o '"back_pass_ms": [0-9.]*\|"forward_ms": [0-9.]*' | paste - -)  forced-on: $(DDP_PEND_CHUNK=1 DDP_C3_B=$B DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 300 python profiles/bench_configs.py c3 2>&1 | grep -o '"forward_ms": [0-9.]*') forced-off: $(DDP_PEND_CHUNK=0 DDP_C3_B=$B DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 300 python profiles/bench_configs.py c3 2>&1 | grep -o '"forward_ms": [0-9.]*')"; done
