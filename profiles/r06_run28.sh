cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pend_chunk.py tests/test_gpu_parity.py tests/test_gpu_diff_fun.py -x -q -m gpu 2>&1 | grep "passed\|failed\|rror\|assert" | tail -5 | cut -c1-300
DDP_PEND_CHUNK=1 timeout 900 python -m pytest tests/test_gpu_scheduler.py tests/test_gpu_kl.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | grep "passed\|failed\|rror\|assert" | tail -5 | cut -c1-300
for V in base prev base prev; do
if [ $V != base ]; then export DDP_AMD_LIB=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_$V.so; else unset DDP_AMD_LIB; fi
for B in 1024 2048 4096 8192; do echo "$V B=$B on: $(DDP_PEND_CHUNK=1 DDP_C3_B=$B DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 300 python profiles/bench_configs.py c3 2>&1 | grep -o '"forward_ms": [0-9.]*') off: $(DDP_PEND_CHUNK=0 DDP_C3_B=$B DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 300 python profiles/bench_configs.py c3 2>&1 | grep -o '"forward_ms": [0-9.]*')"; done; done
