cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_boxqp2.py tests/test_gpu_row_shapes.py tests/test_gpu_parity.py tests/test_gpu_tile_shapes.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | grep "passed\|failed" | cut -c1-300
export DDP_BC_WARMUP=3 DDP_BC_STEPS=20
for V in base nogen nogen4; do
if [ $V != base ]; then export DDP_AMD_LIB=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_$V.so; fi
echo "$V offL $(timeout 300 python profiles/bench_configs.py offL 2>&1 | grep -o '"back_pass_ms": [0-9.]*')  LTV-lims $(DDP_OFFX="10 2 1000 1024 1 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep -o '"back_pass_ms": [0-9.]*')  n12m2 $(DDP_OFFX="12 2 500 2048 1 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep -o '"back_pass_ms": [0-9.]*')"
done
