cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_boxqp2.py tests/test_gpu_parity.py tests/test_gpu_tile_shapes.py tests/test_gpu_row_shapes.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz_slice.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06_t7.txt
DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py offL offB > gpurun_out/r06_qp2.txt 2>&1
for x in "10 2 1000 4096 0 1" "6 2 1000 1024 0 1" "12 3 500 2048 1 1" "10 2 1000 2048 0 1" "10 2 1000 1024 1 1" "4 1 600 1024 1 1"; do
DDP_OFFX="$x" DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py offX >> gpurun_out/r06_qp2.txt 2>&1
done
cat gpurun_out/r06_t7.txt; grep -o '"config": "[^"]*"\|"batch": [0-9]*\|"back_pass_ms": [0-9.]*\|"back_pass_kernel": "[a-z_0-9<>A-Z]*"' gpurun_out/r06_qp2.txt | paste - - - -
