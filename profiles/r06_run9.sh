cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_tests_full.txt 2>&1
tail -5 gpurun_out/r06_tests_full.txt | cut -c1-200
{ ./profiles/microbench/q4c_chain_floor; echo; ./profiles/microbench/pend_row_chain_floor; } > gpurun_out/r06_c3_floor.txt 2>&1
DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 600 python profiles/bench_configs.py c3 c5 2>&1 | grep -o '"config": "[^"]*"\|"back_pass_ms": [0-9.]*\|"forward_ms": [0-9.]*' | paste - - - >> gpurun_out/r06_c3_floor.txt
cat gpurun_out/r06_c3_floor.txt
