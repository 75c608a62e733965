cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
{ ./profiles/microbench/q4c_chain_floor; echo; ./profiles/microbench/pend_row_chain_floor; } > gpurun_out/r06_c3_floor.txt 2>&1
cat gpurun_out/r06_c3_floor.txt
DDP_BC_WARMUP=20 DDP_BC_STEPS=60 timeout 600 python profiles/bench_configs.py c3 2>&1 | grep -o '"back_pass_ms": [0-9.]*\|"forward_ms": [0-9.]*' | paste - -
