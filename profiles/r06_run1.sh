cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
./profiles/microbench/lds_bank_patterns > gpurun_out/r06_lds_bank.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_shared_lti.py tests/test_gpu_row_shapes.py tests/test_gpu_bench_contract.py -x -q -m gpu -k "c4 or shared or tile or timed or forward_mid or bench_line" 2>&1 | tail -8 > gpurun_out/r06_t1.txt
DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 c4tv offE offL offB > gpurun_out/r06_base.txt 2>&1
cat gpurun_out/r06_t1.txt gpurun_out/r06_base.txt
