cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06f
timeout 600 python -m pytest tests/test_gpu_pend_chunk.py -x -q -m gpu 2>&1 | grep "passed\|failed\|rror\|assert" | tail -8 | cut -c1-300
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep "passed\|failed\|error" | tail -3 > gpurun_out/r06f/tests_full.txt; cat gpurun_out/r06f/tests_full.txt
bash profiles/pmc_config.sh r06_c3 c3 back_pass_q4c,forward_pend_row_kernel,df_pendcart_kernel > /dev/null 2>&1
bash profiles/pmc_config.sh r06_c5 c5 back_pass_q4c,forward_pend_row_kernel,fcov_q4l_kernel,kl_div_lds_kernel > /dev/null 2>&1
for c in c3 c5; do head -4 gpurun_out/r06_$c/summary.txt | cut -c1-400; grep "forward_pend_row" gpurun_out/r06_$c/summary.txt | cut -c1-250 | head -4; done
timeout 900 python profiles/bench_configs.py c3 2>&1 | grep '"config"' | cut -c1-600
