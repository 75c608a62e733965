cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
./profiles/microbench/lds_bank_patterns > gpurun_out/r06_lds_bank.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_large_state.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_t2.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c4" 2>&1 | tail -8 >> gpurun_out/r06_t2.txt
DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 c4tv offE > gpurun_out/r06_mf2.txt 2>&1
DDP_BACKPASS=old DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 >> gpurun_out/r06_mf2.txt 2>&1
DDP_C4_LIMS=0.05 DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 >> gpurun_out/r06_mf2.txt 2>&1
DDP_C4_LIMS=0.05 DDP_BACKPASS=old DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 >> gpurun_out/r06_mf2.txt 2>&1
cat gpurun_out/r06_t2.txt; cut -c1-330 gpurun_out/r06_mf2.txt
