cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD/differentialdynamicprogramming.jl_amd/build
timeout 1500 python -m pytest tests/test_gpu_large_state.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_t2.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "c4" 2>&1 | tail -3 >> gpurun_out/r06_t2.txt
DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 c4tv offE > gpurun_out/r06_mf2.txt 2>&1
DDP_BACKPASS=old DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 c4tv >> gpurun_out/r06_mf2.txt 2>&1
DDP_C4_LIMS=0.05 DDP_C4_SOLVE=0 DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py c4 >> gpurun_out/r06_mf2.txt 2>&1
DDP_OFFX="48 6 300 1024 1 1" DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py offX >> gpurun_out/r06_mf2.txt 2>&1
DDP_OFFX="33 2 300 1024 1 0" DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py offX >> gpurun_out/r06_mf2.txt 2>&1
DDP_OFFX="40 4 300 1024 1 0" DDP_BC_WARMUP=8 DDP_BC_STEPS=20 timeout 600 python profiles/bench_configs.py offX >> gpurun_out/r06_mf2.txt 2>&1
{
echo "== mf2prof"
DDP_C4_SOLVE=0 DDP_BC_WARMUP=1 DDP_BC_STEPS=4 DDP_AMD_LIB=$R/libddp_mf2prof.so timeout 300 python profiles/bench_configs.py c4 2>&1 | grep "PROF" | tail -4 | sort
echo "== mf2prof offE"
DDP_BC_WARMUP=1 DDP_BC_STEPS=4 DDP_AMD_LIB=$R/libddp_mf2prof.so timeout 300 python profiles/bench_configs.py offE 2>&1 | grep "PROF" | tail -4 | sort
} > gpurun_out/r06_mf2prof.txt 2>&1
cat gpurun_out/r06_t2.txt; grep -o '"config": "[^"]*"\|"back_pass_ms": [0-9.]*\|"back_pass_kernel": "[a-z_0-9]*"' gpurun_out/r06_mf2.txt | paste - - -; cat gpurun_out/r06_mf2prof.txt
