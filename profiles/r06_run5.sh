cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD/differentialdynamicprogramming.jl_amd/build
{
echo "== mf2exp8 (slot C = wait for the Jacobian, fstore = LDS writes only)"
DDP_C4_SOLVE=0 DDP_BC_WARMUP=1 DDP_BC_STEPS=4 DDP_AMD_LIB=$R/libddp_mf2exp8.so timeout 300 python profiles/bench_configs.py c4 2>&1 | grep "PROF" | tail -4 | sort
} > gpurun_out/r06_mf2exp8.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_tests_full.txt
cat gpurun_out/r06_mf2exp8.txt gpurun_out/r06_tests_full.txt
