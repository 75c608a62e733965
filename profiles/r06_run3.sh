cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD/differentialdynamicprogramming.jl_amd/build
{
for v in "$@"; do
echo "== $v"
DDP_C4_SOLVE=0 DDP_BC_WARMUP=1 DDP_BC_STEPS=4 DDP_AMD_LIB=$R/libddp_$v.so timeout 300 python profiles/bench_configs.py c4 2>&1 | grep "PROF\|back_pass_ms" | tail -5 | sort | cut -c1-260
done
} > gpurun_out/r06_mf2prof.txt 2>&1
cat gpurun_out/r06_mf2prof.txt
