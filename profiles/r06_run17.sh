cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export DDP_AMD_LIB=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_mxgprof.so
export DDP_BC_WARMUP=1 DDP_BC_STEPS=2
echo "== offL (LTI, lims)"; timeout 300 python profiles/bench_configs.py offL 2>&1 | grep "MXGPROF\|QP2STATS" | tail -2 | cut -c1-600
echo "== LTV lims"; DDP_OFFX="10 2 1000 1024 1 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep "MXGPROF\|QP2STATS" | tail -2 | cut -c1-600
