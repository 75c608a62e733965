cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export DDP_BC_WARMUP=3 DDP_BC_STEPS=20
for V in head v000 v100 v010 v110 v111; do
export DDP_AMD_LIB=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_$V.so
echo "$V offL $(timeout 300 python profiles/bench_configs.py offL 2>&1 | grep -o '"back_pass_ms": [0-9.]*')  LTV-lims $(DDP_OFFX="10 2 1000 1024 1 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep -o '"back_pass_ms": [0-9.]*')  n12m2 $(DDP_OFFX="12 2 500 2048 1 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep -o '"back_pass_ms": [0-9.]*') n12m3 $(DDP_OFFX="12 3 500 2048 1 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep -o '"back_pass_ms": [0-9.]*') n8m2 $(DDP_OFFX="8 2 500 2048 1 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep -o '"back_pass_ms": [0-9.]*') n4m2lti $(DDP_OFFX="4 2 1000 2048 0 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep -o '"back_pass_ms": [0-9.]*\|"back_pass_kernel": "[a-z_0-9]*"' | paste - -)"
done
