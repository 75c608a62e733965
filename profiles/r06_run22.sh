cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06e
( export DDP_AMD_LIB=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_mxgprof.so DDP_BC_WARMUP=1 DDP_BC_STEPS=1
  echo "# phase profile of back_pass_mxg<12, LIMS> (s_memtime ticks per time step, trajectory 0; build -DDDP_MXGPROF) at offL: n=10 m=2 N=1000 B=1024 LTI, limits +-0.05"
  timeout 300 python profiles/bench_configs.py offL 2>&1 | grep "MXGPROF" | tail -1 ) > gpurun_out/r06e/mxg_phases.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep "passed\|failed\|error" | tail -3 > gpurun_out/r06e/tests_full.txt
export DDP_BC_WARMUP=3 DDP_BC_STEPS=20
for V in base head base head; do
if [ $V != base ]; then export DDP_AMD_LIB=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_$V.so; else unset DDP_AMD_LIB; fi
echo "$V offA $(timeout 300 python profiles/bench_configs.py offA 2>&1 | grep -o '"back_pass_ms": [0-9.]*') c2tv $(timeout 300 python profiles/bench_configs.py c2tv 2>&1 | grep -o '"back_pass_ms": [0-9.]*') offL $(timeout 300 python profiles/bench_configs.py offL 2>&1 | grep -o '"back_pass_ms": [0-9.]*') n12m3B2048 $(DDP_OFFX="12 3 500 2048 1 1" timeout 300 python profiles/bench_configs.py offX 2>&1 | grep -o '"back_pass_ms": [0-9.]*')"
done > gpurun_out/r06e/ab.txt
cat gpurun_out/r06e/tests_full.txt gpurun_out/r06e/mxg_phases.txt gpurun_out/r06e/ab.txt | cut -c1-400
