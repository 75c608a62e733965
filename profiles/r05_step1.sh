mkdir -p gpurun_out
B=$PWD/differentialdynamicprogramming.jl_amd/build
(python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r05_tests_full.txt
./profiles/microbench/sh_chain_floor > gpurun_out/r05_sh_chain_floor.txt 2>&1
for r in 1 2; do for v in "" shrcp1; do
  if [ -n "$v" ]; then export DDP_AMD_LIB=$B/libddp_$v.so; else unset DDP_AMD_LIB; fi
  echo "== ${v:-two Newton steps}" >> gpurun_out/r05_sh_rcp1.txt; python profiles/ab_sh.py 1024 2>&1 | grep "shared" >> gpurun_out/r05_sh_rcp1.txt
done; done
DDP_AMD_LIB=$B/libddp_shrcp1.so python -m pytest tests/test_gpu_shared_lti.py tests/test_gpu_parity.py -q 2>&1 | tail -4 >> gpurun_out/r05_sh_rcp1.txt
unset DDP_AMD_LIB
DDP_BC_STEPS=40 DDP_BC_WARMUP=10 python profiles/bench_configs.py offA offB c3 > gpurun_out/r05_off_shapes.json 2>&1
cat gpurun_out/r05_tests_full.txt gpurun_out/r05_sh_chain_floor.txt gpurun_out/r05_sh_rcp1.txt gpurun_out/r05_off_shapes.json
