#!/bin/bash
# removal experiments on the rollout pipeline (forward_pass_pipe.hip, -DPIPE_EXP=k): who bounds a period, the chain or the helpers?
# build here (CPU container):  bash profiles/ab_pipe_exp.sh build     run on the GPU box:  bash profiles/ab_pipe_exp.sh
D=differentialdynamicprogramming.jl_amd
if [ "$1" = build ]; then
  mkdir -p gpurun_in
  for k in 1 2 3 4; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPIPE_EXP=$k -c $D/csrc/forward_pass_pipe.hip -o /tmp/pipe_exp$k.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_in/libexp$k.so $(ls $D/build/*.o | grep -v forward_pass_pipe) /tmp/pipe_exp$k.o -ldl
  done
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPIPE_PROF -c $D/csrc/forward_pass_pipe.hip -o /tmp/pipe_prof.o &&
  hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_in/libprof.so $(ls $D/build/*.o | grep -v forward_pass_pipe) /tmp/pipe_prof.o -ldl
  exit
fi
if [ "$1" = prof ]; then DDP_PIPE_PROF=1 DDP_AMD_LIB=$PWD/gpurun_in/libprof.so python profiles/ab_forward.py 2>&1 | tail -8; exit; fi
python profiles/ab_forward.py 2>&1 | tail -2
for k in 1 2 3 4; do echo "PIPE_EXP=$k"; DDP_AMD_LIB=$PWD/gpurun_in/libexp$k.so python profiles/ab_forward.py 2>&1 | tail -1; done
