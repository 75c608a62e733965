# round 5: the shared-LTI backward pass before / after the straight-line chain (old chain = -DSH_STRAIGHT=0 -DSH_SYM=4, same consumers),
# and the phase profile of the chain wave:  bash profiles/r05_sh_ab.sh   (needs build/libddp_shold.so, build/libddp_shprof.so: profiles/build_variant.sh)
mkdir -p gpurun_out
B=$PWD/differentialdynamicprogramming.jl_amd/build
rm -f gpurun_out/r05_sh_ab.txt
for v in shold ""; do
  if [ -n "$v" ]; then export DDP_AMD_LIB=$B/libddp_$v.so; else unset DDP_AMD_LIB; fi
  echo "== variant ${v:-new}" >> gpurun_out/r05_sh_ab.txt
  python profiles/ab_sh.py 1024 2048 32768 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_sh_ab.txt
done
DDP_AMD_LIB=$B/libddp_shprof.so python profiles/sh_phase_profile.py > gpurun_out/r05_sh_phase_after.json 2>&1
cat gpurun_out/r05_sh_ab.txt gpurun_out/r05_sh_phase_after.json
