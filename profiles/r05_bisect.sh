B=$PWD/differentialdynamicprogramming.jl_amd/build
for v in shallold shold shaffold shsym4; do
  echo "== $v"; DDP_AMD_LIB=$B/libddp_$v.so python -m pytest tests/test_gpu_shared_lti.py -q 2>&1 | tail -4
done
