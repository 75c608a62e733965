cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_large_state.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -4
