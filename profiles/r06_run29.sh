cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r06g
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep "passed\|failed\|error" | tail -3 > gpurun_out/r06g/tests_full.txt; cat gpurun_out/r06g/tests_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/r06g/smoke.txt
timeout 1200 python bench.py > gpurun_out/r06g/bench.json 2> gpurun_out/r06g/bench.err; cut -c1-250 gpurun_out/r06g/bench.json
