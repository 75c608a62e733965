cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
(python profiles/ilqg_c2.py; python profiles/ilqg_c3.py; python profiles/ilqgkl_c5.py; python profiles/ilqg_queue_c3.py) 2>&1 | grep -E "^C[235]|GPU phases|iterations per|live traj|^queue|^lock step|same summaries" > gpurun_out/r06_solves.txt
timeout 1200 python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cut -c1-400 gpurun_out/r06_solves.txt; cut -c1-300 gpurun_out/r06_bench_final.json
