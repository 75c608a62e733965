cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
( cd profiles/microbench && hipcc --offload-arch=gfx950 -O3 -o /tmp/narrow_streams narrow_streams.hip 2>/dev/null && /tmp/narrow_streams ) > gpurun_out/r06_narrow_streams.txt 2>&1
bash profiles/pmc_config.sh r06_offL offL back_pass_mxg,forward_dpp > /dev/null 2>&1
timeout 1200 python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
cat gpurun_out/r06_narrow_streams.txt; head -4 gpurun_out/r06_offL/summary.txt | cut -c1-300; grep "^## back_pass" gpurun_out/r06_offL/summary.txt | cut -c1-300; cut -c1-300 gpurun_out/r06_bench_final.json
