#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: bench + rocprofv3 kernel stats + PMC HBM traffic.
# Outputs land in gpurun_out/<tag>/ (scratch); the summaries worth judging are copied to profiles/ afterwards.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 3 --batch 8192 --no-cpu-baseline > $OUT/bench_b8192.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -30
find $OUT -name "*kernel_trace.csv" -size +8M -delete
cat $OUT/bench.json; cat $OUT/bench_b8192.json; tail -3 $OUT/bench.err
