#!/bin/bash
# rocprofv3 kernel statistics of the non-headline configurations (C3 pendcart + limits, C4 n=64/m=8), via gpurun from the repo root.
# Output: gpurun_out/<tag>_configs/  and the compact CSV  gpurun_out/<tag>_configs/profiles/<tag>_configs_kernel_stats.csv
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/${TAG}_configs
mkdir -p $OUT/profiles
python profiles/bench_configs.py > $OUT/bench_configs.json 2> $OUT/bench_configs.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $REPO/profiles/bench_configs.py > $OUT/stats.log 2>&1
cd $REPO
cp $OUT/stats/stats_kernel_stats.csv $OUT/profiles/${TAG}_configs_kernel_stats.csv 2>/dev/null
cp $OUT/bench_configs.json $OUT/profiles/${TAG}_bench_configs.json
find $OUT -name "*kernel_trace.csv" -size +4M -delete
head -12 $OUT/profiles/${TAG}_configs_kernel_stats.csv | cut -c1-160
cat $OUT/bench_configs.json
