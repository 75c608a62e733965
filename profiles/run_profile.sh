#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root:
#   bench (headline + machine-filling batch), rocprofv3 kernel stats, PMC HBM traffic (separate --pmc passes).
# Raw outputs land in gpurun_out/<tag>/ (scratch); the compact summaries are written to profiles/<tag>_*.{json,csv,txt}
# on the GPU box AND copied under gpurun_out/<tag>/profiles/ so they come back with the call.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT $OUT/profiles
python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
for B in 1024 32768; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b$B -o stats -- python $REPO/bench.py --steps 50 --warmup 5 --batch $B --no-cpu-baseline --no-other-configs --no-traffic --fill-batch 0 > $OUT/stats_b$B.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_b$B -o fetch -- python $REPO/bench.py --steps 3 --warmup 1 --preheat 0 --batch $B --no-cpu-baseline --no-other-configs --no-traffic --fill-batch 0 > $OUT/pmc_fetch_b$B.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_b$B -o write -- python $REPO/bench.py --steps 3 --warmup 1 --preheat 0 --batch $B --no-cpu-baseline --no-other-configs --no-traffic --fill-batch 0 > $OUT/pmc_write_b$B.log 2>&1
done
cd $REPO
python profiles/summarize_profile.py $OUT $TAG > $OUT/profiles/${TAG}_summary.txt 2>&1
cp $OUT/bench.json $OUT/profiles/${TAG}_bench.json
for B in 1024 32768; do cp $OUT/stats_b$B/stats_kernel_stats.csv $OUT/profiles/${TAG}_kernel_stats_b$B.csv 2>/dev/null; done
find $OUT -name "*kernel_trace.csv" -size +4M -delete
cat $OUT/profiles/${TAG}_summary.txt
cat $OUT/bench.json | cut -c1-2500; tail -3 $OUT/bench.err
