# where does a global iteration of the slot scheduler go?  rocprofv3 kernel trace of the 32 768-pendulum queue run (queue part only):
# per-kernel totals and the time the GPU ran nothing (gaps between consecutive kernels on the timeline)
REPO=$(pwd); OUT=$REPO/gpurun_out/r05_queue; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DDP_QUEUE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o q -- python $REPO/profiles/ilqg_queue_c3.py > $OUT/run.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the second queue call (warm): the last sched_take-bounded block
takes = [i for i, r in enumerate(rows) if "sched_take" in r[2]]
# split the two calls at the largest gap between consecutive sched_take kernels
gaps = [(rows[takes[i + 1]][0] - rows[takes[i]][1], i) for i in range(len(takes) - 1)]
cut = max(gaps)[1]
sel = rows[takes[cut + 1]:]
t0, t1 = sel[0][0], max(r[1] for r in sel)
busy = 0; cur_end = t0; idle = 0
per = collections.Counter(); cnt = collections.Counter()
for s, e, nm in sel:
    if s > cur_end: idle += s - cur_end
    cur_end = max(cur_end, e)
    key = nm.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    per[key] += e - s; cnt[key] += 1
print("second queue call: %.1f ms on the timeline, %.1f ms with no kernel running (%.1f %%), %d kernels, %d sched_take (global iterations)" % ((t1 - t0) / 1e6, idle / 1e6, 100.0 * idle / (t1 - t0), len(sel), sum(1 for r in sel if "sched_take" in r[2])))
for k, v in per.most_common(14):
    print("  %-42s %8.1f ms  %6d launches  %7.1f us each" % (k, v / 1e6, cnt[k], v / 1e3 / cnt[k]))
PY
