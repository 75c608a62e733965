cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5 -o c5 -- python $R/profiles/ilqgkl_c5.py > $R/gpurun_out/c5.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/root/repo/gpurun_out/c5/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.3f s (2 solves)" % (tot / 1e9))
    for r in rows[:14]:
        print("%-80s calls %5s avg %10.1f us total %8.2f ms %5.1f%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
find $R/gpurun_out/c5 -name "*kernel_trace.csv" -size +4M -delete
