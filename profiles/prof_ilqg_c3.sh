cd /tmp && export TMPDIR=/tmp
R=/root/repo
C3_PROFILE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3solve -o c3 -- python $R/profiles/ilqg_c3.py > $R/gpurun_out/c3solve.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/root/repo/gpurun_out/c3solve/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.3f s" % (tot / 1e9))
    for r in rows[:25]:
        print("%-90s calls %6s avg %10.1f us total %8.2f ms %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
find $R/gpurun_out/c3solve -name "*kernel_trace.csv" -size +4M -delete
