cd /tmp && export TMPDIR=/tmp DDP_C4_SOLVE=0
R=$GRAFT_REPO_ROOT
for L in "" 0.05; do
  export DDP_C4_LIMS=$L
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/ic$L -o ic -- python $R/profiles/bench_configs.py c4 > /dev/null 2>&1
  python - /tmp/ic$L <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "back_pass_mfma" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
done
