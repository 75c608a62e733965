#!/bin/bash
# Evidence for one non-headline config (C3 / C4) in ONE gpurun call: un-profiled HIP-event pass time, rocprofv3 kernel statistics,
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) and SQ counters of the backward kernel.
#   bash profiles/pmc_config.sh <tag> <c3|c4|c2tv|c5> <kernel-substring[,more,...]> [ENV=VAL ...]
# The first substring names the backward kernel (its traffic goes to pmc_traffic.json); the others get the same counter tables.
# Output: gpurun_out/<tag>/summary.txt (copy into profiles/ to have it judged)
set -u
TAG=$1; CFG=$2; KSUB=$3; shift 3
for kv in "$@"; do export "$kv"; done
export DDP_C4_SOLVE=0
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
DDP_BC_STEPS=120 DDP_BC_WARMUP=40 python profiles/bench_configs.py $CFG > $OUT/events.json 2> $OUT/events.err     # clocks settled
cd /tmp && export TMPDIR=/tmp
export DDP_BC_STEPS=480 DDP_BC_WARMUP=40                     # kernel statistics: warm, >= 100 launches (C4 runs a quarter of the steps)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $REPO/profiles/bench_configs.py $CFG > $OUT/stats.log 2>&1
export DDP_BC_STEPS=12 DDP_BC_WARMUP=2                       # the counter runs: few launches (every launch is serialised by the counters)
P0="FETCH_SIZE"
P1="WRITE_SIZE"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P3="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_THREAD_CYCLES_VALU"
P4="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P0" "$P1" "$P2" "$P3" "$P4"; do
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $REPO/profiles/bench_configs.py $CFG > $OUT/p$i.log 2>&1
  i=$((i+1))
done
cd $REPO
python - "$OUT" "$KSUB" "$CFG" > $OUT/summary.txt <<'PY'
import csv, glob, json, os, sys, collections
root, ksubs, cfg = sys.argv[1:4]
ksubs = ksubs.split(",")
print("# config %s, kernel substrings %r" % (cfg, ksubs))
print("## un-profiled HIP-event pass times (profiles/bench_configs.py, 120 passes after 40 warm-up passes)")
ev = open(os.path.join(root, "events.json")).read().strip()
print(ev)
print("## rocprofv3 --kernel-trace --stats (warm: 40 warm-up + 480 timed passes, C4 a quarter of that; top kernels)")
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 8 or any(k in row["Name"] for k in ksubs):
            print("  %-110s calls %4s avg %12.1f ns" % (row["Name"][:110], row["Calls"], float(row["AverageNs"])))
upd = {}
for ki, ksub in enumerate(ksubs):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if ksub in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    if not acc:
        print("## no counter rows for %r" % ksub)
        continue
    print("## PMC, mean per launch of kernels matching %r" % ksub)
    for c, v in sorted(acc.items()):
        print("  %-26s %18.1f (n=%d)" % (c, sum(v) / len(v), len(v)))
    mean = lambda k: sum(acc[k]) / len(acc[k])
    if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
        fs, ws = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        print("## %s HBM traffic per launch: FETCH_SIZE KiB x1024 x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KiB x1024 = %.1f MB"
              % (ksub, (2 * fs + ws) * 1024 / 1e6))
        batch = json.loads(ev.splitlines()[0])["batch"]
        key = "%s_back_pass_bytes_per_launch_B%d" % (cfg.upper(), batch) if ki == 0 else "%s_%s_bytes_per_launch_B%d" % (cfg.upper(), ksub, batch)
        upd[key] = int((2 * fs + ws) * 1024)
    if acc.get("SQ_WAVES") and acc.get("SQ_INSTS_VALU"):
        w = mean("SQ_WAVES")
        print("## %s per wave: VALU %.0f (MFMA f64 %.0f), SALU %.0f, LDS %.0f, VMEM rd %.0f wr %.0f, SMEM %.0f; SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f"
              % (ksub, mean("SQ_INSTS_VALU") / w, (mean("SQ_INSTS_VALU_MFMA_F64") if acc.get("SQ_INSTS_VALU_MFMA_F64") else 0) / w,
                 mean("SQ_INSTS_SALU") / w, mean("SQ_INSTS_LDS") / w, mean("SQ_INSTS_VMEM_RD") / w, mean("SQ_INSTS_VMEM_WR") / w,
                 (mean("SQ_INSTS_SMEM") if acc.get("SQ_INSTS_SMEM") else 0) / w,
                 mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES") if acc.get("SQ_WAIT_ANY") and acc.get("SQ_WAVE_CYCLES") else float("nan")))
    if acc.get("SQ_VALU_MFMA_BUSY_CYCLES") and acc.get("GRBM_GUI_ACTIVE") and sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the 1024 matrix pipes (check: = MFMA instructions x their pass cycles);
        # GRBM_GUI_ACTIVE sums the kernel's cycles over the 8 XCDs
        frac = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * mean("GRBM_GUI_ACTIVE") / 8.0)
        print("## %s MFMA pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = %.3f" % (ksub, frac))
        if ki == 0:
            upd["%s_back_pass_mfma_busy_frac" % cfg.upper()] = round(frac, 4)
if upd:
    for tf in (os.path.join(os.path.dirname(root.rstrip("/")), "..", "profiles", "pmc_traffic.json"), os.path.join(root, "pmc_traffic_update.json")):
        try:
            prev = json.load(open(tf)) if os.path.exists(tf) else {}
            prev.update(upd)
            json.dump(prev, open(tf, "w"), indent=1)
        except Exception as exc:
            print("## could not update", tf, exc)
PY
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +4M -delete
