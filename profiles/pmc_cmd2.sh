#!/bin/bash
# extra SQ counters (instruction fetch, branches, MFMA busy, scalar/VMEM issue cycles):
#   bash profiles/pmc_cmd2.sh <tag> <kernel-substring> -- <command...>
set -u
TAG=$1; KSUB=$2; shift 3
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
P2="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  (cd $REPO && rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/q$i -o q$i -- "$@" > $OUT/q$i.log 2>&1)
done
cd $REPO
python - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, sys, collections
root, ksub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("  %-28s %16.1f (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
