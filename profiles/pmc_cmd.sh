#!/bin/bash
# SQ-level PMC passes for an arbitrary command:  bash profiles/pmc_cmd.sh <tag> <kernel-substring> -- <command...>
set -u
TAG=$1; KSUB=$2; shift 3
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  (cd $REPO && rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- "$@" > $OUT/p$i.log 2>&1)
done
cd $REPO
python - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, sys, collections
root, ksub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list); waves = None
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if ksub in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("  %-24s %16.1f (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
