#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_c4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $REPO/profiles/bench_configs.py c4 > $OUT/p$i.log 2>&1
done
cd $REPO
python - "$OUT" back_pass_mfma <<'PY'
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
