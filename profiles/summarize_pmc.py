"""Summarises rocprofv3 --pmc CSV output: mean counter value per kernel per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        short = "back_pass_fast" if "back_pass_fast" in name else "back_pass" if "back_pass" in name else \
                "forward_pass" if "forward_pass" in name else None
        if short:
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print("  %-28s %14.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
