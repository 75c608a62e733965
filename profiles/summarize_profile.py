"""Summarises a profiles/run_profile.sh run: per-kernel average duration (rocprofv3 --stats) and HBM traffic
per launch from the FETCH_SIZE / WRITE_SIZE PMC passes, corrected as /opt/skills/guides/MI355X_MICROARCH.md
prescribes (FETCH_SIZE/WRITE_SIZE are in KiB-like 1024-byte units; on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced read stream, so it is doubled).  Also writes profiles/pmc_traffic.json entries that
bench.py picks up for `roofline.traffic`."""
import csv
import glob
import json
import os
import sys

root, tag = sys.argv[1], sys.argv[2]
KERNELS = ("sh_back_kernel", "sh_group_kernel", "back_pass_mx2_kernel", "forward_pipe4_kernel", "forward_pipe_kernel", "back_pass_mx_kernel", "back_pass_fast_kernel", "back_pass_dppw_kernel", "back_pass_dpp_kernel", "back_pass_q4p_kernel", "back_pass_q4_kernel", "back_pass_kernel", "forward_dpp_kernel", "cost_kernel", "forward_pass_kernel")


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


traffic = {}
print("== %s" % tag)
for B in (1024, 32768):
    print("-- batch %d" % B)
    f = os.path.join(root, "stats_b%d" % B, "stats_kernel_stats.csv")
    if os.path.exists(f):
        for row in csv.DictReader(open(f)):
            k = short(row["Name"])
            if k:
                print("   %-24s calls=%-4s avg=%10.1f us  total%%=%s" % (k, row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
    vals = {}
    for kind in ("fetch", "write"):
        for g in glob.glob(os.path.join(root, "pmc_%s_b%d" % (kind, B), "*counter_collection.csv")):
            acc = {}
            for row in csv.DictReader(open(g)):
                k = short(row["Kernel_Name"])
                if k:
                    acc.setdefault(k, []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                vals.setdefault(k, {})[kind] = sum(v) / len(v)
    for k, d in vals.items():
        fetch = d.get("fetch", 0.0) * 1024 * 2          # gfx950: FETCH_SIZE counts 64 B per 128-B request
        write = d.get("write", 0.0) * 1024
        print("   %-24s HBM read %.1f MB (FETCH_SIZE x2)  write %.1f MB  total %.1f MB per launch" % (k, fetch / 1e6, write / 1e6, (fetch + write) / 1e6))
        traffic["%s_bytes_per_launch_B%d" % (k, B)] = int(fetch + write)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
prev = {}
if os.path.exists(out):
    try:
        prev = json.load(open(out))
    except Exception:
        prev = {}
prev.update(traffic)
prev["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), run tag %s; FETCH_SIZE doubled per MI355X_MICROARCH.md" % tag
json.dump(prev, open(out, "w"), indent=1)
json.dump(prev, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
