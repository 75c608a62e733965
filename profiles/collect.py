"""Copies what profiles/run_all_r04.sh / run_all_r05.sh left under gpurun_out/ (scratch, not tracked) into profiles/ (tracked, judged):
    python profiles/collect_r04.py [tag]"""
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [("%s/profiles/%s_bench.json" % (tag, tag), "%s_bench.json" % tag), ("%s/profiles/%s_summary.txt" % (tag, tag), "%s_summary.txt" % tag),
         ("%s/profiles/%s_kernel_stats_b1024.csv" % (tag, tag), "%s_kernel_stats_b1024.csv" % tag),
         ("%s/profiles/%s_kernel_stats_b32768.csv" % (tag, tag), "%s_kernel_stats_b32768.csv" % tag),
         ("%s_c4_lims.json" % tag, "%s_c4_lims.json" % tag), ("%s_solves.txt" % tag, "%s_solves.txt" % tag),
         ("%s_shared_lti_ab.txt" % tag, "%s_shared_lti_ab.txt" % tag)]
for extra in ("sh_ab.txt", "sh_stores.txt", "sh_chain_floor.txt", "sh_phase_before.json", "sh_phase_after.json", "off_shapes.json", "tests_full.txt"):
    pairs.append(("%s_%s" % (tag, extra), "%s_%s" % (tag, extra)))
for c in ("c3", "c2tv", "c4", "c5", "offA", "offB", "offC"):
    pairs.append(("%s_%s/summary.txt" % (tag, c), "%s_%s_pmc.txt" % (tag, c)))
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)
tf = os.path.join(P, "pmc_traffic.json")
cur = json.load(open(tf)) if os.path.exists(tf) else {}
for src in ["%s/profiles/pmc_traffic.json" % tag] + ["%s_%s/pmc_traffic_update.json" % (tag, c) for c in ("c3", "c2tv", "c4", "c5", "offA", "offB", "offC", "offD", "offL")]:
    s = os.path.join(G, src)
    if os.path.exists(s):
        cur.update(json.load(open(s)))
json.dump(cur, open(tf, "w"), indent=1)
print("merged pmc_traffic.json (%d keys)" % len(cur))
