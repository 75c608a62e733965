"""copies the round-6 summaries that came back under gpurun_out/ into profiles/ (tracked) and merges the PMC traffic figures into pmc_traffic.json"""
import glob, json, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tj = os.path.join(root, "profiles", "pmc_traffic.json")
t = json.load(open(tj))
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "r06_*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)
    s = os.path.join(d, "summary.txt")
    if os.path.exists(s):
        shutil.copy(s, os.path.join(root, "profiles", tag + "_pmc.txt"))
    u = os.path.join(d, "pmc_traffic_update.json")
    if os.path.exists(u) and tag != "r06_c4old":
        t.update({k: v for k, v in json.load(open(u)).items() if not k.startswith("_")})
    for f in glob.glob(os.path.join(d, "profiles", "*")):
        shutil.copy(f, os.path.join(root, "profiles", os.path.basename(f)))
t["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), run tags r05 / r06 (profiles/run_all_r06.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md"
json.dump(t, open(tj, "w"), indent=1)
for f in ("r06_tests_full.txt", "r06_floors.txt", "r06_mid_sweep.txt", "r06_solves.txt", "r06_mf2_phases.txt"):
    p = os.path.join(root, "gpurun_out", f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(root, "profiles", f))
print("ok")
