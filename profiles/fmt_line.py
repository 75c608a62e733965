"""one short line per JSON line of profiles/bench_configs.py on stdin (sweeps): config, tag, backward ms, fraction of 8 TB/s, rollout ms, kernel"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for l in sys.stdin:
    try:
        d = json.loads(l)
        print(d["config"], tag, "back", d["back_pass_ms"], d["back_pass_frac_of_8TBs"], "fwd", d["forward_ms"], d.get("back_pass_kernel"))
    except Exception:
        if "rror" in l:
            print(l[:300].rstrip())
