"""demoQP (src/boxQP.jl:190-199: m = 500, H = M·M', bounds ±1) through ddp_boxqp_f64 (csrc/boxqp_big.hip, one work-group per problem)
next to the CPU oracle; and a batch of 256 such problems (one per CU)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ddp_amd
from oracle import oracle_ctypes as oc

rng = np.random.default_rng(0)
out = {}
for m in (100, 500):
    M = rng.standard_normal((m, m)); H = M @ M.T; g = rng.standard_normal(m); lo, up, x0 = -np.ones(m), np.ones(m), rng.standard_normal(m)
    ddp_amd.boxQP(H, g, lo, up, x0)
    t0 = time.perf_counter(); x, res, Hf, free = ddp_amd.boxQP(H, g, lo, up, x0); t1 = time.perf_counter()
    xr, rr, Hfr, fr, it = oc.boxqp(H, g, lo, up, x0); t2 = time.perf_counter()
    out["m%d" % m] = {"gpu_s": round(t1 - t0, 5), "oracle_cpu_s": round(t2 - t1, 5), "result": int(res), "iterations": int(it), "free": int(free.sum()),
                      "same_result": bool(res == rr and (free == fr).all()), "relerr_x": float(np.abs(x - xr).max() / np.abs(xr).max())}
m, cnt = 500, 256
Hs = np.empty((m, m, cnt), order="F")
for c in range(cnt):
    M = rng.standard_normal((m, m)); Hs[:, :, c] = M @ M.T
g = rng.standard_normal((m, cnt)); lo, up, x0 = -np.ones((m, cnt)), np.ones((m, cnt)), rng.standard_normal((m, cnt))
ddp_amd.boxQP(Hs, g, lo, up, x0)
t0 = time.perf_counter(); x, res, Hf, free = ddp_amd.boxQP(Hs, g, lo, up, x0); t1 = time.perf_counter()
out["batch256_m500"] = {"gpu_s_host_to_host": round(t1 - t0, 4), "results": sorted(set(int(r) for r in res))}
print(json.dumps(out))
