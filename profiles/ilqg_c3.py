"""C3 end-to-end: 4096 pendcart iLQG solves (limits, boxQP) through the device-resident driver (host-pointer entry)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddp_amd
B, T = int(os.environ.get("C3_B", 4096)), 600
rng = np.random.default_rng(0)
x0 = np.tile(np.array([np.pi - 0.6, 0, 0, 0])[:, None], (1, B)); x0[0] += rng.uniform(-0.1, 0.1, B)
u0 = np.zeros((1, T, B))
kw = dict(regType=2, α=10.0 ** np.linspace(0.2, -3, 6), λmax=1e15, tol_fun=1e-8, tol_grad=1e-8, max_iter=1000)
for it in range(2):
    t = time.perf_counter()
    r = ddp_amd.iLQG(ddp_amd.PendcartProblem(), x0, u0, lims=5.0 * np.array([[-1.0, 1.0]]), **kw)
    dt = time.perf_counter() - t
    st = r[6]["stats"]
    print("C3 iLQG pendcart B=%d: %.3f s, %d batch iterations, exit reasons %s, mean iterations %.1f, mean cost %.1f"
          % (B, dt, r[6]["global_iters"], dict(zip(*np.unique(st[0].astype(int), return_counts=True))), st[1].mean(), st[7].mean()))
    tr = r[6]
    parts = {k: float(np.nansum(tr[k])) for k in ("time_derivs", "time_backward", "time_forward")}
    print("   GPU phases (HIP events, s): derivs %.3f  back %.3f  forward(+cost, %d alphas) %.3f  | other (host, H2D/D2H, state machine) %.3f"
          % (parts["time_derivs"], parts["time_backward"], len(kw["α"]), parts["time_forward"], tr["time_total"] - sum(parts.values())))
for it in range(2):
    t = time.perf_counter()
    r = ddp_amd.iLQG(ddp_amd.PendcartProblem(), x0, u0, lims=5.0 * np.array([[-1.0, 1.0]]), timing=False, **kw)
    dt = time.perf_counter() - t
    print("C3 iLQG pendcart B=%d without the time_* keys (host poll every 4th batch iteration): %.3f s, %.3f s inside the C call, %d batch iterations"
          % (B, dt, r[6]["time_total"], r[6]["global_iters"]))
