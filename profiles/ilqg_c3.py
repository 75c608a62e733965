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
# straggler cost of the lock-step batch iteration: how many iterations each trajectory needs vs the number of batch iterations
it_ = r[6]["stats"][1].astype(int)
q = np.percentile(it_, [0, 10, 25, 50, 75, 90, 99, 100]).astype(int)
print("   iterations per trajectory: min/p10/p25/p50/p75/p90/p99/max = %s; sum %d = %.1f %% of (batch iterations x B) %d"
      % ("/".join(map(str, q)), it_.sum(), 100.0 * it_.sum() / (r[6]["global_iters"] * B), r[6]["global_iters"] * B))
live = [(it_ > g).sum() for g in range(0, int(it_.max()), max(1, int(it_.max()) // 12))]
print("   live trajectories after every %d batch iterations: %s" % (max(1, int(it_.max()) // 12), live))
