"""C3 end-to-end: 4096 pendcart iLQG solves (limits, boxQP) through the device-resident driver (host-pointer entry)."""
import os, sys, time
import numpy as np
import torch      # (before the library initialises HIP: the other order leaves torch without a device)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddp_amd
B, T = int(os.environ.get("C3_B", 4096)), 600
rng = np.random.default_rng(0)
x0 = np.tile(np.array([np.pi - 0.6, 0, 0, 0])[:, None], (1, B)); x0[0] += rng.uniform(-0.1, 0.1, B)
u0 = np.zeros((1, T, B))
x0, u0 = np.asfortranarray(x0), np.asfortranarray(u0)      # the layout a Julia caller has (the mirror transposes a C-ordered array first)
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

# device-resident: operands and results stay in HBM (what a resident pipeline pays; the host-pointer entry adds ~0.9 GB of results over PCIe)
import ctypes as C
from ddp_amd import _lib
dev = torch.device("cuda", 0); L = _lib.lib(); h = ddp_amd.default_handle()
prob = ddp_amd.PendcartProblem()
f64 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))).to(dev)
p_ = lambda t: C.c_void_p(t.data_ptr())
n, m = 4, 1
dQ, dR, dx0, du0, dl = map(f64, (prob.Q, prob.R, x0, u0, 5.0 * np.array([[-1.0, 1.0]])))
pr = _lib.Problem(); pr.kind, pr.n, pr.m, pr.N, pr.B = 1, n, m, T, B
pr.Q, pr.R = dQ.data_ptr(), dR.data_ptr()
pr.g, pr.l, pr.h, pr.d = prob.g, prob.l, prob.h, prob.d
for i in range(4):
    pr.goal[i] = float(prob.goal[i])
pr.cost_diag = 1
o = _lib.ILQGOpts(); L.ddp_ilqg_default_opts(C.byref(o))
o.regType, o.lambda_max, o.tol_fun, o.tol_grad, o.max_iter, o.n_alpha = 2, 1e15, 1e-8, 1e-8, 1000, len(kw["α"])
for i, a in enumerate(kw["α"]):
    o.alpha[i] = a
e = lambda c: torch.empty(int(c), dtype=torch.float64, device=dev)
CL = T + 1
x, u, K, k, Quu, Vx, Vxx, cost, stats = e(n*T*B), e(m*T*B), e(m*n*T*B), e(m*T*B), e(m*m*T*B), e(n*T*B), e(n*n*T*B), e(CL*B), e(8*B)
git = C.c_int(0)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    _lib.check(L.ddp_ilqg_f64_dev(h.raw, C.byref(pr), C.byref(o), p_(dx0), p_(du0), p_(dl), p_(x), p_(u), p_(K), p_(k), p_(Quu), p_(Vx), p_(Vxx), p_(cost), p_(stats), 0, None, C.byref(git)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("C3 iLQG pendcart B=%d device-resident: %.4f s, %d batch iterations, mean cost %.1f" % (B, dt, git.value, float(stats.view(B, 8)[:, 7].mean())))
