"""C2 end-to-end: 1024 demo_linear iLQG solves (n=10, m=2, N=1000, default options) through the device-resident driver,
host-pointer entry (H2D/D2H of every result included) and device-pointer entry (what a resident pipeline pays)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ddp_amd
from ddp_amd import _lib
from oracle import np_restatement as npr
B, T = int(os.environ.get("C2_B", 1024)), 1000
rng = np.random.default_rng(1234)
P = npr.make_lq_problem(rng)
prob = ddp_amd.LQProblem(P["A"], P["B"], P["Q"], P["R"])
x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B)); u0 = 0.1 * rng.standard_normal((2, T, B))
x0, u0 = np.asfortranarray(x0), np.asfortranarray(u0)      # the layout a Julia caller has (a C-ordered u0 costs the mirror an 8 ms transposing copy)
for it in range(3):
    t = time.perf_counter()
    r = ddp_amd.iLQG(prob, x0, u0)
    dt = time.perf_counter() - t
    tr = r[6]; st = tr["stats"]
    parts = {k: float(np.nansum(tr[k])) for k in ("time_derivs", "time_backward", "time_forward")}
    print("C2 iLQG LQ B=%d host-pointer: %.3f s (%.3f s inside the C call), %d batch iterations, exit %s, mean cost %.3f | GPU phases derivs %.4f back %.4f fwd(11 alphas) %.4f"
          % (B, dt, tr["time_total"], tr["global_iters"], dict(zip(*np.unique(st[0].astype(int), return_counts=True))), st[7].mean(), parts["time_derivs"], parts["time_backward"], parts["time_forward"]))
# device-resident
dev = torch.device("cuda", 0); L = _lib.lib(); h = ddp_amd.default_handle()
f64 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))).to(dev)
p_ = lambda t: C.c_void_p(t.data_ptr())
n, m = 10, 2
dA, dB, dQ, dR, dx0, du0 = map(f64, (P["A"], P["B"], P["Q"], P["R"], x0, u0))
pr = _lib.Problem(); pr.kind, pr.n, pr.m, pr.N, pr.B = 0, n, m, T, B
pr.A, pr.Bm, pr.Q, pr.R = dA.data_ptr(), dB.data_ptr(), dQ.data_ptr(), dR.data_ptr()
pr.cost_diag = 1
e = lambda c: torch.empty(int(c), dtype=torch.float64, device=dev)
x, u, K, k, Quu, Vx, Vxx, cost, stats = e(n*T*B), e(m*T*B), e(m*n*T*B), e(m*T*B), e(m*m*T*B), e(n*T*B), e(n*n*T*B), e(T*B), e(8*B)
git = C.c_int(0)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    _lib.check(L.ddp_ilqg_f64_dev(h.raw, C.byref(pr), None, p_(dx0), p_(du0), None, p_(x), p_(u), p_(K), p_(k), p_(Quu), p_(Vx), p_(Vxx), p_(cost), p_(stats), 0, None, C.byref(git)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("C2 iLQG LQ B=%d device-resident: %.4f s, %d batch iterations" % (B, dt, git.value))
