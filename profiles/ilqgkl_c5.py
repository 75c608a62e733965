"""C5 (= C3 + KL constraint): a batch of KL-constrained pendcart solves.  First through the host mirror (kl.iLQGkl -> ddp_ilqgkl_f64:
the loop is one library call; the time includes the upload of the inputs and the download of the results into pageable host arrays),
then through ddp_ilqgkl_f64_dev with every array resident on the device (what the kernels and the loop cost)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddp_amd
import ddp_amd.kl as kl
B, N = int(os.environ.get("C5_B", 4096)), 600
rng = np.random.default_rng(0)
prob = ddp_amd.PendcartProblem()
lims = 5.0 * np.array([[-1.0, 1.0]])
u = 2.0 * np.sin(np.arange(N) / 37.0)[None, :, None] * np.ones((1, 1, B)) + 0.05 * rng.standard_normal((1, N, B))
x0 = np.tile(np.array([np.pi - 0.6, 0, 0, 0])[:, None], (1, B)); x0[0] += rng.uniform(-0.1, 0.1, B)
x, _, c0 = ddp_amd.forward_pass(None, x0, u, None, 1.0, prob, lims)
cost0 = c0.sum(axis=0)
fx, fu = ddp_amd.df(prob, x, u)[:2]
R1 = 1e-3 * np.eye(4)
eye = np.ones((1, 1, N, B))
prev = ddp_amd.GaussianPolicy(N, 4, 1, np.zeros((1, 4, N, B)), u, eye, eye.copy())
for it in range(2):
    t = time.perf_counter()
    xo, uo, pol, Vx, Vxx, cost, tr = kl.iLQGkl(prob, x, prev, kl.Model(fx, fu, R1), kl_step=0.05, lims=lims, max_iter=30, cost=cost0)
    dt = time.perf_counter() - t
    print("C5 iLQGkl pendcart B=%d N=%d: %.3f s, status counts %s, mean iterations %.1f, mean back passes %.1f, mean cost %.1f -> %.1f"
          % (B, N, dt, dict(zip(*np.unique(tr["status"], return_counts=True))), tr["iter"].mean(), tr["n_backpass"].mean(), cost0.mean(), cost.sum(axis=0).mean()))

# ---- device-resident: ddp_ilqgkl_f64_dev
import ctypes as C
from ddp_amd import _lib
L, h = _lib.lib(), ddp_amd.default_handle()
up = lambda a: h.to_device(_lib.f64(a))                                    # noqa: E731
dev = lambda *sh: h.malloc(int(np.prod(sh)) * 8)                           # noqa: E731
P = _lib.Problem()
P.kind, P.n, P.m, P.N, P.B = 1, 4, 1, N, B
P.Q, P.R, P.cost_diag = up(prob.Q), up(prob.R), 1
P.g, P.l, P.h, P.d = prob.g, prob.l, prob.h, prob.d
for i in range(4):
    P.goal[i] = float(prob.goal[i])
o = _lib.ILQGKLOpts()
L.ddp_ilqgkl_default_opts(C.byref(o))
o.kl_step, o.max_iter = 0.05, 30
ins = [up(x), up(cost0), up(np.zeros((1, 4, N, B))), up(u), up(eye), up(eye), up(fx)]
dR1, dl = up(R1), up(np.asfortranarray(lims))
outs = [dev(4, N, B), dev(1, N, B), dev(1, 4, N, B), dev(1, 1, N, B), dev(1, 1, N, B), dev(4, N, B), dev(4, 4, N, B), dev(N + 1, B), dev(2, B), dev(12, B)]
its = C.c_int(0)
for rep in range(4):
    h.sync()
    t = time.perf_counter()
    _lib.check(L.ddp_ilqgkl_f64_dev(h.raw, C.byref(P), C.byref(o), *ins[:7], 1, dR1, dl, None, *outs, C.byref(its)))
    h.sync()
    dt = time.perf_counter() - t
    st = h.to_host(outs[9], (12, B))
    print("C5 ddp_ilqgkl_f64_dev B=%d N=%d: %.4f s device-resident, %d batch iterations, mean iterations %.2f, mean back passes %.2f, same as host call: %s"
          % (B, N, dt, its.value, st[1].mean(), st[2].mean(), bool(np.array_equal(st[0].astype(int), tr["status"]) and np.allclose(st[5], tr["η"][1], rtol=1e-12))))
