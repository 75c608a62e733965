"""C5 (= C3 + KL constraint): a batch of KL-constrained pendcart solves through the host mirror (kl.iLQGkl): the iteration runs on
device-resident arrays with the dual variable η updated on the device (ddp_kl_dual_*); the time includes the upload of the
derivative arrays and the download of the results."""
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
