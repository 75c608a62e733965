"""Fixtures added in round 4 so that ONE run of julia/make_reference_fixtures.jl pins every §8 row:

    ilqg_warm_lq     pre-rolled warm start  iLQG(f,costfun,df,x0[n,N],u0; cost)                     src/iLQG.jl:193-197
    ilqg_trace_lq    all per-iteration trace keys of a solve (λ, dλ, α, improvement, cost, reduce_ratio, grad_norm)   src/iLQG.jl:257,325-330
    (kl_gps_* gain the calc_η keys, kl_ilqgkl_* become exported families: tests/golden/rawio.py)

Outputs come from the C restatement (oracle/ddp_oracle.c); the NumPy restatement is the second opinion where it has the entry.
    python tests/golden/make_golden_r4.py && python tests/golden/rawio.py
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_ctypes as oc  # noqa: E402


def main():
    rng = np.random.default_rng(20260930)
    n, m, T, h = 10, 2, 80, 0.01
    A0 = rng.standard_normal((n, n)); A = sla.expm(h * (A0 - A0.T)); B = h * rng.standard_normal((n, m))
    Q, R = h * np.eye(n), 0.1 * h * np.eye(m)
    p = oc.make_problem("lq", n, m, T, A=A, B=B, Q=Q, R=R)
    x0 = np.ones(n) + 0.1 * rng.standard_normal(n)
    u0 = 0.1 * rng.standard_normal((m, T))
    # ---- all trace keys of a cold solve
    x, u, (K, k, Quu), Vx, Vxx, cost, info = oc.ilqg_trace7(p, x0, u0)
    hist = info["history"]
    np.savez(os.path.join(HERE, "ilqg_trace_lq.npz"), A=A, B=B, Q=Q, R=R, x0=x0, u0=u0, x=x, u=u, K=K, k=k, Quu=Quu, Vx=Vx, Vxx=Vxx, cost=cost,
             status=info["status"], iter=info["iter"], tr_lambda=hist["λ"], tr_dlambda=hist["dλ"], tr_alpha=hist["α"], tr_improvement=hist["improvement"],
             tr_cost=hist["cost"], tr_reduce_ratio=hist["reduce_ratio"], tr_grad_norm=hist["grad_norm"])
    # ---- warm start: the solution of a SHORT solve, shifted by one step (what an MPC loop hands over), pre-rolled with its cost
    xs, us, _, _, _, cs, _ = oc.ilqg(p, x0, u0, max_iter=3)
    uw = np.concatenate([us[:, 1:], us[:, -1:]], axis=1)
    xw = np.zeros((n, T)); xw[:, 0] = xs[:, 1]
    for t in range(T - 1):
        xw[:, t + 1] = A @ xw[:, t] + B @ uw[:, t]
    cw = 0.5 * np.sum(xw * (Q @ xw), axis=0) + 0.5 * np.sum(uw * (R @ uw), axis=0)
    x, u, (K, k, Quu), Vx, Vxx, cost, info = oc.ilqg_prerolled(p, xw, uw, cost0=cw)
    np.savez(os.path.join(HERE, "ilqg_warm_lq.npz"), A=A, B=B, Q=Q, R=R, x0=xw, u0=uw, cost0=cw, x=x, u=u, K=K, k=k, Quu=Quu, Vx=Vx, Vxx=Vxx,
             cost=cost, status=info["status"], iter=info["iter"], lam=info["lam"], n_backpass=info["n_backpass"], n_forward=info["n_forward"])
    print("wrote ilqg_trace_lq.npz (iter %d), ilqg_warm_lq.npz (iter %d)" % (np.load(os.path.join(HERE, "ilqg_trace_lq.npz"))["iter"], info["iter"]))


if __name__ == "__main__":
    main()
