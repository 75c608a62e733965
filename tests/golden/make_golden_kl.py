"""Generates the KL-path fixtures tests/golden/kl_*.npz from oracle/np_kl.py (NumPy restatement of back_pass_gps, ∇kl,
forward_covariance, kl_div_wiki and the single-constraint iLQGkl loop; the reference itself is Julia and cannot run here).

    python tests/golden/make_golden_kl.py

DATA only (inputs + expected outputs); "parity unpinned" with respect to the Julia code — see oracle/ddp_oracle_kl.c.
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_kl, np_restatement as npr  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(name, sum(np.asarray(a).nbytes for a in arrs.values()), "bytes raw")


def spd(rng, d, s=1.0):
    a = rng.standard_normal((d, d))
    return s * (a @ a.T / d + 0.5 * np.eye(d))


def gps_problem(rng, n, m, N):
    fx = np.stack([np.eye(n) + 0.1 * rng.standard_normal((n, n)) for _ in range(N)], -1)
    fu = 0.3 * rng.standard_normal((n, m, N))
    cxx = np.stack([spd(rng, n) for _ in range(N)], -1)
    cuu = np.stack([spd(rng, m, 0.5) for _ in range(N)], -1)
    cxu = 0.05 * rng.standard_normal((n, m, N))
    cx, cu = rng.standard_normal((n, N)), rng.standard_normal((m, N))
    u, x = 0.3 * rng.standard_normal((m, N)), rng.standard_normal((n, N))
    Kp, kp = 0.2 * rng.standard_normal((m, n, N)), 0.1 * rng.standard_normal((m, N))
    Sip = np.stack([spd(rng, m, 2.0) for _ in range(N)], -1)
    Sp = np.stack([np.linalg.inv(Sip[:, :, t]) for t in range(N)], -1)
    return dict(fx=fx, fu=fu, cxx=cxx, cuu=cuu, cxu=cxu, cx=cx, cu=cu, u=u, x=x, Kp=Kp, kp=kp, Sip=Sip, Sp=Sp)


def gps_case(name, P, etab, lims):
    kl = np_kl.grad_kl(P["Kp"], P["kp"], P["Sip"])
    d, (K, k, Quui, Quu), Vx, Vxx, dV = np_kl.back_pass_gps(P["cx"], P["cu"], P["cxx"], P["cxu"], P["cuu"], P["fx"], P["fu"], lims,
                                                          P["x"], P["u"], (kl, etab))
    n = P["fx"].shape[0]
    R1 = 0.01 * np.eye(n)
    sig = np_kl.forward_covariance(P["fx"], R1, K, Quui)
    xnew = P["x"] + 0.1 * np.cos(np.arange(P["x"].size).reshape(P["x"].shape))
    kld = np_kl.kl_div_wiki(xnew, P["x"], sig, dict(K=K, k=k, S=Quui, Si=Quu), dict(K=P["Kp"], k=P["kp"], S=P["Sp"], Si=P["Sip"]))
    save(name, etab=np.asarray(etab, float), lims=np.zeros((0, 0)) if lims is None else lims, cxkl=kl[0], cukl=kl[1], cxxkl=kl[2],
         cxukl=kl[3], cuukl=kl[4], diverge=d, K=K, k=k, Quui=Quui, Quu=Quu, Vx=Vx, Vxx=Vxx, dV=dV, R1=R1, sigmanew=sig, xnew=xnew,
         kldiv=kld, **P)


def main():
    rng = np.random.default_rng(20260929)
    P = gps_problem(rng, 4, 2, 40)
    gps_case("kl_gps_n4m2", P, [1e-8, 1.0, 1e16], None)
    gps_case("kl_gps_n4m2_lims", P, [1e-8, 3.0, 1e16], np.array([[-0.2, 0.3], [-0.25, 0.2]]))
    N = 40
    gps_case("kl_gps_n4m2_eta_per_step", P, np.stack([1e-8 * np.ones(N), np.linspace(0.5, 4, N), 1e16 * np.ones(N)]), None)
    P1 = gps_problem(rng, 4, 1, 50)                                      # C5 shape (pendcart sizes)
    gps_case("kl_gps_n4m1_lims", P1, [1e-8, 2.0, 1e16], np.array([[-0.4, 0.5]]))
    P2 = gps_problem(rng, 10, 2, 30)
    gps_case("kl_gps_n10m2", P2, [1e-8, 0.7, 1e16], None)
    P3 = dict(P)                                                          # non-PD Quu at one step -> diverge
    P3["cuu"] = P["cuu"].copy(); P3["cuu"][:, :, 17] = -40 * np.eye(2)
    gps_case("kl_gps_n4m2_diverge", P3, [1e-8, 1.0, 1e16], None)

    # ---- single-constraint iLQGkl on a small LQ problem (demo_linear_kl shapes, short horizon)
    n, m, T, h = 6, 2, 60, 0.01
    A0 = rng.standard_normal((n, n)); A = sla.expm(h * (A0 - A0.T)); B = h * rng.standard_normal((n, m))
    Q, R = h * np.eye(n), 0.1 * h * np.eye(m)
    x0, u = np.ones(n), 0.1 * rng.standard_normal((m, T))
    f, costfun, _ = npr.lq_closures(A, B, Q, R)
    x = np.zeros((n, T)); x[:, 0] = x0
    for t in range(T - 1):
        x[:, t + 1] = A @ x[:, t] + B @ u[:, t]
    fx, fu = np.repeat(A[:, :, None], T, 2), np.repeat(B[:, :, None], T, 2)

    def derivs(x, u):
        return fx, fu, Q @ x, R @ u, np.repeat(Q[:, :, None], T, 2), np.zeros((n, m, T)), np.repeat(R[:, :, None], T, 2)
    eye = np.repeat(np.eye(m)[:, :, None], T, 2)
    prev = dict(K=np.zeros((m, n, T)), k=u.copy(), S=eye.copy(), Si=eye.copy())
    model = dict(fx=fx, R1=1e-4 * np.eye(n))
    cost0 = 0.5 * np.sum(x * (Q @ x)) + 0.5 * np.sum(u * (R @ u))
    for tag, kl_step in (("a", 1e-3), ("b", 1e-4)):
        xo, uo, pol, Vx, Vxx, cost, info = np_kl.iLQGkl(f, costfun, derivs, x, prev, model, kl_step=kl_step, max_iter=50)
        save("kl_ilqgkl_lq_" + tag, A=A, B=B, Q=Q, R=R, x=x, u=u, R1=model["R1"], kl_step=kl_step, cost0=cost0, xnew=xo, unew=uo,
             K=pol["K"], S=pol["S"], Si=pol["Si"], Vx=Vx, Vxx=Vxx, cost=cost, status=info["status"], iter=info["iter"], eta=info["eta"],
             divergence=info["divergence"], n_backpass=info["n_backpass"])


if __name__ == "__main__":
    main()
