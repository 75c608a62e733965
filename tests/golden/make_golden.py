"""Generates tests/golden/*.npz from oracle/np_restatement.py (NumPy restatement of the
reference semantics; the reference itself is Julia and cannot run in this image).

    python tests/golden/make_golden.py

The fixtures are DATA (inputs + expected outputs).  They pin the C oracle and the HIP path to
the NumPy restatement; they are not outputs of the Julia code ("parity unpinned" — see
oracle/ddp_oracle.h).
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_restatement as npr  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(name, sum(np.asarray(a).nbytes for a in arrs.values()), "bytes raw")


def bp_case(name, cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, lims, x, u):
    d, (K, k, Quu), Vx, Vxx, dV = npr.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, lims, x, u)
    save(name, cx=cx, cu=cu, cxx=cxx, cxu=cxu, cuu=cuu, fx=fx, fu=fu, lam=lam, regType=regType,
         lims=np.zeros((0, 0)) if lims is None else lims, x=x, u=u,
         diverge=d, K=K, k=k, Quu=Quu, Vx=Vx, Vxx=Vxx, dV=dV)


def main():
    rng = np.random.default_rng(20260928)
    # ---- a1: LTI / time-invariant cost (demo_linear shapes, short horizon)
    P = npr.make_lq_problem(rng, n=10, m=2, T=60)
    f, costfun, df = npr.lq_closures(P['A'], P['B'], P['Q'], P['R'])
    x, u, c = npr.forward_pass(None, P['x0'], P['u0'], None, 1, f, costfun, None)
    fx, fu, cx, cu, cxx, cxu, cuu = df(x, u)
    bp_case("bp_lti_n10m2_reg1", cx, cu, cxx, cxu, cuu, fx, fu, 1.0, 1, None, x, u)
    bp_case("bp_lti_n10m2_reg2", cx, cu, cxx, cxu, cuu, fx, fu, 0.37, 2, None, x, u)
    lims = np.array([[-0.05, 0.08], [-0.1, 0.02]])
    bp_case("bp_lti_n10m2_lims", cx, cu, cxx, cxu, cuu, fx, fu, 1e-3, 1, lims, x, u)
    # forward pass fixtures on the same problem
    d, (K, k, Quu), Vx, Vxx, dV = npr.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 1e-3, 1, None, x, u)
    alphas = 10.0 ** np.linspace(0, -3, 11)
    xs, us, cs = zip(*[npr.forward_pass((K, k), P['x0'], u, x, a, f, costfun, None) for a in alphas])
    save("fwd_lq_n10m2", A=P['A'], B=P['B'], Q=P['Q'], R=P['R'], x0=P['x0'], u=u, x=x, K=K, k=k,
         alphas=alphas, xnew=np.stack(xs, -1), unew=np.stack(us, -1), cnew=np.stack(cs, -1), dV=dV,
         cost0=c)
    xs, us, cs = zip(*[npr.forward_pass((K, k), P['x0'], u, x, a, f, costfun, lims) for a in alphas[:3]])
    save("fwd_lq_n10m2_lims", A=P['A'], B=P['B'], Q=P['Q'], R=P['R'], x0=P['x0'], u=u, x=x, K=K, k=k,
         alphas=alphas[:3], lims=lims, xnew=np.stack(xs, -1), unew=np.stack(us, -1), cnew=np.stack(cs, -1))

    # ---- a2: LTV dynamics / time-invariant cost with limits, m = 1 (pendcart linearisation)
    PC = dict(npr.PENDCART); T = 80
    fp, cp, dfp = npr.pendcart_closures(PC)
    u0 = 2.0 * np.sin(np.arange(T) / 7.0)[None, :]
    x, u, c = npr.forward_pass(None, PC['x0'], u0, None, 1, fp, cp, PC['lims'])
    fx, fu, cx, cu, cxx, cxu, cuu = dfp(x, u)
    bp_case("bp_ltv_pendcart_lims", cx, cu, cxx, cxu, cuu, fx, fu, 1.0, 2, PC['lims'], x, u)
    bp_case("bp_ltv_pendcart_nolims", cx, cu, cxx, cxu, cuu, fx, fu, 1.0, 2, None, x, u)
    save("df_pendcart", x=x, u=u, fx=fx, fu=fu, cx=cx, cu=cu)
    d, (K, k, Quu), Vx, Vxx, dV = npr.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 1.0, 2, PC['lims'], x, u)
    al = 10.0 ** np.linspace(0.2, -3, 6)
    xs, us, cs = zip(*[npr.forward_pass((K, k), PC['x0'], u, x, a, fp, cp, PC['lims']) for a in al])
    save("fwd_pendcart", x0=PC['x0'], u=u, x=x, K=K, k=k, alphas=al, lims=PC['lims'],
         xnew=np.stack(xs, -1), unew=np.stack(us, -1), cnew=np.stack(cs, -1))

    # ---- a3: LTV dynamics / time-varying cost, m = 3, with and without limits
    n, m, N = 6, 3, 40
    fx = np.stack([sla.expm(0.05 * (lambda a: a - a.T)(rng.standard_normal((n, n)))) for _ in range(N)], -1)
    fu = 0.1 * rng.standard_normal((n, m, N))
    def spd(d, s):
        a = rng.standard_normal((d, d)); return s * (a @ a.T / d + 0.5 * np.eye(d))
    cxx = np.stack([spd(n, 0.1) for _ in range(N)], -1)
    cuu = np.stack([spd(m, 0.05) for _ in range(N)], -1)
    cxu = 0.01 * rng.standard_normal((n, m, N))
    x = rng.standard_normal((n, N)); u = 0.3 * rng.standard_normal((m, N))
    cx = rng.standard_normal((n, N)) * 0.1; cu = rng.standard_normal((m, N)) * 0.1
    bp_case("bp_tv_n6m3_reg1", cx, cu, cxx, cxu, cuu, fx, fu, 0.5, 1, None, x, u)
    bp_case("bp_tv_n6m3_reg2", cx, cu, cxx, cxu, cuu, fx, fu, 0.5, 2, None, x, u)
    lims3 = np.array([[-0.5, 0.5], [-0.3, 0.4], [-1.0, 0.2]])
    bp_case("bp_tv_n6m3_lims", cx, cu, cxx, cxu, cuu, fx, fu, 0.5, 1, lims3, x, u)
    # divergence: make cuu indefinite at one step so the Cholesky fails there (diverge == 18)
    cuu_bad = cuu.copy(); cuu_bad[:, :, 17] = -np.eye(m)
    bp_case("bp_tv_n6m3_diverge", cx, cu, cxx, cxu, cuu_bad, fx, fu, 1e-3, 1, None, x, u)
    bp_case("bp_tv_n6m3_diverge_lims", cx, cu, cxx, cxu, cuu_bad, fx, fu, 1e-3, 1, lims3, x, u)

    # ---- a6: boxQP
    Hs, gs, los, ups, x0s, xs_, res, frees, Hfs = [], [], [], [], [], [], [], [], []
    mq = 8
    for t in range(48):
        m_ = [1, 2, 3, 5, 8][t % 5]
        a = rng.standard_normal((m_, m_)); H = a @ a.T + 0.1 * np.eye(m_)
        g = rng.standard_normal(m_) * (3.0 if t % 3 else 0.3)
        lo = -np.abs(rng.standard_normal(m_)) * 0.5; up = np.abs(rng.standard_normal(m_)) * 0.5
        if t % 7 == 0:
            lo[:] = -np.inf
        x0 = rng.standard_normal(m_)
        xq, r, Hf, fr = npr.boxQP(H, g, lo, up, x0)
        pad = lambda v, fill=0.0: np.pad(np.asarray(v, float), (0, mq - m_), constant_values=fill)
        Hp = np.zeros((mq, mq)); Hp[:m_, :m_] = H
        Hfp = np.zeros((mq, mq)); Hfp[:Hf.shape[0], :Hf.shape[1]] = Hf
        Hs.append(Hp); gs.append(pad(g)); los.append(pad(lo)); ups.append(pad(up)); x0s.append(pad(x0))
        xs_.append(pad(xq)); res.append(r); frees.append(pad(fr)); Hfs.append(Hfp)
    save("boxqp", m=np.array([[1, 2, 3, 5, 8][t % 5] for t in range(48)]), H=np.stack(Hs), g=np.stack(gs),
         lower=np.stack(los), upper=np.stack(ups), x0=np.stack(x0s), x=np.stack(xs_), result=np.array(res),
         free=np.stack(frees), Hfree=np.stack(Hfs))

    # ---- a8: full iLQG solves (small horizons)
    P = npr.make_lq_problem(rng, n=10, m=2, T=120)
    f, costfun, df = npr.lq_closures(P['A'], P['B'], P['Q'], P['R'])
    x, u, (K, k, Quu), Vx, Vxx, cost, info = npr.iLQG(f, costfun, df, P['x0'], P['u0'])
    save("ilqg_lq_n10m2", A=P['A'], B=P['B'], Q=P['Q'], R=P['R'], x0=P['x0'], u0=P['u0'], x=x, u=u, K=K, k=k,
         Quu=Quu, Vx=Vx, Vxx=Vxx, cost=cost, status=info['status'], iter=info['iter'], lam=info['lam'],
         n_backpass=info['n_backpass'], n_forward=info['n_forward'], tr_cost=np.array(info['trace']['cost']))
    PC = dict(npr.PENDCART); T = 150
    fp, cp, dfp = npr.pendcart_closures(PC)
    kw = dict(regType=2, alpha=10.0 ** np.linspace(0.2, -3, 6), lam_max=1e15, tol_fun=1e-8, tol_grad=1e-8, max_iter=1000)
    x, u, (K, k, Quu), Vx, Vxx, cost, info = npr.iLQG(fp, cp, dfp, PC['x0'], np.zeros((1, T)), lims=PC['lims'], **kw)
    save("ilqg_pendcart", x0=PC['x0'], T=T, x=x, u=u, K=K, k=k, Quu=Quu, Vx=Vx, Vxx=Vxx, cost=cost,
         status=info['status'], iter=info['iter'], lam=info['lam'], n_backpass=info['n_backpass'],
         n_forward=info['n_forward'], tr_cost=np.array(info['trace']['cost']))


if __name__ == "__main__":
    main()
