"""Randomised GPU-vs-oracle parity sweep (not collected by pytest; run on a GPU box: python tests/fuzz_gpu_parity.py [cases] [seed]).
Draws shapes, operand layouts (LTI / LTV, shared / per-trajectory, time-varying cost), regType, λ, limits and horizon, runs
back_pass and forward_pass through the C ABI and compares every output with the C restatement at 1e-8."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import relerr  # noqa: E402

RTOL = 1e-8


def spd(rng, d, s):
    a = rng.standard_normal((d, d))
    return s * (a @ a.T / d + 0.5 * np.eye(d))


def one_case(ddp, oc, rng, case):
    n, m = [(10, 2), (4, 1), (6, 3), (64, 8), (7, 2), (12, 4), (40, 4)][rng.integers(0, 7)]
    big = rng.integers(0, 6) == 0                                 # now and then a long horizon / several waves
    N = (int(rng.integers(1, 300 if big else 40))) if n < 40 else int(rng.integers(2, 14))
    B = int(rng.integers(1, 70 if (big and n < 40) else 6))
    fx_tv, fx_b = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    c_tv = bool(rng.integers(0, 2))
    regType = int(rng.integers(1, 3))
    lims = None
    if rng.integers(0, 3) == 0:
        lims = np.stack([-rng.uniform(0.05, 1.0, m), rng.uniform(0.05, 1.0, m)], 1)
    h = 0.05
    shp = ((N,) if fx_tv else ()) + ((B,) if fx_b else ())
    if fx_b and not fx_tv:
        fx_tv = True; shp = (N, B)              # per-trajectory operands use the time-varying layout (a3)
    fx = np.eye(n).reshape((n, n) + (1,) * len(shp)) + h * rng.standard_normal((n, n) + shp) / np.sqrt(n)
    fu = h * rng.standard_normal((n, m) + shp)
    cshp = (N, B) if c_tv else ()
    if c_tv:
        cxx = np.stack([np.stack([spd(rng, n, h) for _ in range(N)], -1) for _ in range(B)], -1)
        cuu = np.stack([np.stack([spd(rng, m, 0.1 * h) for _ in range(N)], -1) for _ in range(B)], -1)
        cxu = 0.01 * h * rng.standard_normal((n, m, N, B))
    else:
        cxx, cuu, cxu = spd(rng, n, h), spd(rng, m, 0.1 * h), 0.01 * h * rng.standard_normal((n, m))
    cx = h * rng.standard_normal((n, N, B)); cu = 0.1 * h * rng.standard_normal((m, N, B))
    u = 0.3 * rng.standard_normal((m, N, B))
    lam = 10.0 ** rng.uniform(-4, 1, B)
    if rng.integers(0, 8) == 0 and N > 3 and c_tv:                 # provoke a divergence somewhere
        cuu[:, :, int(rng.integers(0, N - 1)), int(rng.integers(0, B))] = -np.eye(m)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, lims, None, u)
    worst = 0.0
    for b in range(B):
        sl = lambda a_, nd: a_[..., b] if a_.ndim == nd + 1 else a_          # noqa: E731
        fxb = fx[..., b] if fx_b else fx
        fub = fu[..., b] if fx_b else fu
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], sl(cxx, 3), sl(cxu, 3), sl(cuu, 3), fxb, fub,
                                                  lam[b], regType, lims, None, u[..., b])
        assert div[b] == d, ("diverge", case, n, m, N, B, div[b], d)
        for got, ref, name in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (Vx[..., b], vx, "Vx"), (Vxx[..., b], vxx, "Vxx"),
                               (dV[:, b], dv, "dV")):
            e = relerr(got, ref)
            worst = max(worst, e)
            assert e < RTOL, (name, e, case, dict(n=n, m=m, N=N, B=B, fx_tv=fx_tv, fx_b=fx_b, c_tv=c_tv, regType=regType, lims=lims is not None))
    # forward rollout with the gains just computed (LQ family), two step sizes
    if not fx_b or fx_tv:
        Q, R = spd(rng, n, h), spd(rng, m, 0.1 * h)
        prob = ddp.LQProblem(fx, fu, Q, R, dyn_batched=fx_b) if fx_tv else ddp.LQProblem(fx, fu, Q, R)
        x0 = rng.standard_normal((n, B)); xnom = rng.standard_normal((n, N, B))
        Kc = np.where(np.isfinite(pol.K), pol.K, 0.0); kc = np.where(np.isfinite(pol.k), pol.k, 0.0)
        al = np.array([1.0, 0.3])
        xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, n, m, Kc, kc), x0, u, xnom, al, prob, lims)
        for b in range(B):
            p = oc.make_problem("lq", n, m, N, A=fx[..., b] if fx_b else fx, B=fu[..., b] if fx_b else fu, Q=Q, R=R)
            for j, a in enumerate(al):
                xr, ur, cr = oc.forward_pass(p, (Kc[..., b], kc[..., b]), x0[:, b], u[..., b], xnom[..., b], float(a), lims)
                for got, ref, name in ((xn[..., b, j], xr, "xnew"), (un[..., b, j], ur, "unew"), (cn[..., b, j], cr, "cnew")):
                    e = relerr(got, ref)
                    worst = max(worst, e)
                    assert e < RTOL, (name, e, case, dict(n=n, m=m, N=N, B=B, fx_tv=fx_tv, fx_b=fx_b, lims=lims is not None))
    return worst


def main():
    import ddp_amd as ddp
    from oracle import oracle_ctypes as oc
    ddp.default_handle()
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst = 0.0
    for c in range(cases):
        worst = max(worst, one_case(ddp, oc, rng, c))
    print("fuzz: %d cases passed, worst relative error %.3g" % (cases, worst))


if __name__ == "__main__":
    main()
