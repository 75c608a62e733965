"""Randomised GPU-vs-oracle parity sweep (not collected by pytest; run on a GPU box: python tests/fuzz_gpu_parity.py [cases] [seed];
`--cond seed case` runs on the CPU and prints how far the two CPU restatements are apart on that case).
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


# every (n, m) the padded row kernels (csrc/back_pass_row.hip, forward_pass_row.hip) hold: n <= 14, m <= 4, n + m <= 15 and their table of
# padded sizes — the sweep of round 5 draws from these with the row kernel forced, dispatched by default, or the run-time-sized kernel
ROW_SHAPES = [(n_, m_) for n_ in range(1, 15) for m_ in range(1, 5) if n_ + m_ <= 15 and not (n_ > 12 and m_ > 1) and not (n_ > 10 and m_ > 3)]
ROW_IMPLS = [None, None, "row", "row", "general", "tile", "wtile", "mid"]       # a forced kernel that does not hold the shape falls through


def gen_case(rng, shapes=None, impls=None):
    default_shapes = [(10, 2), (4, 1), (6, 3), (64, 8), (7, 2), (12, 4), (40, 4), (37, 3), (51, 2)]
    shapes = shapes or default_shapes
    n, m = shapes[rng.integers(0, len(shapes))]
    big = rng.integers(0, 6) == 0                                 # now and then a long horizon / several waves
    N = (int(rng.integers(1, 300 if big else 40))) if n < 40 else int(rng.integers(2, 50 if big else 14))
    if rng.integers(0, 3) == 0:
        N = max(8, N // 8 * 8)                                    # the kernels that move groups of 8 steps through the LDS (q4l, mx)
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
    Q, R = spd(rng, n, h), spd(rng, m, 0.1 * h)
    x0 = rng.standard_normal((n, B)); xnom = rng.standard_normal((n, N, B))
    impls = impls or [None, None, None, "x", "dpp", "dpp", "general", "big", "q"]
    impl = impls[rng.integers(0, len(impls))]                      # forced kernel (falls back when it has no such shape)
    return dict(impl=impl, n=n, m=m, N=N, B=B, fx_tv=fx_tv, fx_b=fx_b, c_tv=c_tv, regType=regType, lims=lims, fx=fx, fu=fu, cxx=cxx, cuu=cuu, cxu=cxu,
                cx=cx, cu=cu, u=u, lam=lam, Q=Q, R=R, x0=x0, xnom=xnom)


def ref_back_pass(bp, c, b):
    sl = lambda a_, nd: a_[..., b] if a_.ndim == nd + 1 else a_          # noqa: E731
    fxb = c["fx"][..., b] if c["fx_b"] else c["fx"]
    fub = c["fu"][..., b] if c["fx_b"] else c["fu"]
    return bp(c["cx"][..., b], c["cu"][..., b], sl(c["cxx"], 3), sl(c["cxu"], 3), sl(c["cuu"], 3), fxb, fub, c["lam"][b], c["regType"],
              c["lims"], None, c["u"][..., b])


COND_CAP = 1e-6                 # restatements further apart than this cannot judge a draw (it is counted, cross-checked on the device, bounded)
# committed ceilings of the escapes per 1 000 cases of their kind (the slice of tests/test_gpu_fuzz_slice.py, seed 31, stays well below)
MAX_ILL_PER_1000, MAX_UNJUDGED_PER_1000, MAX_KNIFE_PER_1000 = 8.0, 1.0, 30.0


def escape_counts():
    return dict(ill_conditioned=getattr(one_case, "ill_conditioned", 0), unjudged=getattr(one_case, "unjudged", 0),
                knife_edge=getattr(ilqg_case, "knife_edge", 0), exploded_kl_draws=getattr(gps_case, "skipped", 0))


def one_case(ddp, oc, rng, case, shapes=None, impls=None):
    c = gen_case(rng, shapes, impls)
    n, m, N, B, lims = c["n"], c["m"], c["N"], c["B"], c["lims"]
    tag = (case, {k: c[k] for k in ("n", "m", "N", "B", "fx_tv", "fx_b", "c_tv", "regType", "impl")}, "lims" if lims is not None else "no lims")
    os.environ.pop("DDP_BACKPASS", None)
    if c["impl"]:
        os.environ["DDP_BACKPASS"] = c["impl"]
    # the work-group pipelines of round 3 (back_pass_mx2: chain + writer wave; forward_pass_pipe) forced on / off / as dispatched
    # back_pass_dppw (row kernel + write-back waves, default only from B = 6 144) forced on / off; the rollout's FAST variant (mirrored idle lanes) off / as dispatched
    for var, pick in (("DDP_MX2", case % 3), ("DDP_FORWARD_PIPE", (case // 3) % 3), ("DDP_DPPW", (case // 2) % 3), ("DDP_FORWARD_FAST", 2 * ((case // 5) % 2))):
        os.environ.pop(var, None)
        if pick < 2:
            os.environ[var] = str(pick)
    div, pol, Vx, Vxx, dV = ddp.back_pass(c["cx"], c["cu"], c["cxx"], c["cxu"], c["cuu"], c["fx"], c["fu"], c["lam"], c["regType"], lims, None, c["u"])
    os.environ.pop("DDP_BACKPASS", None)
    worst = 0.0
    cond = None
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = ref_back_pass(oc.back_pass, c, b)
        assert div[b] == d, ("diverge", tag, div[b], d)
        for got, ref, name in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (Vx[..., b], vx, "Vx"), (Vxx[..., b], vxx, "Vxx"), (dV[:, b], dv, "dV")):
            e = relerr(got, ref)
            if e >= RTOL:
                # An ill-conditioned draw or a defect?  The second CPU restatement of the same reference code (NumPy: other summation
                # orders, same statement order) decides: the draw is compared at its own conditioning — the worst distance between the two
                # restatements over the trajectories of the draw (they share dynamics and cost; the distance of a single trajectory scatters
                # by an order of magnitude) — x3, and counted; a draw whose restatements are more than 1e-3 apart fails (nothing could be concluded).  Seen: 1.8e-8 / 1.1e-8 / 2.9e-8 on draws whose
                # restatements are 3.2e-8 / 6.7e-8 / 4.6e-8 apart (n = 4, m = 1, regType 2, horizons of 200-300 steps), 2.9e-6 on one with limits
                # whose restatements are 2e-4 apart (seed 113, case 7195).  A defect shows on
                # well-conditioned draws, where this branch changes nothing.
                # Round 4 (ADVICE): the escape is bounded.  Restatements up to COND_CAP = 1e-6 apart: the draw is judged at 3x their distance
                # and counted in `ill_conditioned`.  Further apart than that (seed 113, case 7195: 2e-4) the restatements cannot decide
                # anything: the draw is counted in `unjudged` and the kernel under test is held against the run-time-sized reference kernel
                # instead (DDP_BACKPASS=g, other arithmetic, same device): a defect of ONE kernel still shows.  The callers (the slice in
                # tests/test_gpu_fuzz_slice.py, main below) fail when either count passes its committed ceiling.
                if cond is None:
                    cond = draw_conditioning(c)
                assert e <= 3.0 * cond[name], (name, e, "C vs NumPy restatement over the draw: %.3g" % cond[name], tag, "trajectory %d" % b)
                if cond[name] <= COND_CAP:
                    one_case.ill_conditioned = getattr(one_case, "ill_conditioned", 0) + 1
                else:
                    assert cond[name] <= 1e-3, (name, e, "restatements %.3g apart: nothing can be concluded" % cond[name], tag)
                    if c["impl"] != "g":
                        os.environ["DDP_BACKPASS"] = "g"
                        gen = ddp.back_pass(c["cx"], c["cu"], c["cxx"], c["cxu"], c["cuu"], c["fx"], c["fu"], c["lam"], c["regType"], lims, None, c["u"])
                        os.environ.pop("DDP_BACKPASS", None)
                        gref = {"K": gen[1].K, "k": gen[1].k, "Vx": gen[2], "Vxx": gen[3], "dV": gen[4]}[name][..., b]
                        assert relerr(got, gref) <= 3.0 * cond[name], (name, "against the run-time-sized kernel", relerr(got, gref), cond[name], tag)
                    one_case.unjudged = getattr(one_case, "unjudged", 0) + 1
                continue
            worst = max(worst, e)
    # forward rollout with the gains just computed (LQ family), two step sizes
    fx, fu, fx_b, fx_tv, Q, R = c["fx"], c["fu"], c["fx_b"], c["fx_tv"], c["Q"], c["R"]
    prob = ddp.LQProblem(fx, fu, Q, R, dyn_batched=fx_b) if fx_tv else ddp.LQProblem(fx, fu, Q, R)
    Kc = np.where(np.isfinite(pol.K), pol.K, 0.0); kc = np.where(np.isfinite(pol.k), pol.k, 0.0)
    al = np.array([1.0, 0.3, 0.55, 0.1, 0.03, 0.8, 0.2, 0.01, 0.4, 0.6, 0.05])[: 1 + (case * 7) % 11]        # 1..11 step sizes
    xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, n, m, Kc, kc), c["x0"], c["u"], c["xnom"], al, prob, lims)
    for b in range(B):
        p = oc.make_problem("lq", n, m, N, A=fx[..., b] if fx_b else fx, B=fu[..., b] if fx_b else fu, Q=Q, R=R)
        for j, a in enumerate(al):
            xr, ur, cr = oc.forward_pass(p, (Kc[..., b], kc[..., b]), c["x0"][:, b], c["u"][..., b], c["xnom"][..., b], float(a), lims)
            for got, ref, name in ((xn[..., b, j], xr, "xnew"), (un[..., b, j], ur, "unew"), (cn[..., b, j], cr, "cnew")):
                e = relerr(got, ref)
                worst = max(worst, e)
                assert e < RTOL, (name, e, tag, "trajectory %d" % b)
    for var in ("DDP_MX2", "DDP_FORWARD_PIPE", "DDP_DPPW", "DDP_FORWARD_FAST"):
        os.environ.pop(var, None)
    return worst


def draw_conditioning(c):
    """worst distance between the C and the NumPy restatement of the reference over the trajectories of a draw, per result array"""
    from oracle import oracle_ctypes as oc
    from oracle import np_restatement as npr
    worst = {}
    for b in range(c["B"]):
        d1, (K1, k1, Q1), vx1, vxx1, dv1 = ref_back_pass(oc.back_pass, c, b)
        d2, (K2, k2, Q2), vx2, vxx2, dv2 = ref_back_pass(npr.back_pass, c, b)
        assert d1 == d2
        for name, g, r in (("K", K1, K2), ("k", k1, k2), ("Vx", vx1, vx2), ("Vxx", vxx1, vxx2), ("dV", dv1, dv2)):
            worst[name] = max(worst.get(name, 0.0), relerr(g, r))
    return worst


def conditioning(seed, case):
    """`--cond seed case`: how far the two CPU restatements of the reference are apart on that draw — the rounding-error
    amplification of the problem itself, to tell an ill-conditioned draw from a kernel defect"""
    c = gen_case(np.random.default_rng([seed, case]))
    print("case (%d, %d): %s  C vs NumPy restatement: %s" % (seed, case, {k: c[k] for k in ("n", "m", "N", "B", "regType")}, draw_conditioning(c)))


def ilqg_case(ddp, oc, rng, case):
    """whole iLQG solves on random LQ problems (own λ schedule, line search and exit per trajectory)"""
    n, m = [(10, 2), (6, 3), (4, 1), (12, 4)][rng.integers(0, 4)]
    N, B = int(rng.integers(5, 60)), int(rng.integers(1, 5))
    h = 0.02
    a0 = rng.standard_normal((n, n)); A = np.eye(n) + h * (a0 - a0.T); Bm = h * rng.standard_normal((n, m))
    Q, R = spd(rng, n, h), spd(rng, m, 0.1 * h)
    lims = None if rng.integers(0, 2) else np.stack([-rng.uniform(0.2, 2.0, m), rng.uniform(0.2, 2.0, m)], 1)
    x0 = rng.standard_normal((n, B)); u0 = 0.2 * rng.standard_normal((m, N, B))
    regType = int(rng.integers(1, 3))
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(ddp.LQProblem(A, Bm, Q, R), x0, u0, lims=lims, regType=regType)
    p = oc.make_problem("lq", n, m, N, A=A, B=Bm, Q=Q, R=R)
    worst = 0.0
    for b in range(B):
        xr, ur, polr, vxr, vxxr, cr, info = oc.ilqg(p, x0[:, b], u0[..., b], lims=lims, regType=regType)
        st = tr["stats"][:, b]
        same_path = (int(st[0]), int(st[1])) == (info["status"], info["iter"])
        # The accept / terminate tests (tol_fun, tol_grad) and, with limits, boxQP's own stopping tests sit at thresholds: a
        # rounding-level difference can move a solve one iteration or change the exit reason.  The SOLUTION must agree either way;
        # the whole trajectory to 1e-8 only when both took the same decisions and no QP tolerance is involved.
        cg, cr_ = float(cost[:, b].sum()), float(cr.sum())
        # (two solves that leave one iteration apart differ by what that iteration gained: a few tol_fun = 1e-7 in absolute terms; seen:
        #  1.07e-6 relative at cost 5.39 with control limits)
        if abs(cg - cr_) > 5e-6 * abs(cr_):
            # The reference's algorithm is not continuous in its inputs: with limits a control can sit on its bound with a multiplier that is
            # zero to rounding, boxQP's `grad > 0` then decides by the last bit whether the coordinate is clamped, K_i loses a row or not, dV
            # changes in the third digit and the line search goes another way (seed 97, case 39: states equal to 4e-16, the ORACLE's own
            # back_pass gives dV = (-1.33e-5, 6.6e-6) on its state and (-3.17e-5, 1.58e-5) on the HIP path's; case 127: the HIP solve ends on
            # tol_fun at 5.70322 after 97 rollouts where the oracle reaches 5.70131).  A defect or that?  The oracle decides: restarted from
            # the HIP path's state at the iteration where the two cost traces part (same λ, dλ: the decisions agree up to there), it has
            # to arrive where the HIP path arrived.
            assert lims is not None, ("cost", cg, cr_, case, "no limits: nothing discontinuous to blame")
            to, tg = np.asarray(info["trace"]["cost"]), tr["cost"][:, b]
            L = min(len(to), int(st[1]) + 1)
            part = [i for i in range(L) if abs(to[i] - tg[i]) > 1e-12 * abs(to[i])]
            i0 = part[0] if part else L - 1
            xa, ua, _, _, _, ca, tra = ddp.iLQG(ddp.LQProblem(A, Bm, Q, R), x0, u0, lims=lims, regType=regType, max_iter=i0)
            xo, uo, _, _, _, co, io = oc.ilqg(p, x0[:, b], u0[..., b], lims=lims, regType=regType, max_iter=i0)
            assert relerr(ua[..., b], uo) < 1e-10 and tra["stats"][5, b] == io["lam"], ("state where the traces part", case, i0)
            x2, u2, _, _, _, c2, i2 = oc.ilqg_prerolled(p, xa[..., b], ua[..., b], cost0=ca[:, b], lims=lims, regType=regType, lam=io["lam"],
                                                        dlam=io["dlam"])
            assert abs(float(c2.sum()) - cg) <= 5e-6 * abs(cg), ("cost", cg, cr_, "oracle restarted from the HIP state at iteration %d" % i0,
                                                                float(c2.sum()), case, dict(n=n, m=m, N=N, B=B, regType=regType))
            ilqg_case.knife_edge = getattr(ilqg_case, "knife_edge", 0) + 1
            continue
        if same_path and lims is None:
            for got, ref, name in ((x[..., b], xr, "x"), (u[..., b], ur, "u"), (Vxx[..., b], vxxr, "Vxx")):
                e = relerr(got, ref)
                worst = max(worst, e)
                assert e < RTOL, (name, e, case, dict(n=n, m=m, N=N, B=B, regType=regType))
    return worst


def pendcart_case(ddp, oc, rng, case):
    """the pendcart family (C3 path): rollout with limits, df (5x5 expm, all Padé orders via random h), limited back_pass"""
    N, B = int(rng.integers(3, 120)), int(rng.integers(1, 9))
    hstep = float(10.0 ** rng.uniform(-3, -1.3))                    # the reference uses 0.01; much larger Euler steps make the
                                                                    # rollout itself chaotic (sin/cos of huge angles) and nothing is comparable
    prob = ddp.PendcartProblem(h=hstep, Q=np.diag(rng.uniform(0.5, 10.0, 4)), R=np.array([[float(rng.uniform(0.1, 2.0))]]))
    lims = np.array([[-1.0, 1.0]]) * float(rng.uniform(0.5, 6.0))
    x0 = np.array([np.pi, 0, 0, 0])[:, None] + rng.uniform(-1.0, 1.0, (4, B))
    u = rng.uniform(-3.0, 3.0, (1, N, B))
    pend = dict(g=prob.g, l=prob.l, h=prob.h, d=prob.d, goal=prob.goal)
    p = oc.make_problem("pendcart", 4, 1, N, Q=prob.Q, R=prob.R, pend=pend)
    x, un, cn = ddp.forward_pass(None, x0, u, None, 1.0, prob, lims)
    x = x.reshape(4, N, B); un = un.reshape(1, N, B)
    fx, fu, _, _, _, cx, cu, cxx, cxu, cuu = ddp.df(prob, x, un)
    lam = 10.0 ** rng.uniform(-3, 1, B)
    regType = int(rng.integers(1, 3))
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, lims, None, un)
    worst = 0.0
    for b in range(B):
        xr, ur, cr = oc.forward_pass(p, None, x0[:, b], u[..., b], None, 1.0, lims)
        dr = oc.df(p, xr, ur)
        for got, ref, name in ((x[..., b], xr, "x"), (un[..., b], ur, "u"), (fx[..., b], dr[0], "fx"), (fu[..., b], dr[1], "fu"),
                               (cx[..., b], dr[2], "cx"), (cu[..., b], dr[3], "cu")):
            e = relerr(got, ref)
            worst = max(worst, e)
            assert e < RTOL, (name, e, case, dict(N=N, B=B, h=hstep))
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(dr[2], dr[3], prob.Q, np.zeros((4, 1)), prob.R, dr[0], dr[1], lam[b], regType, lims, None, ur)
        assert div[b] == d, ("diverge", case, div[b], d)
        for got, ref, name in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (Vxx[..., b], vxx, "Vxx"), (Vx[..., b], vx, "Vx")):
            e = relerr(got, ref)
            worst = max(worst, e)
            assert e < RTOL, (name, e, case, dict(N=N, B=B, h=hstep, regType=regType))
    # closed-loop rollouts (forward_pass.jl:17-24) with a random policy around the nominal trajectory, 1..6 step sizes, with and without
    # limits: the pendulum's row kernel, the generic row kernel or the lane-per-rollout kernel (forced at random)
    Kc = 0.3 * rng.standard_normal((1, 4, N, B)); kc = 0.2 * rng.standard_normal((1, N, B))
    al = np.array([1.0, 0.5, 0.25, 0.1, 0.03, 0.01])[: 1 + case % 6]
    L2 = lims if case % 2 else None
    pick = case % 4
    for var in ("DDP_FORWARD_LANE", "DDP_FORWARD_PEND"):
        os.environ.pop(var, None)
    if pick == 1:
        os.environ["DDP_FORWARD_LANE"] = "1"
    elif pick == 2:
        os.environ["DDP_FORWARD_PEND"] = "0"
    xn, un2, cn2 = ddp.forward_pass(ddp.GaussianPolicy(N, 4, 1, Kc, kc), x0, u, x, al, prob, L2)
    for var in ("DDP_FORWARD_LANE", "DDP_FORWARD_PEND"):
        os.environ.pop(var, None)
    xn = xn.reshape(4, N, B, len(al)); un2 = un2.reshape(1, N, B, len(al)); cn2 = cn2.reshape(N + 1, B, len(al))
    for b in range(B):
        for j, a_ in enumerate(al):
            xr, ur, cr = oc.forward_pass(p, (Kc[..., b], kc[..., b]), x0[:, b], u[..., b], x[..., b], float(a_), L2)
            if not np.isfinite(xr).all() or np.abs(xr).max() > 1e6:
                continue                                                    # a diverging closed loop: nothing comparable
            for got, ref, name in ((xn[..., b, j], xr, "xnew"), (un2[..., b, j], ur, "unew"), (cn2[:, b, j], cr, "cnew")):
                e = relerr(got, ref)
                worst = max(worst, e)
                assert e < 1e-7, (name, e, case, dict(N=N, B=B, h=hstep, kernel=pick, lims=L2 is not None, alpha=float(a_)))
    return worst


def gps_case(ddp, oc, rng, case):
    """KL path: ∇kl, back_pass_gps (scalar or per-step η, limits or not), forward_covariance, kl_div_wiki on a random chain"""
    import ddp_amd.kl as kl
    n, m = [(4, 2), (4, 1), (10, 2), (6, 3), (7, 2)][rng.integers(0, 5)]
    N = int(rng.integers(2, 60))
    h = 0.05
    # contractive dynamics: there is no λ on this path, and on growing ones V reaches 1e10-1e16 within the horizon, where the
    # boxQP exit codes (hence `diverge`) turn on the last bit of the KL terms
    fx = 0.97 * np.eye(n)[:, :, None] + 0.3 * h * rng.standard_normal((n, n, N)) / np.sqrt(n)
    fu = h * rng.standard_normal((n, m, N))
    cxx = np.stack([spd(rng, n, h) for _ in range(N)], -1); cuu = np.stack([spd(rng, m, 0.1 * h) for _ in range(N)], -1)
    cxu = 0.01 * h * rng.standard_normal((n, m, N))
    cx = h * rng.standard_normal((n, N)); cu = 0.1 * h * rng.standard_normal((m, N))
    x = rng.standard_normal((n, N)); u = 0.3 * rng.standard_normal((m, N))
    lims = None if rng.integers(0, 2) else np.stack([-rng.uniform(0.2, 1.0, m), rng.uniform(0.2, 1.0, m)], 1)
    Sp = np.stack([spd(rng, m, 1.0) for _ in range(N)], -1)
    Sip = np.stack([np.linalg.inv(Sp[:, :, t]) for t in range(N)], -1)
    Kp = 0.1 * rng.standard_normal((m, n, N)); kp = 0.1 * rng.standard_normal((m, N))
    prev = ddp.GaussianPolicy(N, n, m, Kp, kp, Sp, Sip)
    eta = 10.0 ** rng.uniform(0, 2)
    etab = np.array([1e-8, eta, 1e16]) if rng.integers(0, 2) else np.stack([np.full(N, 1e-8), 10.0 ** rng.uniform(0, 2, N), np.full(N, 1e16)])
    terms = kl.grad_kl(prev)
    tr = oc.kl_terms(Kp, kp, Sip)
    worst = 0.0
    for got, ref in zip(terms, tr):
        worst = max(worst, relerr(got, ref))
        assert relerr(got, ref) < RTOL, ("kl terms", case)
    d, pol, Vx, Vxx, dV = kl.back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, (terms, etab))
    dr, (Kr, kr, Quuir, Quur), vxr, vxxr, dvr = oc.back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, (tr, etab))
    if np.abs(vxxr).max() > 1e8 or not np.isfinite(vxxr).all():
        gps_case.skipped = getattr(gps_case, "skipped", 0) + 1
        return worst                      # the recursion has no λ here: a draw whose value function explodes has nothing left to compare
    if d != dr:
        # boxQP leaves with result 0 (= diverge) when its search direction is not a descent direction, `sdotg >= 0` (boxQP.jl:133):
        # at a minimiser whose gradient is rounding noise above the ABSOLUTE threshold minGrad the sign of sdotg is noise too.
        # Such a case is not comparable — it counts as a mismatch only if the C restatement keeps its answer under 1e-15 perturbations.
        alt = {oc.back_pass_gps(cx * (1 + s_), cu, cxx, cxu, cuu, fx, fu, lims, x, u, (terms if s_ == 0 else tr, etab))[0] for s_ in (0.0, 1e-15, -1e-15, 3e-15)}
        assert d in alt, ("diverge", case, d, dr, alt)
        return worst
    if d == 0:
        for got, ref, name in ((pol.K, Kr, "K"), (pol.k, kr, "k"), (pol.Σ, Quuir, "Quui"), (pol.Σi, Quur, "Quu"), (Vx, vxr, "Vx"), (Vxx, vxxr, "Vxx")):
            e = relerr(got, ref)
            worst = max(worst, e)
            # with limits k is boxQP's iterate, exact to ITS stopping tests only (1e-8 relative improvement of the QP value)
            assert e < (1e-6 if (lims is not None and name in ("k", "Vx")) else RTOL), (name, e, case, dict(n=n, m=m, N=N, lims=lims is not None, eta_tv=etab.ndim == 2))
        R1 = spd(rng, n, 1e-3)
        sig = kl.forward_covariance(kl.Model(fx, fu, R1), x, u, pol)
        sr = oc.forward_covariance(fx, R1, Kr, Quuir)
        worst = max(worst, relerr(sig, sr))
        assert relerr(sig, sr) < RTOL, ("sigma", case)
        xnew = x + 0.1 * rng.standard_normal((n, N))
        kld = kl.kl_div_wiki(xnew, x, sig, pol, prev)
        kr_ = oc.kl_div_wiki(xnew, x, sr, dict(K=Kr, k=kr, S=Quuir), dict(K=Kp, k=kp, S=Sp, Si=Sip))
        if np.ndim(kr_) and np.ndim(kld):
            fin = np.isfinite(kr_)
            worst = max(worst, relerr(kld[fin], kr_[fin]))
            assert relerr(kld[fin], kr_[fin]) < 1e-7, ("kl_div", case, dict(n=n, m=m, N=N))
    return worst


def main():
    import ddp_amd as ddp
    from oracle import oracle_ctypes as oc
    ddp.default_handle()
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    worst, at = 0.0, -1
    for c in range(cases):
        e = one_case(ddp, oc, np.random.default_rng([seed, c]), c)          # every case reproducible on its own
        if e > worst:
            worst, at = e, c
    print("fuzz: %d cases passed, worst relative error %.3g (case %d: `--cond %d %d` shows its conditioning); %d array comparisons of ill-conditioned "
          "draws made at the spread of the two CPU restatements" % (cases, worst, at, seed, at, getattr(one_case, "ill_conditioned", 0)))
    worst = 0.0
    for c in range(cases // 10):
        worst = max(worst, ilqg_case(ddp, oc, rng, c))
    print("fuzz: %d iLQG solves passed, worst relative error %.3g; %d solves with limits more than 5e-6 apart in cost (boxQP knife edge: the oracle restarted from the HIP "
          "state at the parting iteration arrives where the HIP path did)" % (cases // 10, worst, getattr(ilqg_case, "knife_edge", 0)))
    worst = 0.0
    for c in range(cases // 4):
        worst = max(worst, pendcart_case(ddp, oc, np.random.default_rng([seed, 100000 + c]), c))
    print("fuzz: %d pendcart cases passed, worst relative error %.3g" % (cases // 4, worst))
    worst = 0.0
    for c in range(cases // 4):
        worst = max(worst, gps_case(ddp, oc, np.random.default_rng([seed, 200000 + c]), c))
    print("fuzz: %d KL-path cases passed (%d exploded draws not compared), worst relative error %.3g" % (cases // 4, getattr(gps_case, "skipped", 0), worst))
    n = escape_counts()
    print("fuzz escapes: %s (ceilings per 1 000: ill-conditioned %.0f, unjudged %.0f; knife edge %.0f per 1 000 solves)" % (n, MAX_ILL_PER_1000, MAX_UNJUDGED_PER_1000, MAX_KNIFE_PER_1000))
    assert n["ill_conditioned"] <= max(2.0, MAX_ILL_PER_1000 * cases / 1000.0), n
    assert n["unjudged"] <= max(1.0, MAX_UNJUDGED_PER_1000 * cases / 1000.0), n
    assert n["knife_edge"] <= max(2.0, MAX_KNIFE_PER_1000 * (cases // 10) / 1000.0), n


def main_ilqg(solves, seed):
    """`--ilqg solves seed`: only the whole-solve part of a sweep (the same draws as `main` makes for cases = 10 solves)"""
    import ddp_amd as ddp
    from oracle import oracle_ctypes as oc
    ddp.default_handle()
    rng = np.random.default_rng(seed)
    worst = 0.0
    for c in range(solves):
        try:
            worst = max(worst, ilqg_case(ddp, oc, rng, c))
        except AssertionError as e:
            if os.environ.get("FUZZ_KEEP_GOING") != "1":
                raise
            print("FAILED", e)
    print("fuzz: %d iLQG solves passed (seed %d), worst relative error %.3g; %d on the boxQP knife edge" % (solves, seed, worst, getattr(ilqg_case, "knife_edge", 0)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cond":
        conditioning(int(sys.argv[2]), int(sys.argv[3]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--ilqg":
        main_ilqg(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
