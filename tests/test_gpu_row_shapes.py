"""The 16-lane-row backward kernel compiled for PADDED sizes (csrc/back_pass_row.hip): every shape with n <= 14, m <= 4, n + m <= 15 that
has no exact instantiation runs on it instead of the 64-lanes-per-trajectory kernel.  The reference's back_pass is size-generic
(src/backward_pass.jl:162-252 + :28-79): every case is compared with the C oracle on every trajectory, for the three rank dispatches
(LTI :217, LTV / TI-cost :162, LTV / TV-cost :179), both regularisations, with and without control limits, per-trajectory operands,
inactive trajectories and a diverging λ; the general kernel (DDP_BACKPASS=general) gives the second opinion."""
import os

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8
# (n, m): odd and even n, every padded size NP = 4 .. 14, every MP = 1, 2, 3/4; (10,2) and (4,1) are forced onto the row kernel too
SHAPES = [(1, 1), (2, 1), (3, 1), (3, 2), (4, 1), (4, 3), (5, 2), (6, 2), (6, 3), (7, 3), (7, 4), (8, 1), (8, 4), (9, 2), (10, 2), (10, 4),
          (11, 3), (12, 2), (12, 3), (13, 1), (14, 1)]


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    ddp_amd.default_handle()
    return ddp_amd


def _problem(rng, n, m, N, B, kind):
    """kind: 'lti' one (fx, fu, cxx, cxu, cuu); 'ltv' fx, fu [.., N], cost time-invariant; 'tv' everything [.., N]; 'btv' [.., N, B]"""
    import scipy.linalg as sla

    def spd(d, s):
        a = rng.standard_normal((d, d)); return s * (a @ a.T / d + 0.5 * np.eye(d))
    A0 = rng.standard_normal((n, n))
    A = sla.expm(0.1 * (A0 - A0.T)) * rng.uniform(0.95, 1.02)
    Bm = 0.3 * rng.standard_normal((n, m))
    cxx, cuu, cxu = spd(n, 0.2), spd(m, 0.1), 0.02 * rng.standard_normal((n, m))
    tshape = {"lti": (), "ltv": (N,), "tv": (N,), "btv": (N, B)}[kind]
    cshape = {"lti": (), "ltv": (), "tv": (N,), "btv": (N, B)}[kind]

    def vary(M, shape, amp):
        if not shape:
            return M
        out = M.reshape(M.shape + (1,) * len(shape)) * (1 + amp * rng.standard_normal((1,) * M.ndim + shape))
        return np.ascontiguousarray(out)
    fx, fu = vary(A, tshape, 0.02), vary(Bm, tshape, 0.05)
    cxxv = vary(cxx, cshape, 0.0) * (1 + 0.1 * rng.uniform(size=(1, 1) + cshape)) if cshape else cxx
    cuuv = vary(cuu, cshape, 0.0) * (1 + 0.1 * rng.uniform(size=(1, 1) + cshape)) if cshape else cuu
    cxuv = vary(cxu, cshape, 0.1)
    cx = 0.3 * rng.standard_normal((n, N, B)); cu = 0.2 * rng.standard_normal((m, N, B))
    u = 0.3 * rng.standard_normal((m, N, B)); x = np.zeros((n, N, B))
    return cx, cu, cxxv, cxuv, cuuv, fx, fu, x, u


def _slice(M, b, batched):
    return M[..., b] if batched else M


def _check(ddp, out, args, lam, regType, L, batched, who=None):
    from oracle import oracle_ctypes as oc
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    div, pol, Vx, Vxx, dV = out
    B = cx.shape[-1]
    lam = np.broadcast_to(np.asarray(lam, float), (B,))
    for b in (range(B) if who is None else who):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], _slice(cxx, b, batched), _slice(cxu, b, batched),
                                                  _slice(cuu, b, batched), _slice(fx, b, batched), _slice(fu, b, batched), lam[b], regType, L,
                                                  x[..., b], u[..., b])
        assert d == div[b], (b, d, div[b])
        for got, ref, name in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (Vx[..., b], vx, "Vx"), (Vxx[..., b], vxx, "Vxx"),
                               (dV[:, b], dv, "dV")):
            assert relerr(got, ref) < RTOL, (name, b, relerr(got, ref))
        if d == 0:
            assert relerr(pol.Σi[..., b], Quu) < RTOL, ("Quu", b)
            assert np.array_equal(Vxx[..., b], np.transpose(Vxx[..., b], (1, 0, 2)))
        else:
            assert not pol.K[:, :, : d - 1, b].any() and not Vxx[:, :, : d - 1, b].any() and not Vx[:, : d - 1, b].any()


def _default_kernel(n, m, lims, B=1):
    """what the dispatcher picks for a small batch of a shape without an exact instantiation: the fp64 tile kernels (run-time sizes inside
    the (10, 2) tile for m <= 2 without limits; one tile for n <= 12, m <= 4, with or without limits), else the row kernel"""
    if lims is None and n <= 10 and m <= 2:
        return "back_pass_mx_kernel<RT>"
    wide_lims = 512 if m == 1 else (2048 if (n > 8 or m >= 3 or n <= 4) else 1024)   # (back_pass.hip: measured cross-over with the row kernels)
    if n <= 12 and m <= (4 if n <= 8 else 3) and n + m <= 15 and (lims is None or B <= wide_lims):   # with limits: the box-QP as a wave-uniform solve
        return "back_pass_mxg_kernel"
    return "back_pass_row_kernel"


def _run(ddp, args, lam, regType, L, force):
    from ddp_amd import _lib
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    os.environ["DDP_BACKPASS"] = force
    try:
        out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, L, x, u)
        name = _lib.default_handle().last_kernel(0)
    finally:
        del os.environ["DDP_BACKPASS"]
        _lib.default_handle().raw
    return out, name


@pytest.mark.parametrize("n,m", SHAPES)
@pytest.mark.parametrize("kind", ["lti", "ltv", "tv"])
def test_row_kernel_every_shape_vs_oracle(ddp, n, m, kind):
    rng = np.random.default_rng(1000 * n + 10 * m + len(kind))
    N, B = 23, 11                                     # B not a multiple of 4: a wave with idle rows; N not a multiple of the ring
    args = _problem(rng, n, m, N, B, kind)
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    for regType, lims in ((1, False), (2, True)):
        L = np.stack([-0.25 * np.ones(m), 0.3 * np.ones(m)], 1) if lims else None
        out, name = _run(ddp, args, lam, regType, L, "row")
        assert name == "back_pass_row_kernel", name
        _check(ddp, out, args, lam, regType, L, False)


@pytest.mark.parametrize("n,m", [(3, 1), (5, 2), (6, 3), (9, 2), (12, 3), (13, 1)])
def test_row_kernel_is_the_default_dispatch(ddp, n, m):
    """no switch set, control limits: up to n = 12 the wide tile kernel takes a small batch (the box-QP as a wave-uniform solve), the row
    kernel the rest; both agree with the general kernel far below the oracle tolerance"""
    from ddp_amd import _lib
    rng = np.random.default_rng(7 * n + m)
    N, B = 40, 9
    args = _problem(rng, n, m, N, B, "ltv")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    L = np.stack([-0.3 * np.ones(m), 0.3 * np.ones(m)], 1)
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.3, 1, L, x, u)
    assert _lib.default_handle().last_kernel(0) == _default_kernel(n, m, L)
    row, name = _run(ddp, args, 0.3, 1, L, "row")
    assert name == "back_pass_row_kernel"
    for a_, b_ in ((out[1].K, row[1].K), (out[1].k, row[1].k), (out[2], row[2]), (out[3], row[3]), (out[4], row[4])):
        assert relerr(a_, b_) < 1e-10
    ref, name = _run(ddp, args, 0.3, 1, L, "general")
    assert name == "back_pass_kernel"
    assert np.array_equal(out[0], ref[0])
    for a_, b_ in ((out[1].K, ref[1].K), (out[1].k, ref[1].k), (out[2], ref[2]), (out[3], ref[3]), (out[4], ref[4]), (out[1].Σi, ref[1].Σi)):
        assert relerr(a_, b_) < 1e-10
    _check(ddp, out, args, 0.3, 1, L, False)


@pytest.mark.parametrize("n,m", [(5, 2), (7, 3), (12, 3)])
def test_row_kernel_per_trajectory_operands_inactive_and_divergence(ddp, n, m):
    """a3 layout with a batch axis on every operand; trajectories switched off keep their buffers; a hugely negative λ makes QuuF
    indefinite at once (diverge = N - 1, everything below zero-filled) for some trajectories only"""
    rng = np.random.default_rng(31 * n + m)
    N, B = 19, 10
    args = _problem(rng, n, m, N, B, "btv")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    lam = np.full(B, 0.2); lam[[1, 6]] = -50.0
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, 1, None, x, u)
    assert out[0][1] == N - 1 and out[0][6] == N - 1 and out[0][0] == 0
    _check(ddp, out, args, lam, 1, None, True)
    L = np.stack([-0.2 * np.ones(m), 0.2 * np.ones(m)], 1)
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, np.abs(lam), 2, L, x, u)
    _check(ddp, out, args, np.abs(lam), 2, L, True)


def test_row_kernel_full_size_off_shapes(ddp):
    """the two off-shape lines of bench.py's other_configs at full size (default dispatch: the wide tile kernel for the one without limits,
    the row kernel for the one with), and the row kernel forced at the first: a sample of trajectories against the oracle"""
    from ddp_amd import _lib
    rng = np.random.default_rng(5150)
    for n, m, N, B, kind, lims in ((12, 3, 500, 2048, "ltv", False), (6, 2, 1000, 4096, "lti", True)):
        args = _problem(rng, n, m, N, B, kind)
        cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
        L = np.stack([-0.3 * np.ones(m), 0.3 * np.ones(m)], 1) if lims else None
        out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.1, 1, L, x, u)
        assert _lib.default_handle().last_kernel(0) == _default_kernel(n, m, L, B)
        who = sorted({0, 1, 2, 3, B - 1, B - 2} | set(int(v) for v in rng.integers(0, B, 18)))
        _check(ddp, out, args, 0.1, 1, L, False, who=who)
        if not lims:
            out, name = _run(ddp, args, 0.1, 1, L, "row")
            assert name == "back_pass_row_kernel"
            _check(ddp, out, args, 0.1, 1, L, False, who=who)


# ------------------------------------------------------------------------------------------------------------------ forward_pass
FSHAPES = [(1, 1), (2, 1), (3, 2), (4, 3), (5, 2), (6, 2), (6, 3), (7, 4), (8, 1), (9, 2), (10, 4), (11, 3), (12, 2), (12, 4), (13, 1), (14, 2)]


def _lq(rng, n, m, N, B, ltv, batched):
    import scipy.linalg as sla
    A0 = rng.standard_normal((n, n))
    A = sla.expm(0.1 * (A0 - A0.T)) * 0.99
    Bm = 0.2 * rng.standard_normal((n, m))
    q = rng.standard_normal((n, n)); Q = 0.1 * (q @ q.T / n + 0.3 * np.eye(n))          # FULL Q, R: the separate cost kernel
    r = rng.standard_normal((m, m)); R = 0.05 * (r @ r.T / m + 0.3 * np.eye(m))
    shape = ((N,) if ltv else ()) + ((B,) if batched else ())
    if shape:
        A = np.ascontiguousarray(A.reshape(n, n, *([1] * len(shape))) * (1 + 0.02 * rng.standard_normal((1, 1) + shape)))
        Bm = np.ascontiguousarray(Bm.reshape(n, m, *([1] * len(shape))) * (1 + 0.05 * rng.standard_normal((1, 1) + shape)))
    return A, Bm, Q, R


@pytest.mark.parametrize("n,m", FSHAPES)
def test_forward_row_kernel_every_shape_vs_oracle(ddp, n, m):
    """src/forward_pass.jl:9-33 at sizes no exact kernel exists for: closed-loop rollouts of 7 step sizes with limits, the initial
    rollout (empty policy), LTI and LTV dynamics, on the padded row kernel; every rollout against the C oracle"""
    from ddp_amd import _lib
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(77 * n + m)
    N, B = 29, 6
    alphas = 10.0 ** np.linspace(0, -3, 7)
    L = np.stack([-0.4 * np.ones(m), 0.5 * np.ones(m)], 1)
    for ltv in (False, True):
        A, Bm, Q, R = _lq(rng, n, m, N, B, ltv, False)
        prob = ddp.LQProblem(A, Bm, Q, R)
        po = oc.make_problem("lq", n, m, N, A=A, B=Bm, Q=Q, R=R)
        x0 = rng.standard_normal((n, B)); u = 0.3 * rng.standard_normal((m, N, B))
        K = 0.2 * rng.standard_normal((m, n, N, B)); k = 0.1 * rng.standard_normal((m, N, B))
        x, _, _ = ddp.forward_pass(ddp.GaussianPolicy(), x0, u, None, 1.0, prob, None)
        if not (n == 10 and m == 2):
            assert _lib.default_handle().last_kernel(1) == "forward_row_kernel"
        for b in range(B):
            xr, ur, cr = oc.forward_pass(po, None, x0[:, b], u[..., b], None, 1.0, None)
            assert relerr(x[..., b], xr) < 1e-10
        for lims in (None, L):
            pol = ddp.GaussianPolicy(N, n, m, K, k)
            xn, un, cn = ddp.forward_pass(pol, x0, u, x, alphas, prob, lims)
            for b in range(B):
                for ai, al in enumerate(alphas):
                    xr, ur, cr = oc.forward_pass(po, (K[..., b], k[..., b]), x0[:, b], u[..., b], x[..., b], float(al), lims)
                    assert relerr(xn[..., b, ai], xr) < RTOL and relerr(un[..., b, ai], ur) < RTOL and relerr(cn[:, b, ai], cr) < RTOL, (n, m, ltv, b, ai)


@pytest.mark.parametrize("n,m", [(5, 2), (12, 3)])
def test_forward_row_kernel_agrees_with_the_group_kernel(ddp, n, m):
    """the run-time-sized kernel (DDP_FORWARD=group) on the same rollouts, per-trajectory time-varying dynamics, inactive trajectories"""
    from ddp_amd import _lib
    rng = np.random.default_rng(5 * n + m)
    N, B = 33, 9
    A, Bm, Q, R = _lq(rng, n, m, N, B, True, True)
    prob = ddp.LQProblem(A, Bm, Q, R, dyn_batched=True)
    x0 = rng.standard_normal((n, B)); u = 0.3 * rng.standard_normal((m, N, B))
    K = 0.2 * rng.standard_normal((m, n, N, B)); k = 0.1 * rng.standard_normal((m, N, B))
    x, _, _ = ddp.forward_pass(ddp.GaussianPolicy(), x0, u, None, 1.0, prob, None)
    pol = ddp.GaussianPolicy(N, n, m, K, k)
    al = np.array([1.0, 0.3, 0.01])
    got = ddp.forward_pass(pol, x0, u, x, al, prob, None)
    assert _lib.default_handle().last_kernel(1) == "forward_row_kernel"
    os.environ["DDP_FORWARD"] = "group"
    try:
        ref = ddp.forward_pass(pol, x0, u, x, al, prob, None)
        assert _lib.default_handle().last_kernel(1) == "forward_pass_kernel"
    finally:
        del os.environ["DDP_FORWARD"]
        _lib.default_handle().raw
    for a_, b_ in zip(got, ref):
        assert relerr(a_, b_) < 1e-11


@pytest.mark.parametrize("n,m", [(6, 2), (12, 3)])
def test_ilqg_solves_at_an_off_shape_match_the_oracle(ddp, n, m):
    """whole iLQG solves (src/iLQG.jl:143-341) of an LQ problem of demo_linear's recipe at (n, m): both passes on the row kernels"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(900 + n)
    N, B = 60, 5
    P = npr.make_lq_problem(rng, n=n, m=m, T=N)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B)); u0 = 0.1 * rng.standard_normal((m, N, B))
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(prob, x0, u0)
    po = oc.make_problem("lq", n, m, N, A=P["A"], B=P["B"], Q=P["Q"], R=P["R"])
    for b in range(B):
        xr, ur, (Kr, kr, Quur), Vxr, Vxxr, cr, info = oc.ilqg(po, x0[:, b], u0[:, :, b])
        assert int(tr["stats"][0, b]) == info["status"] and abs(int(tr["stats"][1, b]) - info["iter"]) <= 1
        for got, ref in ((x[..., b], xr), (u[..., b], ur), (cost[:, b], cr)):
            assert relerr(got, ref) < 1e-7


@pytest.mark.parametrize("N", [1, 2, 3, 5, 9])
@pytest.mark.parametrize("n,m", [(3, 2), (7, 3), (12, 2)])
def test_row_kernels_short_horizons(ddp, n, m, N):
    """N = 1 (only the terminal step, backward_pass.jl:21-23), N = 2 .. 9: shorter than the prefetch rings and the unrolled groups of both
    row kernels; B = 1 and B = 5 (one wave with idle rows / two waves)"""
    from ddp_amd import _lib
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(1000 * n + 10 * m + N)
    for B in (1, 5):
        args = _problem(rng, n, m, N, B, "ltv")
        cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
        L = np.stack([-0.3 * np.ones(m), 0.3 * np.ones(m)], 1)
        for lims in (None, L):
            out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.2, 1, lims, x, u)
            assert _lib.default_handle().last_kernel(0) == _default_kernel(n, m, lims)
            _check(ddp, out, args, 0.2, 1, lims, False)
        A, Bm, Q, R = _lq(rng, n, m, N, B, True, False)
        prob = ddp.LQProblem(A, Bm, Q, R)
        po = oc.make_problem("lq", n, m, N, A=A, B=Bm, Q=Q, R=R)
        x0 = rng.standard_normal((n, B)); uu = 0.3 * rng.standard_normal((m, N, B))
        K = 0.2 * rng.standard_normal((m, n, N, B)); k = 0.1 * rng.standard_normal((m, N, B))
        xr0, _, _ = ddp.forward_pass(ddp.GaussianPolicy(), x0, uu, None, 1.0, prob, None)
        xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, n, m, K, k), x0, uu, xr0, np.array([1.0, 0.1]), prob, L)
        assert _lib.default_handle().last_kernel(1) == "forward_row_kernel"
        for b in range(B):
            for ai, al in enumerate((1.0, 0.1)):
                xr, ur, cr = oc.forward_pass(po, (K[..., b], k[..., b]), x0[:, b], uu[..., b], xr0[..., b], al, L)
                assert relerr(xn[..., b, ai], xr) < RTOL and relerr(un[..., b, ai], ur) < RTOL and relerr(cn[:, b, ai], cr) < RTOL


# ------------------------------------------------------------------------------------------------- 14 < n <= 32 (or m > 4): back_pass_mid.hip
MID_SHAPES = [(15, 1), (16, 2), (17, 3), (20, 6), (24, 4), (25, 8), (31, 5), (32, 8), (13, 4), (8, 5), (3, 7), (32, 1), (16, 8), (12, 4)]


@pytest.mark.parametrize("n,m", MID_SHAPES)
@pytest.mark.parametrize("kind", ["lti", "ltv", "tv"])
def test_mid_kernel_every_shape_vs_oracle(ddp, n, m, kind):
    """one wave per trajectory, products on the fp64 matrix cores with LDS operands (csrc/back_pass_mid.hip): every shape up to n = 32,
    m = 8 — incl. odd n, m > 4 below n = 14, both tile counts and both sizes of the m x m system — against the C oracle"""
    rng = np.random.default_rng(2000 * n + 10 * m + len(kind))
    N, B = 21, 5
    args = _problem(rng, n, m, N, B, kind)
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    for regType, lims in ((1, False), (2, True), (1, True)):
        L = np.stack([-0.25 * np.ones(m), 0.3 * np.ones(m)], 1) if lims else None
        out, name = _run(ddp, args, lam, regType, L, "mid")
        assert name == "back_pass_mid_kernel", name
        _check(ddp, out, args, lam, regType, L, False)


@pytest.mark.parametrize("n,m", [(16, 2), (24, 4), (32, 8), (9, 6)])
def test_mid_kernel_default_dispatch_batched_operands_inactive_divergence(ddp, n, m):
    from ddp_amd import _lib
    rng = np.random.default_rng(41 * n + m)
    N, B = 17, 7
    args = _problem(rng, n, m, N, B, "btv")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    lam = np.full(B, 0.2); lam[[2, 5]] = -50.0
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, 1, None, x, u)
    assert _lib.default_handle().last_kernel(0) == "back_pass_mid_kernel"
    assert out[0][2] == N - 1 and out[0][5] == N - 1 and out[0][0] == 0
    _check(ddp, out, args, lam, 1, None, True)
    ref, name = _run(ddp, args, np.abs(lam), 2, None, "general")
    assert name == "back_pass_kernel"
    got = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, np.abs(lam), 2, None, x, u)
    for a_, b_ in ((got[1].K, ref[1].K), (got[1].k, ref[1].k), (got[2], ref[2]), (got[3], ref[3]), (got[4], ref[4]), (got[1].Σi, ref[1].Σi)):
        assert relerr(a_, b_) < 1e-9


@pytest.mark.parametrize("N", [1, 2, 3])
def test_mid_kernel_short_horizons(ddp, N):
    rng = np.random.default_rng(17 + N)
    for n, m in ((18, 3), (30, 7)):
        args = _problem(rng, n, m, N, 3, "ltv")
        out, name = _run(ddp, args, 0.3, 1, None, "mid")
        assert name == "back_pass_mid_kernel"
        _check(ddp, out, args, 0.3, 1, None, False)


# ------------------------------------------------------------------------------------- 14 < n <= 32 (or m > 4): forward_mid_kernel
MID_FSHAPES = [(15, 1), (16, 2), (17, 3), (20, 6), (24, 4), (25, 8), (31, 5), (32, 8), (8, 5), (3, 7), (32, 1), (16, 8), (23, 7),
               (13, 3), (13, 4), (14, 3), (14, 4)]      # the last four: declined by the row launcher (n > 12 with m > 2), ADVICE r5


@pytest.mark.parametrize("n,m", MID_FSHAPES)
def test_forward_mid_kernel_every_shape_vs_oracle(ddp, n, m):
    """src/forward_pass.jl:9-33 at the sizes no 16-lane row holds (csrc/forward_pass_big.hip, forward_mid_kernel: one wave per rollout,
    the operands of a step requested a step ahead): 5 step sizes with and without limits, the initial rollout (empty policy), LTI and
    LTV dynamics, full Q and R; every rollout against the C oracle"""
    from ddp_amd import _lib
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(91 * n + m)
    N, B = 23, 4
    alphas = 10.0 ** np.linspace(0, -3, 5)
    L = np.stack([-0.4 * np.ones(m), 0.5 * np.ones(m)], 1)
    for ltv in (False, True):
        A, Bm, Q, R = _lq(rng, n, m, N, B, ltv, False)
        prob = ddp.LQProblem(A, Bm, Q, R)
        po = oc.make_problem("lq", n, m, N, A=A, B=Bm, Q=Q, R=R)
        x0 = rng.standard_normal((n, B)); u = 0.3 * rng.standard_normal((m, N, B))
        K = 0.2 * rng.standard_normal((m, n, N, B)) / np.sqrt(n); k = 0.1 * rng.standard_normal((m, N, B))
        x, u1, c1 = ddp.forward_pass(ddp.GaussianPolicy(), x0, u, None, 1.0, prob, None)
        assert _lib.default_handle().last_kernel(1) == "forward_mid_kernel"
        for b in range(B):
            xr, ur, cr = oc.forward_pass(po, None, x0[:, b], u[..., b], None, 1.0, None)
            assert relerr(x[..., b], xr) < 1e-10 and relerr(c1[:, b], cr) < 1e-10
        for lims in (None, L):
            pol = ddp.GaussianPolicy(N, n, m, K, k)
            xn, un, cn = ddp.forward_pass(pol, x0, u, x, alphas, prob, lims)
            assert _lib.default_handle().last_kernel(1) == "forward_mid_kernel"
            for b in range(B):
                for ai, al in enumerate(alphas):
                    xr, ur, cr = oc.forward_pass(po, (K[..., b], k[..., b]), x0[:, b], u[..., b], x[..., b], float(al), lims)
                    assert relerr(xn[..., b, ai], xr) < RTOL and relerr(un[..., b, ai], ur) < RTOL and relerr(cn[:, b, ai], cr) < RTOL, (n, m, ltv, b, ai)


@pytest.mark.parametrize("n,m", [(18, 3), (24, 4), (32, 8), (9, 6)])
def test_forward_mid_kernel_agrees_with_the_run_time_sized_kernel(ddp, n, m):
    """DDP_FORWARD_MID=0 keeps forward_big_kernel / cost_rt_kernel: the same rollouts with per-trajectory time-varying dynamics, an
    a NaN control (zeroed inside f, forward_pass.jl:21), odd and even horizons (one, exactly one and three 64-step cost chunks); the
    states differ by the order of the row sums only"""
    from ddp_amd import _lib
    rng = np.random.default_rng(7 * n + m)
    for N in (33, 64, 130):
        B = 6
        A, Bm, Q, R = _lq(rng, n, m, N, B, True, True)
        prob = ddp.LQProblem(A, Bm, Q, R, dyn_batched=True)
        x0 = rng.standard_normal((n, B)); u = 0.3 * rng.standard_normal((m, N, B))
        u[0, 5, 1] = np.nan
        K = 0.2 * rng.standard_normal((m, n, N, B)) / np.sqrt(n); k = 0.1 * rng.standard_normal((m, N, B))
        x, _, _ = ddp.forward_pass(ddp.GaussianPolicy(), x0, u, None, 1.0, prob, None)
        pol = ddp.GaussianPolicy(N, n, m, K, k)
        al = np.array([1.0, 0.3, 0.01])
        got = ddp.forward_pass(pol, x0, u, x, al, prob, None)
        assert _lib.default_handle().last_kernel(1) == "forward_mid_kernel"
        os.environ["DDP_FORWARD_MID"] = "0"
        try:
            ref = ddp.forward_pass(pol, x0, u, x, al, prob, None)
            assert _lib.default_handle().last_kernel(1) == "forward_big_kernel"
        finally:
            del os.environ["DDP_FORWARD_MID"]
            _lib.default_handle().raw
        assert np.isfinite(got[0]).all() and got[1][0, 5, 1, 0] == 0.0
        assert np.array_equal(got[1][..., 0], ref[1][..., 0]) or relerr(got[1], ref[1]) < 1e-11
        for a_, b_ in zip(got, ref):
            assert relerr(a_, b_) < 1e-11


@pytest.mark.parametrize("N", [1, 2, 3])
def test_forward_mid_kernel_short_horizons(ddp, N):
    from ddp_amd import _lib
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(300 + N)
    for n, m in ((18, 3), (30, 7)):
        B = 3
        A, Bm, Q, R = _lq(rng, n, m, N, B, True, False)
        prob = ddp.LQProblem(A, Bm, Q, R)
        po = oc.make_problem("lq", n, m, N, A=A, B=Bm, Q=Q, R=R)
        x0 = rng.standard_normal((n, B)); uu = 0.3 * rng.standard_normal((m, N, B))
        K = 0.2 * rng.standard_normal((m, n, N, B)) / np.sqrt(n); k = 0.1 * rng.standard_normal((m, N, B))
        xr0, _, _ = ddp.forward_pass(ddp.GaussianPolicy(), x0, uu, None, 1.0, prob, None)
        xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, n, m, K, k), x0, uu, xr0, np.array([1.0, 0.1]), prob, None)
        assert _lib.default_handle().last_kernel(1) == "forward_mid_kernel"
        for b in range(B):
            for ai, al in enumerate((1.0, 0.1)):
                xr, ur, cr = oc.forward_pass(po, (K[..., b], k[..., b]), x0[:, b], uu[..., b], xr0[..., b], al, None)
                assert relerr(xn[..., b, ai], xr) < RTOL and relerr(un[..., b, ai], ur) < RTOL and relerr(cn[:, b, ai], cr) < RTOL


@pytest.mark.parametrize("n,m", [(18, 3), (24, 4)])
def test_ilqg_solves_at_a_mid_shape_match_the_oracle(ddp, n, m):
    """whole iLQG solves (src/iLQG.jl:143-341) at sizes above the 16-lane rows: back_pass_mid_kernel + forward_mid_kernel under the
    device-resident driver (activity masks, all step sizes per launch)"""
    from ddp_amd import _lib
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(1900 + n)
    N, B = 40, 4
    P = npr.make_lq_problem(rng, n=n, m=m, T=N)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B)); u0 = 0.1 * rng.standard_normal((m, N, B))
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(prob, x0, u0)
    assert _lib.default_handle().last_kernel(1) == "forward_mid_kernel"
    po = oc.make_problem("lq", n, m, N, A=P["A"], B=P["B"], Q=P["Q"], R=P["R"])
    for b in range(B):
        xr, ur, (Kr, kr, Quur), Vxr, Vxxr, cr, info = oc.ilqg(po, x0[:, b], u0[:, :, b])
        assert int(tr["stats"][0, b]) == info["status"] and abs(int(tr["stats"][1, b]) - info["iter"]) <= 1
        for got, ref in ((x[..., b], xr), (u[..., b], ur), (cost[:, b], cr)):
            assert relerr(got, ref) < 1e-7
