"""GPU parity tests for large states (BASELINE config 4 shape: n=64, m=8, per-trajectory LTV dynamics) against the
CPU oracle.  Sizes are reduced (N, B) so the oracle finishes in seconds; the kernels are size-generic."""
import numpy as np
import pytest
import scipy.linalg as sla

from conftest import relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    ddp_amd.default_handle()
    return ddp_amd


def _problem(rng, n, m, N, B, tv_cost):
    h = 0.02
    fx = np.empty((n, n, N, B)); fu = np.empty((n, m, N, B))
    for b in range(B):
        a0 = rng.standard_normal((n, n)) / np.sqrt(n)
        A = sla.expm(h * (a0 - a0.T))
        for t in range(N):
            fx[:, :, t, b] = A * (1.0 + 0.01 * np.sin(0.1 * t + b))
        fu[:, :, :, b] = h * rng.standard_normal((n, m, N))

    def spd(d, s):
        a = rng.standard_normal((d, d)); return s * (a @ a.T / d + 0.5 * np.eye(d))
    if tv_cost:
        cxx = np.stack([np.stack([spd(n, h) for _ in range(N)], -1) for _ in range(B)], -1)
        cuu = np.stack([np.stack([spd(m, 0.1 * h) for _ in range(N)], -1) for _ in range(B)], -1)
        cxu = 0.01 * h * rng.standard_normal((n, m, N, B))
    else:
        cxx, cuu, cxu = spd(n, h), spd(m, 0.1 * h), 0.01 * h * rng.standard_normal((n, m))
    cx = h * rng.standard_normal((n, N, B)); cu = 0.1 * h * rng.standard_normal((m, N, B))
    u = 0.3 * rng.standard_normal((m, N, B))
    return cx, cu, cxx, cxu, cuu, fx, fu, u


@pytest.mark.parametrize("n,m,impl", [(64, 8, "auto"), (64, 8, "big"), (64, 8, "old"), (64, 8, "new"), (40, 4, "auto"),      # auto = the run-time-sized matrix-core kernel (mf2)
                                      (33, 2, "auto"), (35, 3, "auto"), (63, 7, "auto"), (40, 3, "auto"), (47, 8, "auto"),   # 3 tiles (n <= 48) and 4 tiles, odd sizes
                                      (48, 6, "auto"), (49, 1, "auto"), (64, 1, "auto"), (33, 8, "auto"), (48, 8, "auto"), (57, 5, "auto"),
                                      (35, 3, "big")])                                                                       # odd sizes on the vector kernel: padded copies
@pytest.mark.parametrize("tv_cost", [False, True])
@pytest.mark.parametrize("regType,lims", [(1, False), (2, False), (1, True)])
def test_back_pass_large(ddp, monkeypatch, n, m, impl, tv_cost, regType, lims):
    from oracle import oracle_ctypes as oc
    if impl != "auto":
        monkeypatch.setenv("DDP_BACKPASS", impl)
    rng = np.random.default_rng(n + 10 * regType + tv_cost)
    N, B = 12, 3
    cx, cu, cxx, cxu, cuu, fx, fu, u = _problem(rng, n, m, N, B, tv_cost)
    L = np.stack([-0.25 * np.ones(m), 0.4 * np.ones(m)], 1) if lims else None
    lam = np.array([1e-3, 0.1, 2.0])
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, L, None, u)
    from ddp_amd import _lib
    want = {"auto": "back_pass_mf2_kernel", "new": "back_pass_mf2_kernel", "old": "back_pass_mfma_kernel", "big": "back_pass_big_kernel"}[impl]
    if impl == "auto" and (lims or tv_cost) and (n, m) == (64, 8):
        want = "back_pass_mfma_kernel"             # the exact shape with limits or a time-varying cost: the round-5 kernel is the faster one (back_pass.hip)
    assert _lib.default_handle().last_kernel(0) == want or (impl == "big" and n % 2), _lib.default_handle().last_kernel(0)      # (odd sizes forced onto the vector kernel: the padded launcher)
    assert np.array_equal(Vxx, np.transpose(Vxx, (1, 0, 2, 3)))
    for b in range(B):
        sl = lambda a_, nd: a_[..., b] if a_.ndim == nd + 1 else a_
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], sl(cxx, 3), sl(cxu, 3), sl(cuu, 3), fx[..., b], fu[..., b],
                                                  lam[b], regType, L, None, u[..., b])
        assert div[b] == d == 0
        for got, ref, name in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (Vx[..., b], vx, "Vx"), (Vxx[..., b], vxx, "Vxx"),
                               (dV[:, b], dv, "dV"), (pol.Σi[..., b], Quu, "Quu")):
            assert relerr(got, ref) < RTOL, (name, relerr(got, ref))


def test_back_pass_large_divergence(ddp):
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(3)
    n, m, N, B = 64, 8, 10, 2
    cx, cu, cxx, cxu, cuu, fx, fu, u = _problem(rng, n, m, N, B, True)
    cuu[:, :, 4, 1] = -np.eye(m)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 1e-4, 1, None, None, u)
    assert list(div) == [0, 5]
    d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., 1], cu[..., 1], cxx[..., 1], cxu[..., 1], cuu[..., 1], fx[..., 1], fu[..., 1],
                                              1e-4, 1, None, None, u[..., 1])
    assert d == 5 and relerr(Vxx[..., 1], vxx) < RTOL and relerr(pol.K[..., 1], K) < RTOL
    assert not Vxx[:, :, :4, 1].any()


@pytest.mark.parametrize("fwd64", ["1", "0"])      # the n=64/m=8 streaming kernel, and the run-time-sized one on the same shape
@pytest.mark.parametrize("nalpha", [1, 2, 5])       # 1 / 2 / 4 step sizes per wave (5 = one full group + a group with dead slots)
@pytest.mark.parametrize("lims", [False, True])
def test_forward_pass_large(ddp, monkeypatch, lims, fwd64, nalpha):
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_FORWARD64", fwd64)
    rng = np.random.default_rng(8)
    n, m, N, B = 64, 8, 21, 3                      # odd N: the two-step unrolled loop ends on its tail
    cx, cu, cxx, cxu, cuu, fx, fu, u = _problem(rng, n, m, N, B, False)
    Q, R = cxx, cuu
    x0 = rng.standard_normal((n, B))
    K = 0.05 * rng.standard_normal((m, n, N, B)); k = 0.1 * rng.standard_normal((m, N, B))
    x = rng.standard_normal((n, N, B))
    L = np.stack([-0.3 * np.ones(m), 0.35 * np.ones(m)], 1) if lims else None
    prob = ddp.LQProblem(fx, fu, Q, R, dyn_batched=True)
    pol = ddp.GaussianPolicy(N, n, m, K, k)
    alphas = np.array([1.0, 0.4, 0.1, 0.03, 0.55])[:nalpha]
    xn, un, cn = ddp.forward_pass(pol, x0, u, x, alphas, prob, L)
    for b in range(B):
        p = oc.make_problem("lq", n, m, N, A=fx[..., b], B=fu[..., b], Q=Q, R=R)
        for j, a in enumerate(alphas):
            xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], x[..., b], float(a), L)
            assert relerr(xn[..., b, j], xr) < RTOL and relerr(un[..., b, j], ur) < RTOL and relerr(cn[..., b, j], cr) < RTOL


def test_ilqg_large_state_batched(ddp):
    """whole iteration on the C4 shape (n=64, m=8, per-trajectory LTV dynamics), reduced N and B"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(12)
    n, m, N, B = 64, 8, 24, 3
    cx, cu, cxx, cxu, cuu, fx, fu, u = _problem(rng, n, m, N, B, False)
    Q, R = 0.02 * np.eye(n), 0.002 * np.eye(m)
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
    u0 = 0.1 * rng.standard_normal((m, N, B))
    prob = ddp.LQProblem(fx, fu, Q, R, dyn_batched=True)
    x, uu, pol, Vx, Vxx, cost, tr = ddp.iLQG(prob, x0, u0)
    for b in range(B):
        p = oc.make_problem("lq", n, m, N, A=fx[..., b], B=fu[..., b], Q=Q, R=R)
        xr, ur, (Kr, kr, Quur), vxr, vxxr, cr, info = oc.ilqg(p, x0[:, b], u0[..., b])
        st = tr["stats"][:, b]
        assert (int(st[0]), int(st[1])) == (info["status"], info["iter"])
        assert relerr(x[..., b], xr) < RTOL and relerr(uu[..., b], ur) < RTOL and relerr(Vxx[..., b], vxxr) < RTOL
        assert abs(cost[:, b].sum() - cr.sum()) < 1e-9 * abs(cr.sum())


def test_inverted_bounds_take_the_cholesky_branch(ddp):
    """lims[1,1] > lims[1,2] means "no limits" upstream (backward_pass.jl:31): the n = 64 / m = 8 launcher then runs the instantiation
    without the box-QP — the same bits as `lims = []`, not a QP with infinite bounds"""
    rng = np.random.default_rng(17)
    n, m, N, B = 64, 8, 10, 3
    cx, cu, cxx, cxu, cuu, fx, fu, u = _problem(rng, n, m, N, B, False)
    L = np.stack([0.4 * np.ones(m), -0.25 * np.ones(m)], 1)               # lower > upper
    lam = np.array([1e-3, 0.1, 2.0])
    a = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, 1, L, None, u)
    b = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, 1, None, None, u)
    assert np.array_equal(a[0], b[0])
    for got, ref in ((a[1].K, b[1].K), (a[1].k, b[1].k), (a[1].Σi, b[1].Σi), (a[2], b[2]), (a[3], b[3]), (a[4], b[4])):
        assert np.array_equal(got, ref)
