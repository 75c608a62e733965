"""diff_fun (src/forward_pass.jl:5,19; iLQG.jl:156,268; iLQGkl.jl:35,134): what stands in for a user closure on the device is
subtraction with named coordinates wrapped to [-π, π] (ddp_problem::diff_wrap, `WrappedDiff`).  HIP path vs the oracle called with the
same function, and the property that defines the hook: shifting the nominal trajectory by whole turns changes nothing."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8                     # the parity tolerance of the north star (per time step, conftest.relerr)


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    import ddp_amd.kl  # noqa: F401
    ddp_amd.default_handle()
    return ddp_amd


def _pend_case(rng, N, B):
    x0 = np.stack([3.0 + 0.1 * rng.standard_normal(B), 0.3 * rng.standard_normal(B), np.zeros(B), np.zeros(B)])
    x = np.zeros((4, N, B)); x[0] = (-3.1 + 0.01 * np.arange(N))[:, None] + 0.05 * rng.standard_normal((N, B))
    x[1:] = 0.05 * rng.standard_normal((3, N, B))
    u = 0.1 * rng.standard_normal((1, N, B)); K = 0.5 * rng.standard_normal((1, 4, N, B)); k = 0.05 * rng.standard_normal((1, N, B))
    return x0, x, u, K, k


@pytest.mark.parametrize("coords", [(0,), (0, 2)])
def test_forward_pass_wrapped_diff_pendcart_vs_oracle(ddp, coords):
    """rollouts whose state sits across the ±π cut from the nominal trajectory: every (trajectory, α) against the oracle with the same diff"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(5)
    N, B = 80, 7
    x0, x, u, K, k = _pend_case(rng, N, B)
    al = np.array([1.0, 0.5, 0.1])
    lims = np.array([[-5.0, 5.0]])
    d = ddp.WrappedDiff(*coords)
    xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, 4, 1, K, k), x0, u, x, al, ddp.PendcartProblem(), lims, d)
    plain = ddp.forward_pass(ddp.GaussianPolicy(N, 4, 1, K, k), x0, u, x, al, ddp.PendcartProblem(), lims)[0]
    assert relerr(xn, plain) > 1e-3
    P = npr.PENDCART
    p = oc.make_problem("pendcart", 4, 1, N, Q=P["Q"], R=P["R"], pend=P, diff_wrap=d.mask)
    for b in range(B):
        for j, a in enumerate(al):
            xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], x[..., b], float(a), lims)
            assert relerr(xn[..., b, j], xr) < RTOL and relerr(un[..., b, j], ur) < RTOL and relerr(cn[..., b, j], cr) < RTOL


def test_wrapped_diff_ignores_whole_turns_of_the_nominal_trajectory(ddp):
    """LQ, n = 10, m = 2 (the shape with the specialised rollout kernels: a mask must take the run-time-sized kernel): adding 2πk to the wrapped
    coordinates of the nominal trajectory changes nothing, adding it to an unwrapped one does; the mask 0 is the plain difference"""
    from oracle import np_restatement as npr
    rng = np.random.default_rng(8)
    N, B = 120, 33
    P = npr.make_lq_problem(rng, T=N)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = rng.standard_normal((10, B)); x = 0.3 * rng.standard_normal((10, N, B)); u = 0.1 * rng.standard_normal((2, N, B))
    K = 0.2 * rng.standard_normal((2, 10, N, B)); k = 0.05 * rng.standard_normal((2, N, B))
    pol = ddp.GaussianPolicy(N, 10, 2, K, k)
    d = ddp.WrappedDiff(1, 7)
    base = ddp.forward_pass(pol, x0, u, x, [1.0, 0.3], prob, None, d)
    xs = x.copy(); xs[1] += 2 * np.pi * rng.integers(-3, 4, (N, B)); xs[7] -= 4 * np.pi
    shifted = ddp.forward_pass(pol, x0, u, xs, [1.0, 0.3], prob, None, d)
    for a, b_ in zip(base, shifted):
        assert relerr(a, b_) < 1e-12
    xs2 = x.copy(); xs2[2] += 2 * np.pi
    assert relerr(ddp.forward_pass(pol, x0, u, xs2, [1.0, 0.3], prob, None, d)[0], base[0]) > 1e-3
    # |x̂ - x| < π here, so wrapping is the identity up to rounding: the wrapped rollout is the plain one
    plain = ddp.forward_pass(pol, x0, u, x, [1.0, 0.3], prob, None)
    small = np.max(np.abs(plain[0] - x[..., None])) < np.pi
    if small:
        for a, b_ in zip(base, plain):
            assert relerr(a, b_) < 1e-12
    assert all(np.array_equal(a, b_) for a, b_ in zip(ddp.forward_pass(pol, x0, u, x, [1.0, 0.3], prob, None, np.subtract), plain))


def test_ilqg_with_wrapped_diff_vs_oracle(ddp):
    """iLQG(...; diff_fun) (iLQG.jl:156,268): pendulum solves whose line searches see the wrapped difference, against the oracle's loop"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(13)
    T, B = 100, 5
    P = npr.PENDCART
    x0 = np.stack([np.pi - 0.6 + 0.2 * rng.standard_normal(B), np.zeros(B), np.zeros(B), np.zeros(B)])
    u0 = 0.2 * rng.standard_normal((1, T, B))
    lims = 5.0 * np.array([[-1.0, 1.0]])
    kw = dict(regType=2, α=10.0 ** np.linspace(0.2, -3, 6), λmax=1e15, tol_fun=1e-6, tol_grad=1e-6, max_iter=60)
    d = ddp.WrappedDiff(0)
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(ddp.PendcartProblem(), x0, u0, lims=lims, diff_fun=d, **kw)
    p = oc.make_problem("pendcart", 4, 1, T, Q=P["Q"], R=P["R"], pend=P, diff_wrap=d.mask)
    for b in range(B):
        xr, ur, (Kr, kr, Quur), vxr, vxxr, cr, info = oc.ilqg(p, x0[:, b], u0[..., b], lims=lims, regType=2, alpha=kw["α"], lam_max=1e15, tol_fun=1e-6,
                                                              tol_grad=1e-6, max_iter=60)
        st = tr["stats"][:, b]
        # (a solve that ends on `Δcost < tol_fun` with Δcost within rounding of the tolerance may take one iteration more or less)
        assert int(st[0]) == info["status"] and abs(int(st[1]) - info["iter"]) <= 1, (b, st[:2], info["status"], info["iter"])
        assert relerr(x[..., b], xr) < 1e-6 and relerr(u[..., b], ur) < 1e-5 and abs(cost[:, b].sum() - cr.sum()) < 1e-8 * cr.sum()


def test_diff_fun_refusals(ddp):
    """a closure cannot run on the device; coordinates beyond the state, masks on shapes without a kernel for them fail loudly"""
    prob = ddp.PendcartProblem()
    N = 10
    pol = ddp.GaussianPolicy(N, 4, 1, np.zeros((1, 4, N)), np.zeros((1, N)))
    args = (pol, np.zeros(4), np.zeros((1, N)), np.zeros((4, N)), 1.0, prob, None)
    with pytest.raises(TypeError):
        ddp.forward_pass(*args, lambda a, b: a - b)
    with pytest.raises(ValueError):
        ddp.forward_pass(*args, ddp.WrappedDiff(4))
    from ddp_amd import _lib
    import ctypes as C
    # the raw struct with a bit at n: the library refuses it (a caller that did not zero the struct)
    dp = ddp._DevProblem(prob, N, 1)
    dp.struct.diff_wrap = 1 << 4
    xn = np.zeros((4, N)); un = np.zeros((1, N)); cn = np.zeros(N + 1); cs = np.zeros(1)
    one = np.ones(1)
    rc = _lib.lib().ddp_forward_pass_f64(ddp.default_handle().raw, C.byref(dp.struct), None, None, _lib.ptr(np.zeros(4)), _lib.ptr(np.zeros((1, N))), None,
                                        _lib.ptr(one), 1, None, _lib.ptr(xn), _lib.ptr(un), _lib.ptr(cn), _lib.ptr(cs))
    assert rc != 0 and "diff_wrap" in _lib.lib().ddp_last_error().decode()
    # n > 32: no kernel implements the hook there
    A = np.eye(40); Bm = np.zeros((40, 2)); Bm[0, 0] = Bm[1, 1] = 1.0
    big = ddp.LQProblem(A, Bm, np.eye(40), np.eye(2))
    polb = ddp.GaussianPolicy(N, 40, 2, np.zeros((2, 40, N)), np.zeros((2, N)))
    with pytest.raises(ddp.DDPError):
        ddp.forward_pass(polb, np.zeros(40), np.zeros((2, N)), np.zeros((40, N)), 1.0, big, None, ddp.WrappedDiff(3))


@pytest.mark.parametrize("hostloop", ["0", "1"])
def test_ilqgkl_with_wrapped_diff_vs_oracle(ddp, monkeypatch, hostloop):
    """iLQGkl(...; diff_fun) (iLQGkl.jl:35,134): the mask travels with ddp_problem through the library's driver (`ddp_ilqgkl_f64`) and through
    the host-array loop; status, iterations and trajectories against the oracle's loop run with the same diff (here |x̂ - x| < π, so this
    checks the plumbing and the kernel switch, the wrapping itself is exercised by the rollout tests above)"""
    from oracle import oracle_ctypes as oc
    kl = ddp.kl
    monkeypatch.setenv("DDP_KL_HOSTLOOP", hostloop)
    rng = np.random.default_rng(31)
    N, B = 60, 3
    prob = ddp.PendcartProblem()
    lims = np.array([[-5.0, 5.0]])
    u = (1.5 * np.sin(np.arange(N) / 9.0))[None, :, None] * np.array([1.0, 0.7, 1.3]) + 0.05 * rng.standard_normal((1, N, B))
    x0 = np.array([np.pi - 0.6, 0, 0, 0])[:, None] + 0.05 * rng.standard_normal((4, B))
    x, _, c0 = ddp.forward_pass(None, x0, u, None, 1.0, prob, lims)
    cost0 = c0.sum(axis=0)
    fx, fu = ddp.df(prob, x, u)[:2]
    R1 = 1e-3 * np.eye(4)
    eye = np.ones((1, 1, N, B))
    prev = ddp.GaussianPolicy(N, 4, 1, 0.3 * rng.standard_normal((1, 4, N, B)), u, eye, eye.copy())
    d = ddp.WrappedDiff(0)
    xo, uo, pol, Vx, Vxx, cost, tr = kl.iLQGkl(prob, x, prev, kl.Model(fx, fu, R1), kl_step=0.05, lims=lims, max_iter=20, cost=cost0, diff_fun=d)
    pend = dict(g=prob.g, l=prob.l, h=prob.h, d=prob.d, goal=prob.goal)
    p = oc.make_problem("pendcart", 4, 1, N, Q=prob.Q, R=prob.R, pend=pend, diff_wrap=d.mask)
    for b in range(B):
        pb = dict(K=prev.K[..., b], k=u[..., b], S=eye[..., b], Si=eye[..., b])
        xr, ur, polr, vx, vxx, cr, info = oc.ilqgkl(p, x[..., b], float(cost0[b]), pb, dict(fx=fx[..., b], R1=R1), kl_step=0.05, lims=lims,
                                                    max_iter=20)
        assert (tr["status"][b], tr["iter"][b], tr["n_backpass"][b]) == (info["status"], info["iter"], info["n_backpass"]), b
        assert relerr(xo[..., b], xr) < 1e-7 and relerr(uo[..., b], ur) < 1e-7 and relerr(pol.K[..., b], polr["K"]) < 1e-7


@pytest.mark.parametrize("lane", ["0", "1"])
def test_wrapped_diff_in_the_pendulum_kernels(ddp, monkeypatch, lane):
    """the pendulum's own rollout kernels — the 16-lane row kernel (DDP_FORWARD_LANE=0) and the lane-per-rollout kernel (=1) — with a wrapped
    angle (src/forward_pass.jl:19): against the oracle with the same diff, with and without control limits, and against the
    run-time-sized kernel (DDP_FORWARD=group)"""
    from ddp_amd import _lib
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_FORWARD_LANE", lane)
    rng = np.random.default_rng(15)
    N, B = 96, 9
    x0, x, u, K, k = _pend_case(rng, N, B)
    al = np.array([1.0, 0.4, 0.05])
    d = ddp.WrappedDiff(0)
    P = npr.PENDCART
    for lims in (None, np.array([[-5.0, 5.0]])):
        pol = ddp.GaussianPolicy(N, 4, 1, K, k)
        xn, un, cn = ddp.forward_pass(pol, x0, u, x, al, ddp.PendcartProblem(), lims, d)
        assert _lib.default_handle().last_kernel(1) == "forward_dpp_kernel"            # the launcher of the pendulum kernels
        p = oc.make_problem("pendcart", 4, 1, N, Q=P["Q"], R=P["R"], pend=P, diff_wrap=d.mask)
        for b in range(B):
            for j, a in enumerate(al):
                xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], x[..., b], float(a), lims)
                assert relerr(xn[..., b, j], xr) < RTOL and relerr(un[..., b, j], ur) < RTOL and relerr(cn[..., b, j], cr) < RTOL
        monkeypatch.setenv("DDP_FORWARD", "group")
        xg, ug, cg = ddp.forward_pass(pol, x0, u, x, al, ddp.PendcartProblem(), lims, d)
        monkeypatch.delenv("DDP_FORWARD")
        assert relerr(xn, xg) < 1e-11 and relerr(un, ug) < 1e-11 and relerr(cn, cg) < 1e-10


def test_cost_diag_with_a_full_Q_is_rejected(ddp):
    """ddp_problem::cost_diag = 1 is a declaration (the fused rollout cost reads only diag(Q), diag(R)): the C side verifies it — on the
    host copies in the host-pointer flavours, by a cached device-to-host look in the `_dev` flavours — instead of returning the cost of
    diag(Q) silently"""
    import ctypes as C
    from ddp_amd import _lib
    from oracle import np_restatement as npr
    rng = np.random.default_rng(1)
    N, B = 40, 5
    P = npr.make_lq_problem(rng, T=N)
    Qfull = P["Q"] + 1e-3 * np.ones((10, 10))
    prob = ddp.LQProblem(P["A"], P["B"], Qfull, P["R"])
    x0 = rng.standard_normal((10, B)); u = 0.1 * rng.standard_normal((2, N, B))
    L = _lib.lib(); h = _lib.default_handle()
    # the host mirror sets cost_diag from the matrices: a full Q takes the general cost kernel and is right
    xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(), x0, u, None, 1.0, prob, None)
    ref = 0.5 * np.einsum("itb,ij,jtb->tb", xn, Qfull, xn) + 0.5 * np.einsum("itb,ij,jtb->tb", un, P["R"], un)
    assert relerr(cn, ref) < 1e-10
    # a caller of the C ABI that sets the flag although Q is full: refused, host-pointer and device-pointer flavour alike
    dp = ddp._DevProblem(prob, N, B)
    dp.struct.cost_diag = 1
    one = np.array([1.0])
    xo = np.zeros((10, N, B), order="F"); uo = np.zeros((2, N, B), order="F"); co = np.zeros((N, B), order="F"); cs = np.zeros(B)
    rc = L.ddp_forward_pass_f64(h.raw, C.byref(dp.struct), None, None, _lib.ptr(x0), _lib.ptr(np.asfortranarray(u)), None, _lib.ptr(one), 1, None,
                                *map(_lib.ptr, (xo, uo, co, cs)))
    assert rc < 0 and b"off-diagonal" in L.ddp_last_error()
    dQ, dR, dA, dB = (h.to_device(a) for a in (Qfull, P["R"], P["A"], P["B"]))
    dx0, du = h.to_device(x0), h.to_device(u)
    outs = [h.malloc(8 * s) for s in (10 * N * B, 2 * N * B, N * B, B)]
    dp.struct.Q, dp.struct.R, dp.struct.A, dp.struct.Bm = dQ.value, dR.value, dA.value, dB.value
    for _ in range(2):                                            # the second call answers from the handle's cache
        rc = L.ddp_forward_pass_f64_dev(h.raw, C.byref(dp.struct), None, None, dx0, du, None, _lib.ptr(one), 1, None, None, *outs)
        assert rc < 0 and b"off-diagonal" in L.ddp_last_error()
    dp.struct.cost_diag = 0
    assert L.ddp_forward_pass_f64_dev(h.raw, C.byref(dp.struct), None, None, dx0, du, None, _lib.ptr(one), 1, None, None, *outs) == 0
    h.sync()
    for p_ in [dQ, dR, dA, dB, dx0, du] + outs:
        h.free(p_)
