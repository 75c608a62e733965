"""GPU edge cases of the hot path (the reference has no unit tests; these follow its semantics line by line):
degenerate horizons, ragged batches, the no-limits quirk, infinite bounds, NaN controls, divergence bookkeeping,
the maximum number of line-search candidates."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    ddp_amd.default_handle()
    return ddp_amd


def _lq(rng, n, m, N, B):
    from oracle import np_restatement as npr
    P = npr.make_lq_problem(rng, n=n, m=m, T=N)
    x = rng.standard_normal((n, N, B)); u = 0.3 * rng.standard_normal((m, N, B))
    cx = np.einsum("ij,jtb->itb", P["Q"], x); cu = np.einsum("ij,jtb->itb", P["R"], u)
    return P, x, u, cx, cu


@pytest.mark.parametrize("impl", ["general", "dpp", "x"])
@pytest.mark.parametrize("N", [1, 2, 3, 9])
def test_back_pass_short_horizons(ddp, monkeypatch, impl, N):
    """N=1: only the terminal assignments (backward_pass.jl:234-236); N=2: a single Riccati step"""
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_BACKPASS", impl)
    rng = np.random.default_rng(N)
    P, x, u, cx, cu = _lq(rng, 10, 2, N, 3)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, P["Q"], np.zeros((10, 2)), P["R"], P["A"], P["B"], 0.5, 1, None, x, u)
    for b in range(3):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], P["Q"], np.zeros((10, 2)), P["R"], P["A"], P["B"], 0.5, 1,
                                                  None, x[..., b], u[..., b])
        assert div[b] == d == 0
        assert np.array_equal(Vxx[:, :, N - 1, b], P["Q"]) and np.array_equal(Vx[:, N - 1, b], cx[:, N - 1, b])
        assert not pol.K[:, :, N - 1, b].any() and not pol.k[:, N - 1, b].any()
        for got, ref in ((pol.K[..., b], K), (pol.k[..., b], k), (Vx[..., b], vx), (Vxx[..., b], vxx), (dV[:, b], dv)):
            assert np.max(np.abs(got - ref)) <= RTOL * max(1e-300, np.max(np.abs(ref)))


@pytest.mark.parametrize("impl", ["general", "dpp", "x"])
@pytest.mark.parametrize("B", [1, 2, 5, 7])
def test_back_pass_ragged_batches(ddp, monkeypatch, impl, B):
    """batch sizes that do not fill a wavefront's four 16-lane rows"""
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_BACKPASS", impl)
    rng = np.random.default_rng(40 + B)
    P, x, u, cx, cu = _lq(rng, 10, 2, 21, B)
    lam = 10.0 ** rng.uniform(-3, 0, B)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, P["Q"], np.zeros((10, 2)), P["R"], P["A"], P["B"], lam, 2, None, x, u)
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], P["Q"], np.zeros((10, 2)), P["R"], P["A"], P["B"], lam[b], 2,
                                                  None, x[..., b], u[..., b])
        assert div[b] == d == 0 and relerr(Vxx[..., b], vxx) < RTOL and relerr(pol.K[..., b], K) < RTOL and relerr(dV[:, b], dv) < RTOL


@pytest.mark.parametrize("impl", ["general", "dpp"])
def test_no_limits_quirk_lims11_gt_lims12(ddp, monkeypatch, impl):
    """backward_pass.jl:31: `lims[1,1] > lims[1,2]` selects the Cholesky branch even though limits were passed"""
    monkeypatch.setenv("DDP_BACKPASS", impl)
    g = load_golden("bp_lti_n10m2_reg1")
    weird = np.array([[1.0, -1.0], [-0.01, 0.01]])          # first row reversed -> "no limits"
    d0, p0, Vx0, Vxx0, dV0 = ddp.back_pass(g["cx"], g["cu"], g["cxx"], g["cxu"], g["cuu"], g["fx"], g["fu"], 1.0, 1, None, g["x"], g["u"])
    d1, p1, Vx1, Vxx1, dV1 = ddp.back_pass(g["cx"], g["cu"], g["cxx"], g["cxu"], g["cuu"], g["fx"], g["fu"], 1.0, 1, weird, g["x"], g["u"])
    assert d0 == d1 == 0
    assert relerr(p1.K, g["K"]) < RTOL and relerr(Vxx1, g["Vxx"]) < RTOL and relerr(p1.K, p0.K) < 1e-12


def test_boxqp_infinite_bounds_and_codes(ddp):
    from oracle import oracle_ctypes as oc
    H = np.array([[2.0, 0.3], [0.3, 1.0]]); g = np.array([0.3, -0.2])
    inf = np.inf * np.ones(2)
    x, res, Hf, free = ddp.boxQP(H, g, -inf, inf, np.zeros(2))
    xr, rr, *_ = oc.boxqp(H, g, -inf, inf, np.zeros(2))
    assert res == rr and res >= 1 and np.allclose(x, -np.linalg.solve(H, g)) and free.all()
    x, res, Hf, free = ddp.boxQP(np.eye(2), np.array([5.0, -5.0]), -np.ones(2), np.ones(2), np.array([-1.0, 1.0]))
    assert res == 6 and not free.any() and Hf.size == 0 and np.array_equal(x, [-1.0, 1.0])
    x, res, *_ = ddp.boxQP(np.array([[1.0, 0.0], [0.0, -1.0]]), np.ones(2), -np.ones(2), np.ones(2), np.zeros(2))
    assert res == 0                                              # swallowed PosDefException (Q10)
    x, res, *_ = ddp.boxQP(np.eye(2), np.zeros(2), -np.ones(2), np.ones(2), np.zeros(2))
    assert res == 5


@pytest.mark.parametrize("impl", ["dpp", "group"])
def test_forward_pass_nan_controls_are_zeroed(ddp, monkeypatch, impl):
    """`u[isnan.(u)] .= 0` inside f (demo_linear.jl:43) mutates unew through the view"""
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_FORWARD", impl)
    g = load_golden("fwd_lq_n10m2")
    u = g["u"].copy(); u[0, 5] = np.nan; u[1, 17] = np.nan
    prob = ddp.LQProblem(g["A"], g["B"], g["Q"], g["R"])
    xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(), g["x0"], u, None, 1.0, prob, None)
    p = oc.make_problem("lq", 10, 2, u.shape[1], A=g["A"], B=g["B"], Q=g["Q"], R=g["R"])
    xr, ur, cr = oc.forward_pass(p, None, g["x0"], u, None, 1.0, None)
    assert un[0, 5] == 0.0 and un[1, 17] == 0.0 and np.isfinite(xn).all()
    assert relerr(xn, xr) < RTOL and relerr(un, ur) < RTOL and relerr(cn, cr) < RTOL


@pytest.mark.parametrize("impl", ["dpp", "group"])
def test_forward_pass_sixteen_alphas_and_ragged_batch(ddp, monkeypatch, impl):
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_FORWARD", impl)
    g = load_golden("fwd_lq_n10m2")
    N = g["u"].shape[1]
    B = 3
    rng = np.random.default_rng(2)
    x0 = g["x0"][:, None] + 0.1 * rng.standard_normal((10, B))
    u = np.stack([g["u"]] * B, -1); x = np.stack([g["x"]] * B, -1)
    K = np.stack([g["K"]] * B, -1); k = np.stack([g["k"]] * B, -1)
    alphas = 10.0 ** np.linspace(0.2, -3, 16)
    prob = ddp.LQProblem(g["A"], g["B"], g["Q"], g["R"])
    xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, 10, 2, K, k), x0, u, x, alphas, prob, None)
    p = oc.make_problem("lq", 10, 2, N, A=g["A"], B=g["B"], Q=g["Q"], R=g["R"])
    for b in range(B):
        for j in (0, 7, 15):
            xr, ur, cr = oc.forward_pass(p, (g["K"], g["k"]), x0[:, b], g["u"], g["x"], float(alphas[j]), None)
            assert relerr(xn[..., b, j], xr) < RTOL and relerr(un[..., b, j], ur) < RTOL and relerr(cn[..., b, j], cr) < RTOL


def test_ilqg_initial_divergence_is_reported(ddp):
    """iLQG.jl:205-210: an initial control sequence that blows up for every α returns `nothing` (status -1 in a batch)"""
    n, m, N = 10, 2, 400
    A = 1.5 * np.eye(n); Bm = np.ones((n, m)); Q = np.eye(n); R = np.eye(m)
    prob = ddp.LQProblem(A, Bm, Q, R)
    x0 = 1e3 * np.ones(n); u0 = np.ones((m, N))
    assert ddp.iLQG(prob, x0, u0) is None
    x0b = np.stack([x0, 1e-3 * np.ones(n)], 1); u0b = np.stack([u0, np.zeros((m, N))], -1)
    out = ddp.iLQG(prob, x0b, u0b, max_iter=3)
    assert int(out[6]["stats"][0, 0]) == -1


def test_ilqg_lambda_exit_and_per_trajectory_status(ddp):
    """a batch mixing an easy LQ problem with one whose back pass can never succeed (negative-definite R, λmax small)"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    P = npr.make_lq_problem(np.random.default_rng(4), T=80)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], -1e3 * np.eye(2))
    out = ddp.iLQG(prob, P["x0"], P["u0"], λmax=1e2)
    p = oc.make_problem("lq", 10, 2, 80, A=P["A"], B=P["B"], Q=P["Q"], R=-1e3 * np.eye(2))
    ref = oc.ilqg(p, P["x0"], P["u0"], lam_max=1e2)
    st = out[6]["stats"][:, 0]
    assert int(st[0]) == ref[6]["status"] == 3 and int(st[3]) == ref[6]["n_backpass"] and st[5] == ref[6]["lam"]


# ------------------------------------------------------------------ pre-rolled warm start (iLQG.jl:193-197)
@pytest.mark.parametrize("kind", ["lq", "pendcart"])
@pytest.mark.parametrize("give_cost", [False, True])
def test_ilqg_prerolled_warm_start(ddp, kind, give_cost):
    """x0[n,N,B] pre-rolled: no initial rollout, cost given or costfun(x0,u0); an MPC-style shifted (dynamically
    inconsistent) trajectory is taken as it is, like the reference does"""
    from oracle import oracle_ctypes as oc, np_restatement as npr
    rng = np.random.default_rng(41)
    B = 3
    if kind == "lq":
        n, m, N = 10, 2, 60
        P = npr.make_lq_problem(rng, T=N)
        prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
        p = oc.make_problem("lq", n, m, N, A=P["A"], B=P["B"], Q=P["Q"], R=P["R"])
        x0c = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B)); u0 = 0.1 * rng.standard_normal((m, N, B)); lims = None
        kw = dict(max_iter=30)
    else:
        n, m, N = 4, 1, 70
        prob = ddp.PendcartProblem()
        p = oc.make_problem("pendcart", 4, 1, N, Q=prob.Q, R=prob.R, pend=dict(g=prob.g, l=prob.l, h=prob.h, d=prob.d, goal=prob.goal))
        x0c = np.array([np.pi - 0.6, 0, 0, 0])[:, None] + 0.05 * rng.standard_normal((4, B)); u0 = 0.5 * rng.standard_normal((1, N, B))
        lims = np.array([[-5.0, 5.0]])
        kw = dict(max_iter=15, regType=2)
    xr, ur, c0 = ddp.forward_pass(None, x0c, u0, None, 1.0, prob, lims)
    # receding-horizon shift: drop the first step, repeat the last (the result is not a rollout of u any more)
    x0 = np.concatenate([xr[:, 1:], xr[:, -1:]], axis=1); u0s = np.concatenate([ur[:, 1:], ur[:, -1:]], axis=1)
    cost0 = None
    if give_cost:
        cost0 = np.stack([oc.forward_pass(p, None, x0[:, 0, b], u0s[..., b], None, 1.0, lims)[2] for b in range(B)], -1) * 1.0 + 0.01
    opts = dict(kw)
    res = ddp.iLQG(prob, x0, u0s, lims=lims, cost=cost0, λ=kw.get("lam", 1.0), max_iter=kw["max_iter"], regType=kw.get("regType", 1))
    x, u, pol, Vx, Vxx, cost, tr = res
    for b in range(B):
        xo, uo, (K, k, Quu), vx, vxx, co, info = oc.ilqg_prerolled(p, x0[..., b], u0s[..., b], None if cost0 is None else cost0[:, b], lims, **opts)
        st = tr["stats"][:, b]
        assert (int(st[0]), int(st[1])) == (info["status"], info["iter"]), (b, st[:2], info)
        assert relerr(x[..., b], xo) < 1e-7 and relerr(u[..., b], uo) < 1e-7 and relerr(Vxx[..., b], vxx) < 1e-7
        assert abs(cost[:, b].sum() - co.sum()) < 1e-8 * abs(co.sum())


# ------------------------------------------------------------------ trace keys (iLQG.jl:257,325-330)
@pytest.mark.parametrize("kind", ["lq", "pendcart"])
def test_ilqg_trace_history_matches_oracle(ddp, kind):
    """λ, dλ, α, improvement, cost, reduce_ratio, grad_norm per iteration and trajectory, like the reference's MVHistory"""
    from oracle import oracle_ctypes as oc, np_restatement as npr
    rng = np.random.default_rng(51)
    B = 3
    if kind == "lq":
        n, m, N = 10, 2, 50
        P = npr.make_lq_problem(rng, T=N)
        prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
        p = oc.make_problem("lq", n, m, N, A=P["A"], B=P["B"], Q=P["Q"], R=P["R"])
        x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B)); u0 = 0.1 * rng.standard_normal((m, N, B)); lims = None
        kw = dict(max_iter=25)
    else:
        n, m, N = 4, 1, 60
        prob = ddp.PendcartProblem()
        p = oc.make_problem("pendcart", 4, 1, N, Q=prob.Q, R=prob.R, pend=dict(g=prob.g, l=prob.l, h=prob.h, d=prob.d, goal=prob.goal))
        x0 = np.array([np.pi - 0.6, 0, 0, 0])[:, None] + 0.05 * rng.standard_normal((4, B)); u0 = 0.3 * rng.standard_normal((1, N, B))
        lims = np.array([[-5.0, 5.0]])
        kw = dict(max_iter=12, regType=2)
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(prob, x0, u0, lims=lims, max_iter=kw["max_iter"], regType=kw.get("regType", 1))
    for b in range(B):
        info = oc.ilqg_trace7(p, x0[:, b], u0[..., b], lims, **kw)[6]
        tl = info["trace_len"]
        assert tl >= 3
        for key, ref in info["history"].items():
            got = tr["history"][key][:tl, b]
            both_nan = np.isnan(got) & np.isnan(ref)
            scale = max(1e-300, np.nanmax(np.abs(ref))) if np.isfinite(ref).any() else 1.0
            assert np.all(both_nan | (np.abs(got - ref) <= 1e-7 * scale + 1e-12)), (key, b, got, ref)


def test_ilqg_timing_trace_keys(ddp, capsys):
    """time_derivs / time_backward / time_forward (iLQG.jl:227,241,281) and print_timing (:343-366)"""
    rng = np.random.default_rng(5)
    n, m, N, B = 10, 2, 60, 4
    a0 = rng.standard_normal((n, n)); A = np.eye(n) + 0.01 * (a0 - a0.T); Bm = 0.01 * rng.standard_normal((n, m))
    prob = ddp.LQProblem(A, Bm, 0.01 * np.eye(n), 0.001 * np.eye(m))
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(prob, np.ones((n, B)), 0.1 * rng.standard_normal((m, N, B)), verbosity=1)
    g = tr["global_iters"]
    assert g >= 1
    for key in ("time_derivs", "time_backward", "time_forward"):
        t = tr[key]
        assert t.shape == (g,) and np.all(np.isfinite(t)) and np.all(t > 0.0)
    assert sum(tr[k].sum() for k in ("time_derivs", "time_backward", "time_forward")) <= tr["time_total"]
    out = capsys.readouterr().out
    assert "back pass:" in out and "fwd pass:" in out and "derivs:" in out


def test_mpc_shift_and_receding_horizon_loop(ddp):
    """ddp_mpc_shift_f64_dev against NumPy, then three receding-horizon solves warm-started from the shifted solution"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(21)
    a = rng.standard_normal((3, 2, 7, 4))                          # K-like [m,n,N,B]
    for sh, zt in ((1, False), (2, True), (9, False), (0, False)):
        out = ddp.mpc_shift(a, sh, zero_tail=zt)
        ref = np.empty_like(a)
        for i in range(7):
            ref[:, :, i] = a[:, :, i + sh] if i + sh < 7 else (0.0 if zt else a[:, :, 6])
        assert np.array_equal(out, ref)
    assert np.array_equal(ddp.mpc_shift(a[..., 0], 1, batched=False), np.concatenate([a[:, :, 1:, 0], a[:, :, -1:, 0]], axis=2))
    # receding horizon on the LQ family: solve, apply u[:,0], shift, re-solve from the measured state; every solve is checked
    n, m, N = 10, 2, 40
    a0 = rng.standard_normal((n, n)); A = np.eye(n) + 0.01 * (a0 - a0.T); Bm = 0.01 * rng.standard_normal((n, m))
    Q, R = 0.01 * np.eye(n), 0.001 * np.eye(m)
    prob = ddp.LQProblem(A, Bm, Q, R)
    p = oc.make_problem("lq", n, m, N, A=A, B=Bm, Q=Q, R=R)
    xm = np.ones(n); u = 0.1 * rng.standard_normal((m, N))
    for step in range(3):
        x, us, pol, Vx, Vxx, cost, tr = ddp.iLQG(prob, xm, u)
        xr, ur, polr, vxr, vxxr, cr, info = oc.ilqg(p, xm, u)
        assert relerr(us, ur) < RTOL and relerr(x, xr) < RTOL and int(tr["iter"][0]) == info["iter"]
        xm = A @ xm + Bm @ us[:, 0] + 1e-3 * rng.standard_normal(n)      # the plant moved on (with a disturbance)
        u = ddp.mpc_shift(us, 1)
    assert np.array_equal(u[:, :-1], us[:, 1:]) and np.array_equal(u[:, -1], us[:, -1])


def test_demo_entry_points(ddp):
    """demo_linear / demo_pendcart (src/demo_linear.jl:5-60, src/system_pendcart.jl:42-212): the reference's own smoke tests
    (test/runtests.jl:8-12) plus the finite-horizon LQR optimum for the linear demo"""
    rng = np.random.default_rng(42)
    x, u, pol, Vx, Vxx, cost, tr = ddp.demo_linear(rng=rng, T=200)
    assert int(tr["status"][0]) in (1, 2) and np.isfinite(cost).all()
    c0 = tr["cost"][0] if len(tr["cost"]) else cost.sum()
    assert cost.sum() <= c0 + 1e-12                                   # iLQG never accepts an increase
    xb, ub, polb, *_rest, costb, trb = ddp.demo_linear(rng=np.random.default_rng(1), T=120, B=3)
    assert xb.shape == (10, 120, 3) and np.isfinite(costb).all()
    r = ddp.demo_pendcart(T=150, max_iter=20)
    assert r is not None and r[0].shape == (4, 150) and np.abs(r[1]).max() <= 5.0 + 1e-12      # control limits respected


@pytest.mark.parametrize("impl", ["auto", "dpp", "general"])
@pytest.mark.parametrize("n,m", [(4, 1), (10, 2)])
def test_long_unstable_horizon_error_does_not_grow(ddp, monkeypatch, impl, n, m):
    """The value recursion must carry the SYMMETRISED Vxx (backward_pass.jl:71-72): with the column as computed the antisymmetric
    rounding residue grows like ρ(A)^(2·steps) — the row-per-trajectory kernel was at 5e-8 after 210 steps of this problem
    (found by tests/fuzz_gpu_parity.py); every kernel stays at the 1e-12 level now."""
    from oracle import oracle_ctypes as oc
    if impl != "auto":
        monkeypatch.setenv("DDP_BACKPASS", impl)
    rng = np.random.default_rng(333)
    N, B, h = 240, 6, 0.05
    fx = np.eye(n) + h * rng.standard_normal((n, n)) / np.sqrt(n)             # spectral radius > 1
    fu = h * rng.standard_normal((n, m))
    a = rng.standard_normal((n, n)); cxx = h * (a @ a.T / n + 0.5 * np.eye(n))
    a = rng.standard_normal((m, m)); cuu = 0.1 * h * (a @ a.T / m + 0.5 * np.eye(m))
    cxu = 0.01 * h * rng.standard_normal((n, m))
    cx = h * rng.standard_normal((n, N, B)); cu = 0.1 * h * rng.standard_normal((m, N, B))
    lam = 10.0 ** rng.uniform(-3, 0, B)
    for regType in (1, 2):
        div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, None, None, np.zeros((m, N, B)))
        for b in range(B):
            d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx, cxu, cuu, fx, fu, lam[b], regType, None, None, np.zeros((m, N)))
            assert div[b] == d == 0
            assert relerr(pol.K[..., b], K) < 1e-10 and relerr(Vxx[..., b], vxx) < 1e-10 and relerr(Vx[..., b], vx) < 1e-10


@pytest.mark.parametrize("lane", ["0", "1"])                       # 16-lane row per rollout | one lane per rollout (line searches)
@pytest.mark.parametrize("lims", [False, True])
def test_pendcart_rollout_kernels(ddp, monkeypatch, lane, lims):
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_FORWARD_LANE", lane)
    rng = np.random.default_rng(77)
    N, B = 67, 70                                                   # partial last wave in both kernels
    prob = ddp.PendcartProblem()
    L = np.array([[-2.0, 2.5]]) if lims else None
    x0 = np.array([np.pi - 0.6, 0, 0, 0])[:, None] + 0.2 * rng.standard_normal((4, B))
    u = 1.5 * rng.standard_normal((1, N, B))
    xnom = np.array([np.pi, 0, 0, 0])[:, None, None] + 0.3 * rng.standard_normal((4, N, B))
    K = 0.3 * rng.standard_normal((1, 4, N, B)); k = 0.2 * rng.standard_normal((1, N, B))
    alphas = np.array([1.0, 0.5, 0.1])
    pend = dict(g=prob.g, l=prob.l, h=prob.h, d=prob.d, goal=prob.goal)
    p = oc.make_problem("pendcart", 4, 1, N, Q=prob.Q, R=prob.R, pend=pend)
    xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, 4, 1, K, k), x0, u, xnom, alphas, prob, L)
    x1, u1, c1 = ddp.forward_pass(None, x0, u, None, 1.0, prob, L)                 # open loop (initial rollout)
    for b in range(0, B, 7):
        for j, a in enumerate(alphas):
            xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], xnom[..., b], float(a), L)
            assert relerr(xn[..., b, j], xr) < RTOL and relerr(un[..., b, j], ur) < RTOL and relerr(cn[..., b, j], cr) < RTOL
        xr, ur, cr = oc.forward_pass(p, None, x0[:, b], u[..., b], None, 1.0, L)
        assert relerr(x1.reshape(4, N, B)[..., b], xr) < RTOL and relerr(c1.reshape(-1, B)[:, b], cr) < RTOL


@pytest.mark.parametrize("family", ["pendcart", "lq"])
def test_ilqg_compaction_and_line_search_groups_change_nothing(ddp, monkeypatch, family):
    """The driver drops finished trajectories from the working set (compaction) and rolls the line search out in groups of step
    sizes; both are scheduling only: status, iteration counts, n_backpass / n_forward, trace and every array equal the plain
    lock-step run (DDP_ILQG_COMPACT=0, DDP_ILQG_LSGROUPS=0).  DDP_ILQG_COMPACT=4 lets batches of 4+ slots compact."""
    rng = np.random.default_rng(17)
    if family == "pendcart":
        B, T = 40, 90
        prob = ddp.PendcartProblem()
        x0 = np.tile(np.array([np.pi - 0.6, 0, 0, 0])[:, None], (1, B)); x0[0] += rng.uniform(-0.3, 0.3, B)
        u0 = np.zeros((1, T, B))
        kw = dict(lims=5.0 * np.array([[-1.0, 1.0]]), regType=2, α=10.0 ** np.linspace(0.2, -3, 6), λmax=1e15, tol_fun=1e-8, tol_grad=1e-8,
                  max_iter=300)
    else:
        from oracle import np_restatement as npr
        B, T = 30, 120
        P = npr.make_lq_problem(rng, T=T)
        prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
        x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B))
        u0 = 0.1 * rng.standard_normal((2, T, B)) * (1 + 3 * np.arange(B))[None, None, :]
        kw = dict(tol_fun=10.0 ** rng.uniform(-9, -3))                      # spread of iteration counts comes from u0
    runs = {}
    for tag, compact, groups in (("plain", "0", "0"), ("compact", "4", "0"), ("groups", "0", "1"), ("both", "4", "1"),     # forced on
                                 ("sparse_poll", "0", "0"), ("sparse_poll_compact", "4", "0")):
        monkeypatch.setenv("DDP_ILQG_COMPACT", compact); monkeypatch.setenv("DDP_ILQG_LSGROUPS", groups)
        # without the time_* keys the driver looks at the running count every 4th batch iteration only
        runs[tag] = ddp.iLQG(prob, x0, u0, timing=not tag.startswith("sparse_poll"), **kw)
    ref = runs["plain"]
    its = ref[6]["stats"][1]
    assert its.max() >= 2 * np.median(its) or family == "lq"               # stragglers: compaction really happens
    for tag in ("compact", "groups", "both", "sparse_poll", "sparse_poll_compact"):
        r = runs[tag]
        assert r[6]["global_iters"] == ref[6]["global_iters"], tag
        assert np.array_equal(r[6]["stats"][:5], ref[6]["stats"][:5]), tag  # status, iter, accepted_iter, n_backpass, n_forward
        for a, b_ in zip(r[:2] + (r[2].K, r[2].k, r[2].Σi) + r[3:6], ref[:2] + (ref[2].K, ref[2].k, ref[2].Σi) + ref[3:6]):
            assert np.array_equal(a, b_), tag
        assert np.array_equal(np.nan_to_num(r[6]["history"]["cost"]), np.nan_to_num(ref[6]["history"]["cost"])), tag


def test_capi_comm_single_rank(ddp):
    """ddp_comm_* / ddp_allreduce_stats_f64_dev (RCCL owned by the C ABI) with a communicator of one rank: the vector comes back
    unchanged (SUM and MAX over one rank), asynchronously on the handle's stream.  More ranks need more GPUs: the gloo test covers
    the sharding logic, the driver's multi-GPU bench (`bench.py --collective capi`) the real thing."""
    from ddp_amd import sharding
    h = ddp.default_handle()
    comm = sharding.CApiComm(h, 0, 1, exchange=lambda b: b)
    v = np.array([3.5, -1.25, 7.0, 2.0, 9.0])
    d = h.to_device(v)
    try:
        comm.allreduce(d, 3, 2)
        comm.allreduce(d, 5, 0)
        assert np.array_equal(h.to_host(d, (5,)), v)
        with pytest.raises(ddp.DDPError):
            comm.allreduce(d, 60, 10)                       # more than DDP_COMM_MAX_STATS entries
    finally:
        h.free(d); comm.close()


@pytest.mark.parametrize("family,lane", [("lq", None), ("pendcart", "0"), ("pendcart", "1")])
@pytest.mark.parametrize("N", [1, 2, 7, 8, 15, 16, 17, 31, 32, 33, 100])
def test_fused_cost_equals_cost_kernel(ddp, monkeypatch, family, lane, N):
    """ddp_problem::cost_diag: the rollout kernels evaluate the cost themselves (16-step LDS tile in the row kernel, in the lane for
    the one-lane-per-rollout pendcart kernel); every horizon remainder against the separate cost kernel and the oracle"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(N)
    B = 9
    if lane is not None:
        monkeypatch.setenv("DDP_FORWARD_LANE", lane)
    if family == "lq":
        from oracle import np_restatement as npr
        P = npr.make_lq_problem(rng, T=N)
        Q, R = np.diag(rng.uniform(0.5, 2.0, 10)), np.diag(rng.uniform(0.1, 1.0, 2))
        prob = ddp.LQProblem(P["A"], P["B"], Q, R)
        n, m, lims = 10, 2, np.array([[-0.3, 0.4], [-0.2, 0.25]])
        p = oc.make_problem("lq", n, m, N, A=P["A"], B=P["B"], Q=Q, R=R)
    else:
        prob = ddp.PendcartProblem(Q=np.diag(rng.uniform(0.5, 10.0, 4)), R=np.array([[0.7]]))
        n, m, lims = 4, 1, np.array([[-5.0, 5.0]])
        p = oc.make_problem("pendcart", 4, 1, N, Q=prob.Q, R=prob.R, pend=dict(g=prob.g, l=prob.l, h=prob.h, d=prob.d, goal=prob.goal))
    x0 = rng.standard_normal((n, B)) + (np.array([np.pi, 0, 0, 0])[:, None] if family == "pendcart" else 0)
    u = 0.5 * rng.standard_normal((m, N, B)); x = rng.standard_normal((n, N, B))
    K = 0.1 * rng.standard_normal((m, n, N, B)); k = 0.1 * rng.standard_normal((m, N, B))
    al = np.array([1.0, 0.3, 0.05])
    pol = ddp.GaussianPolicy(N, n, m, K, k)
    outs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("DDP_FORWARD_FUSE", fuse)
        outs[fuse] = ddp.forward_pass(pol, x0, u, x, al, prob, lims)
    for a_, b_ in zip(outs["1"], outs["0"]):
        assert a_.shape == b_.shape and relerr(a_, b_) < 1e-13
    xn, un, cn = outs["1"]
    for b in (0, B - 1):
        for j, a in enumerate(al):
            xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], x[..., b], float(a), lims)
            assert relerr(cn[:, b, j], cr) < 1e-12 and relerr(xn[..., b, j], xr) < RTOL


def test_results_are_reproducible_bit_for_bit(ddp):
    """No atomics, no order-dependent reductions on the path: the same call twice gives the same bits — backward pass (matrix-core
    kernels of both families), rollout with the fused cost, and whole batched solves with their per-trajectory bookkeeping."""
    rng = np.random.default_rng(77)
    # n = 10, m = 2 (back_pass_mx + forward_dpp) and n = 4, m = 1 with limits (back_pass_q4l + pendcart rollout)
    B, N = 9, 64
    for (n, m, lims) in ((10, 2, None), (4, 1, np.array([[-0.3, 0.3]]))):
        h = 0.05
        fx = np.eye(n)[:, :, None, None] + h * rng.standard_normal((n, n, N, B)) / np.sqrt(n)
        fu = h * rng.standard_normal((n, m, N, B))
        a = rng.standard_normal((n, n)); cxx = h * (a @ a.T / n + 0.5 * np.eye(n))
        cuu = 0.1 * h * np.eye(m); cxu = np.zeros((n, m))
        cx = h * rng.standard_normal((n, N, B)); cu = 0.1 * h * rng.standard_normal((m, N, B)); u = 0.2 * rng.standard_normal((m, N, B))
        outs = [ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.3, 1, lims, None, u) for _ in range(2)]
        (d0, p0, vx0, vxx0, dv0), (d1, p1, vx1, vxx1, dv1) = outs
        for x_, y_ in ((d0, d1), (p0.K, p1.K), (p0.k, p1.k), (p0.Σi, p1.Σi), (vx0, vx1), (vxx0, vxx1), (dv0, dv1)):
            assert np.array_equal(x_, y_, equal_nan=True)
    # whole solves
    import scipy.linalg as sla
    n, m, T, B = 10, 2, 120, 7
    A0 = rng.standard_normal((n, n)); A = sla.expm(0.01 * (A0 - A0.T)); Bm = 0.01 * rng.standard_normal((n, m))
    prob = ddp.LQProblem(A, Bm, 0.01 * np.eye(n), 0.001 * np.eye(m))
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B)); u0 = 0.1 * rng.standard_normal((m, T, B))
    r0 = ddp.iLQG(prob, x0, u0, lims=0.05 * np.array([[-1.0, 1.0], [-1.0, 1.0]]))
    r1 = ddp.iLQG(prob, x0, u0, lims=0.05 * np.array([[-1.0, 1.0], [-1.0, 1.0]]))
    for k_ in (0, 1, 3, 4, 5):
        assert np.array_equal(r0[k_], r1[k_], equal_nan=True), k_
    assert np.array_equal(r0[2].K, r1[2].K) and np.array_equal(r0[6]["stats"], r1[6]["stats"])


def test_trace_cap_is_reported_and_truncation_flagged(ddp):
    """the history keeps `trace_cap` rows per trajectory (the default shrinks with the batch); a trajectory that iterates
    longer is flagged in trace['truncated'] (with a warning), and timing=False really drops the time_* keys"""
    import warnings
    from oracle import np_restatement as npr
    rng = np.random.default_rng(4)
    P = npr.make_lq_problem(rng, T=60)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    B = 3
    x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B))
    u0 = 0.1 * rng.standard_normal((2, 60, B))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = ddp.iLQG(prob, x0, u0, trace_cap=2, timing=False)
    tr = r[6]
    assert tr["trace_cap"] == 2 and tr["history"]["cost"].shape[0] == 2
    assert np.array_equal(tr["truncated"], tr["iter"] - 1 > 2) and tr["truncated"].any() and any("trace_cap" in str(x.message) for x in w)
    assert not any(k.startswith("time_d") or k.startswith("time_b") or k.startswith("time_f") for k in tr)
    full = ddp.iLQG(prob, x0, u0)[6]
    assert not full["truncated"].any() and "time_derivs" in full and full["trace_cap"] >= int(full["iter"].max())


def test_pinned_result_arrays(ddp, monkeypatch):
    """ddp_host_alloc / ddp_host_free / ddp_host_trim: results of the host-pointer calls live in page-locked blocks from the library's
    cache; a freed block of the same size comes back, the results equal those in plain numpy arrays (DDP_PINNED_RESULTS=0), and views keep
    their block alive"""
    import ctypes as C
    import gc
    from oracle import np_restatement as npr
    _lib = ddp._lib
    L = _lib.lib()
    p1, p2 = C.c_void_p(), C.c_void_p()
    _lib.check(L.ddp_host_trim())                                                        # (blocks earlier tests left in the cache)
    _lib.check(L.ddp_host_alloc(C.c_size_t(3 << 20), C.byref(p1)))
    _lib.check(L.ddp_host_free(p1))
    _lib.check(L.ddp_host_alloc(C.c_size_t((3 << 20) + 5), C.byref(p2)))              # same 2 MB-rounded size: the cached block
    assert p2.value == p1.value
    _lib.check(L.ddp_host_free(p2))
    assert L.ddp_host_free(C.c_void_p(12345)) != 0 and b"not allocated" in L.ddp_last_error()
    _lib.check(L.ddp_host_trim())
    rng = np.random.default_rng(8)
    N, B = 300, 64
    P = npr.make_lq_problem(rng, T=N)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B)); u0 = 0.1 * rng.standard_normal((2, N, B))
    x, u, c = ddp.forward_pass(ddp.GaussianPolicy(), x0, u0, None, 1.0, prob, None)
    cx = np.einsum("ij,jtb->itb", P["Q"], x); cu = np.einsum("ij,jtb->itb", P["R"], u)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DDP_PINNED_RESULTS", mode)
        div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, P["Q"], np.zeros((10, 2)), P["R"], P["A"], P["B"], 1.0, 1, None, x, u)
        outs[mode] = (pol.K, pol.k, Vx, Vxx)
    assert type(outs["1"][3].base).__name__ != "NoneType"                               # Vxx (1.5 MB here) sits in a library block
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a, b)
    view = outs["1"][3][:, :, 5, 7]
    ref = view.copy()
    del outs, pol, Vx, Vxx
    gc.collect()
    monkeypatch.setenv("DDP_PINNED_RESULTS", "1")
    other = ddp.back_pass(cx, cu, P["Q"], np.zeros((10, 2)), 2.0 * P["R"], P["A"], P["B"], 1.0, 1, None, x, u)      # would reuse a freed block
    assert np.array_equal(view, ref) and not np.array_equal(other[3][:, :, 5, 7], ref)


def test_df_pendcart_structured_expm_matches_the_dense_one(ddp, monkeypatch):
    """df for the pendulum (system_pendcart.jl:137-154): the exponential of the ZoH block matrix by the routine that uses its zero last row
    (two passes: small norms in place, norms > 2.1 — controls in the hundreds — by the dense routine in a second launch that otherwise leaves
    at once) against the dense routine for every element: the same operations in the same order, the same bits; the oracle on a sample"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(77)
    N, B = 50, 9
    x = np.stack([rng.uniform(-4, 4, (N, B)), rng.standard_normal((N, B)), rng.standard_normal((N, B)), rng.standard_normal((N, B))])
    u = rng.standard_normal((1, N, B))
    u[0, 7, 3] = 400.0; u[0, 8, 3] = -2500.0; u[0, 0, 0] = 90.0; u[0, N - 1, B - 1] = 1e5; u[0, 5, 5] = np.nan      # norms beyond 2.1 (Padé 13, with squarings), a NaN control
    prob = ddp.PendcartProblem()
    monkeypatch.delenv("DDP_DF_DENSE", raising=False)
    fast = ddp.df(prob, x, u)
    monkeypatch.setenv("DDP_DF_DENSE", "1")
    dense = ddp.df(prob, x, u)
    for a, b_, nm in zip(fast, dense, ("fx", "fu", "fxx", "fxu", "fuu", "cx", "cu", "cxx", "cxu", "cuu")):
        if a is None or b_ is None:
            assert a is None and b_ is None, nm
            continue
        assert np.array_equal(a, b_), nm
    P = npr.PENDCART
    p = oc.make_problem("pendcart", 4, 1, N, Q=P["Q"], R=P["R"], pend=P)
    for b in (0, 3, 5, B - 1):
        r = oc.df(p, x[..., b], u[..., b])
        assert relerr(fast[0][..., b], r[0]) < 1e-12 and relerr(fast[1][..., b], r[1]) < 1e-12


def test_torch_imported_after_the_library_still_finds_the_gpu():
    """one HIP runtime per process: the PyTorch wheel bundles its own libamdhip64, libddp_amd.so is linked against the system ROCm — with the
    system copy loaded first a later `import torch` reported "No HIP GPUs".  The mirror loads torch's bundled runtime first when torch is
    installed (what happens anyway when torch is imported first); a fresh interpreter, library first, torch second"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; import ddp_amd; ddp_amd.default_handle();\n"
            "x, u, c = ddp_amd.forward_pass(None, np.array([3.0, 0, 0, 0]), np.zeros((1, 8)), None, 1.0, ddp_amd.PendcartProblem(), None)\n"
            "import torch; assert torch.cuda.device_count() >= 1; t = (torch.ones(4, device='cuda') * 2).sum().item(); assert t == 8.0\n"
            "x2, u2, c2 = ddp_amd.forward_pass(None, np.array([3.0, 0, 0, 0]), np.zeros((1, 8)), None, 1.0, ddp_amd.PendcartProblem(), None)\n"
            "assert np.array_equal(c, c2); print('both ok')") % root
    env = dict(os.environ); env.pop("DDP_AMD_SHARE_TORCH_HIP", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "both ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
