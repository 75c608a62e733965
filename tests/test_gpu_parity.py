"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the
committed golden fixtures.  Tolerance: 1e-8 relative (BASELINE.json north_star), fp64."""
import numpy as np
import pytest

from conftest import load_golden, par_map, relerr

pytestmark = pytest.mark.gpu

RTOL = 1e-8

BP_CASES = ["bp_lti_n10m2_reg1", "bp_lti_n10m2_reg2", "bp_lti_n10m2_lims", "bp_ltv_pendcart_lims",
            "bp_ltv_pendcart_nolims", "bp_tv_n6m3_reg1", "bp_tv_n6m3_reg2", "bp_tv_n6m3_lims",
            "bp_tv_n6m3_diverge", "bp_tv_n6m3_diverge_lims"]


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    ddp_amd.default_handle()          # raises without a GPU / without the built library
    return ddp_amd


def _lims(g):
    return None if g["lims"].size == 0 else g["lims"]


@pytest.mark.parametrize("name", BP_CASES)
def test_back_pass_golden(ddp, name):
    g = load_golden(name)
    d, pol, Vx, Vxx, dV = ddp.back_pass(g["cx"], g["cu"], g["cxx"], g["cxu"], g["cuu"], g["fx"], g["fu"], float(g["lam"]),
                                        int(g["regType"]), _lims(g), g["x"], g["u"])
    assert d == int(g["diverge"])
    for got, key in ((pol.K, "K"), (pol.k, "k"), (Vx, "Vx"), (Vxx, "Vxx"), (dV, "dV")):
        assert relerr(got, g[key]) < RTOL, (key, relerr(got, g[key]))
    if d == 0:
        assert relerr(pol.Σi, g["Quu"]) < RTOL
        assert np.array_equal(Vxx, np.transpose(Vxx, (1, 0, 2)))
    else:
        assert not pol.K[:, :, : d - 1].any() and not Vxx[:, :, : d - 1].any()


def _rand_lq_batch(rng, n, m, N, B):
    from oracle import np_restatement as npr
    P = npr.make_lq_problem(rng, n=n, m=m, T=N)
    x = rng.standard_normal((n, N, B)); u = 0.3 * rng.standard_normal((m, N, B))
    cx = np.einsum("ij,jtb->itb", P["Q"], x); cu = np.einsum("ij,jtb->itb", P["R"], u)
    return P, x, u, cx, cu


@pytest.mark.parametrize("n,m,regType,lims", [(10, 2, 1, False), (10, 2, 2, True), (4, 1, 2, True), (7, 2, 1, False),
                                              (7, 2, 2, True), (13, 4, 1, True)])
def test_back_pass_batched_vs_oracle(ddp, n, m, regType, lims):
    """batch of independent trajectories with per-trajectory λ; (7,2) and (13,4) use the run-time-sized kernel"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(100 + n)
    N, B = 50, 24
    P, x, u, cx, cu = _rand_lq_batch(rng, n, m, N, B)
    lam = 10.0 ** rng.uniform(-4, 1, B)
    L = np.stack([-0.4 * np.ones(m), 0.5 * np.ones(m)], 1) if lims else None
    cxu = 0.01 * rng.standard_normal((n, m))
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, P["Q"], cxu, P["R"], P["A"], P["B"], lam, regType, L, x, u)
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], P["Q"], cxu, P["R"], P["A"], P["B"], lam[b],
                                                  regType, L, x[..., b], u[..., b])
        assert div[b] == d
        for got, ref in ((pol.K[..., b], K), (pol.k[..., b], k), (Vx[..., b], vx), (Vxx[..., b], vxx), (dV[:, b], dv),
                         (pol.Σi[..., b], Quu)):
            assert relerr(got, ref) < RTOL


def test_back_pass_per_trajectory_dynamics(ddp):
    """a3 layout: fx[n,n,N,B], cxx[n,n,N,B] per trajectory"""
    from oracle import oracle_ctypes as oc
    g = load_golden("bp_tv_n6m3_reg1")
    B = 5
    rng = np.random.default_rng(3)
    sc = 1 + 0.1 * rng.standard_normal(B)
    fx = np.stack([g["fx"] * s for s in sc], -1); fu = np.stack([g["fu"] * s for s in sc], -1)
    cxx = np.stack([g["cxx"] * s for s in sc], -1); cxu = np.stack([g["cxu"]] * B, -1); cuu = np.stack([g["cuu"]] * B, -1)
    cx = np.stack([g["cx"]] * B, -1); cu = np.stack([g["cu"] * s for s in sc], -1); u = np.stack([g["u"]] * B, -1)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.5, 1, None, None, u)
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx[..., b], cxu[..., b], cuu[..., b], fx[..., b],
                                                  fu[..., b], 0.5, 1, None, None, u[..., b])
        assert div[b] == d == 0
        assert relerr(pol.K[..., b], K) < RTOL and relerr(Vxx[..., b], vxx) < RTOL and relerr(Vx[..., b], vx) < RTOL


def test_boxqp_golden(ddp):
    g = load_golden("boxqp")
    for t in range(len(g["m"])):
        m = int(g["m"][t])
        x, res, Hf, free = ddp.boxQP(g["H"][t][:m, :m], g["g"][t][:m], g["lower"][t][:m], g["upper"][t][:m], g["x0"][t][:m])
        assert res == int(g["result"][t])
        assert np.max(np.abs(x - g["x"][t][:m])) < 1e-10
        assert np.array_equal(free, g["free"][t][:m].astype(bool))
        nf = int(free.sum())
        assert relerr(Hf, g["Hfree"][t][:nf, :nf]) < RTOL


def test_boxqp_batched_vs_oracle(ddp):
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(9)
    m, cnt = 4, 300
    a = rng.standard_normal((m, m, cnt)); H = np.einsum("ikc,jkc->ijc", a, a) + 0.05 * np.eye(m)[:, :, None]
    g = 2 * rng.standard_normal((m, cnt)); lo = -np.abs(rng.standard_normal((m, cnt))); up = np.abs(rng.standard_normal((m, cnt)))
    x0 = rng.standard_normal((m, cnt))
    x, res, Hf, free = ddp.boxQP(H, g, lo, up, x0)
    for c in range(cnt):
        xr, rr, Hfr, fr, _ = oc.boxqp(H[..., c], g[:, c], lo[:, c], up[:, c], x0[:, c])
        assert res[c] == rr and np.array_equal(free[:, c], fr)
        assert np.max(np.abs(x[:, c] - xr)) < 1e-10


@pytest.mark.parametrize("name", ["fwd_lq_n10m2", "fwd_lq_n10m2_lims"])
def test_forward_pass_lq_golden(ddp, name):
    g = load_golden(name)
    prob = ddp.LQProblem(g["A"], g["B"], g["Q"], g["R"])
    N = g["u"].shape[1]
    pol = ddp.GaussianPolicy(N, 10, 2, g["K"], g["k"])
    lims = g["lims"] if "lims" in g else None
    xn, un, cn = ddp.forward_pass(pol, g["x0"], g["u"], g["x"], g["alphas"], prob, lims)
    assert relerr(xn, g["xnew"]) < RTOL and relerr(un, g["unew"]) < RTOL and relerr(cn, g["cnew"]) < RTOL
    # empty policy (initial rollout, iLQG.jl:185)
    xe, ue, ce = ddp.forward_pass(ddp.GaussianPolicy(), g["x0"], g["u"], None, 1.0, prob, lims)
    if lims is None:
        assert np.array_equal(ue, g["u"]) and relerr(xe, g["x"]) < 1e-12 and relerr(ce, g["cost0"]) < 1e-10


def test_forward_pass_pendcart_golden(ddp):
    g = load_golden("fwd_pendcart")
    N = g["u"].shape[1]
    pol = ddp.GaussianPolicy(N, 4, 1, g["K"], g["k"])
    xn, un, cn = ddp.forward_pass(pol, g["x0"], g["u"], g["x"], g["alphas"], ddp.PendcartProblem(), g["lims"])
    assert cn.shape[0] == N + 1
    assert relerr(xn, g["xnew"]) < RTOL and relerr(un, g["unew"]) < RTOL and relerr(cn, g["cnew"]) < RTOL


def test_df_pendcart_golden(ddp):
    g = load_golden("df_pendcart")
    fx, fu, _, _, _, cx, cu, cxx, cxu, cuu = ddp.df(ddp.PendcartProblem(), g["x"], g["u"])
    for got, key in ((fx, "fx"), (fu, "fu"), (cx, "cx"), (cu, "cu")):
        assert relerr(got, g[key]) < 1e-12, key


def test_ilqg_lq_golden(ddp):
    g = load_golden("ilqg_lq_n10m2")
    prob = ddp.LQProblem(g["A"], g["B"], g["Q"], g["R"])
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(prob, g["x0"], g["u0"])
    st = tr["stats"][:, 0]
    assert (int(st[0]), int(st[1]), int(st[3]), int(st[4])) == (int(g["status"]), int(g["iter"]), int(g["n_backpass"]), int(g["n_forward"]))
    assert st[5] == float(g["lam"])
    for got, key in ((x, "x"), (u, "u"), (pol.K, "K"), (Vx, "Vx"), (Vxx, "Vxx"), (cost, "cost")):
        assert relerr(got, g[key]) < RTOL, (key, relerr(got, g[key]))
    assert np.max(np.abs(pol.k - g["k"])) < 1e-10
    assert relerr(tr["cost"], g["tr_cost"]) < 1e-9


def test_ilqg_pendcart_golden(ddp):
    g = load_golden("ilqg_pendcart")
    T = int(g["T"])
    kw = dict(regType=2, α=10.0 ** np.linspace(0.2, -3, 6), λmax=1e15, tol_fun=1e-8, tol_grad=1e-8, max_iter=1000)
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(ddp.PendcartProblem(), g["x0"], np.zeros((1, T)), lims=5.0 * np.array([[-1.0, 1.0]]), **kw)
    st = tr["stats"][:, 0]
    # Near convergence the accept/reject decisions sit at the rounding floor of sum(cost) (cost ~3e4, cost
    # changes ~1e-9, tol_fun = 1e-8): the summation order of the cost decides whether the last iterations
    # end with "cost change < tol_fun" (2) or run λ up to λmax (3) — see tests/test_oracle.py.  Both are the
    # same converged solution; what must agree is the solution itself.
    assert int(st[0]) in (2, 3)
    assert abs(int(st[1]) - int(g["iter"])) <= 40
    assert abs(cost.sum() - g["cost"].sum()) < 1e-9 * g["cost"].sum()
    for got, key in ((x, "x"), (u, "u"), (Vx, "Vx"), (Vxx, "Vxx")):
        assert relerr(got, g[key]) < 1e-5, (key, relerr(got, g[key]))
    assert np.all(np.abs(u) <= 5.0)


def test_ilqg_batched_independent_state_machines(ddp):
    """every trajectory of a batch behaves like its own solve (own λ schedule / termination)"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(21)
    P = npr.make_lq_problem(rng, T=150)
    B = 12
    x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B))
    u0 = 0.1 * rng.standard_normal((2, 150, B)) * (1 + np.arange(B))[None, None, :]
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG(prob, x0, u0)
    p = oc.make_problem("lq", 10, 2, 150, A=P["A"], B=P["B"], Q=P["Q"], R=P["R"])
    for b in range(B):
        xr, ur, (Kr, kr, Quur), vxr, vxxr, cr, info = oc.ilqg(p, x0[:, b], u0[..., b])
        st = tr["stats"][:, b]
        assert (int(st[0]), int(st[1]), int(st[3]), int(st[4])) == (info["status"], info["iter"], info["n_backpass"], info["n_forward"])
        assert relerr(x[..., b], xr) < RTOL and relerr(u[..., b], ur) < RTOL and relerr(Vxx[..., b], vxxr) < RTOL
        assert relerr(pol.K[..., b], Kr) < RTOL and abs(cost[:, b].sum() - cr.sum()) < 1e-9 * cr.sum()


def test_full_size_c2_properties(ddp):
    """BASELINE config 2 (n=10, m=2, N=1000, B=1024): size-independent properties + the oracle on all 1 024 trajectories"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(1234)
    n, m, N, B = 10, 2, 1000, 1024
    P = npr.make_lq_problem(rng)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
    u0 = 0.1 * rng.standard_normal((m, N, B))
    x, u, c = ddp.forward_pass(ddp.GaussianPolicy(), x0, u0, None, 1.0, prob, None)
    cx = np.einsum("ij,jtb->itb", P["Q"], x); cu = np.einsum("ij,jtb->itb", P["R"], u)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, P["Q"], np.zeros((n, m)), P["R"], P["A"], P["B"], 0.0, 1, None, x, u)
    assert not div.any()
    assert np.array_equal(Vxx, np.transpose(Vxx, (1, 0, 2, 3)))                      # exactly symmetric
    assert np.allclose(dV[0], -2 * dV[1], rtol=1e-9)                                 # λ = 0: k = -Quu⁻¹Qu
    xn, un, cn = ddp.forward_pass(pol, x0, u, x, [1.0, 0.3], prob, None)
    for j, a in enumerate((1.0, 0.3)):                                               # z == 1 for LQ problems
        z = (c.sum(0) - cn[..., j].sum(0)) / (-a * (dV[0] + a * dV[1]))
        assert np.max(np.abs(z - 1)) < 1e-7
    # ---- the oracle on EVERY trajectory of the batch (all host cores)
    p = oc.make_problem("lq", n, m, N, A=P["A"], B=P["B"], Q=P["Q"], R=P["R"])

    def check(b):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], P["Q"], np.zeros((n, m)), P["R"], P["A"], P["B"], 0.0, 1,
                                                  None, x[..., b], u[..., b])
        assert d == 0
        for got, ref, name in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (Vxx[..., b], vxx, "Vxx"), (Vx[..., b], vx, "Vx"),
                               (pol.Σi[..., b], Quu, "Quu"), (dV[:, b], dv, "dV")):
            assert relerr(got, ref) < RTOL, (name, b)
        xr, ur, cr = oc.forward_pass(p, (K, k), x0[:, b], u[..., b], x[..., b], 0.3, None)
        assert relerr(xn[..., b, 1], xr) < RTOL and relerr(un[..., b, 1], ur) < RTOL and relerr(cn[..., b, 1], cr) < RTOL, b
    par_map(check, range(B))


def test_full_size_c2_the_timed_step(ddp):
    """EXACTLY the step bench.py times (BASELINE config 2: B = 1024, N = 1000, λ = 1, regType 1, then forward_pass(α = 1)), default
    dispatch — asserted through ddp_last_kernel: the shared-LTI backward kernel and the pipeline rollout — every one of the 1 024
    trajectories against the oracle (src/backward_pass.jl:217-252, src/forward_pass.jl:9-33)"""
    from ddp_amd import _lib
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(1234)
    n, m, N, B = 10, 2, 1000, 1024
    P = npr.make_lq_problem(rng)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
    u0 = 0.1 * rng.standard_normal((m, N, B))
    x, u, c = ddp.forward_pass(ddp.GaussianPolicy(), x0, u0, None, 1.0, prob, None)
    cx = np.einsum("ij,jtb->itb", P["Q"], x); cu = np.einsum("ij,jtb->itb", P["R"], u)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, P["Q"], np.zeros((n, m)), P["R"], P["A"], P["B"], 1.0, 1, None, x, u)
    assert _lib.default_handle().last_kernel(0) == "sh_back_kernel"
    xn, un, cn = ddp.forward_pass(pol, x0, u, x, 1.0, prob, None)
    assert _lib.default_handle().last_kernel(1) == "forward_pipe4_kernel"
    assert not div.any()
    assert np.array_equal(Vxx, np.transpose(Vxx, (1, 0, 2, 3)))
    p = oc.make_problem("lq", n, m, N, A=P["A"], B=P["B"], Q=P["Q"], R=P["R"])

    def check(b):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], P["Q"], np.zeros((n, m)), P["R"], P["A"], P["B"], 1.0, 1,
                                                  None, x[..., b], u[..., b])
        assert d == 0
        for got, ref, name in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (Vxx[..., b], vxx, "Vxx"), (Vx[..., b], vx, "Vx"),
                               (pol.Σi[..., b], Quu, "Quu"), (dV[:, b], dv, "dV")):
            assert relerr(got, ref) < RTOL, (name, b)
        xr, ur, cr = oc.forward_pass(p, (K, k), x0[:, b], u[..., b], x[..., b], 1.0, None)
        assert relerr(xn[..., b], xr) < RTOL and relerr(un[..., b], ur) < RTOL and relerr(cn[..., b], cr) < RTOL, b
    par_map(check, range(B))


# ------------------------------------------------------------------ every kernel implementation of back_pass
def _tv_problem(rng, n, m, N, B):
    import scipy.linalg as sla
    fx = np.stack([np.stack([sla.expm(0.05 * (lambda a: a - a.T)(rng.standard_normal((n, n)))) for _ in range(N)], -1) for _ in range(B)], -1)
    fu = 0.1 * rng.standard_normal((n, m, N, B))
    def spd(d, s):
        a = rng.standard_normal((d, d)); return s * (a @ a.T / d + 0.5 * np.eye(d))
    cxx = np.stack([np.stack([spd(n, 0.1) for _ in range(N)], -1) for _ in range(B)], -1)
    cuu = np.stack([np.stack([spd(m, 0.05) for _ in range(N)], -1) for _ in range(B)], -1)
    cxu = 0.01 * rng.standard_normal((n, m, N, B))
    cx = 0.1 * rng.standard_normal((n, N, B)); cu = 0.1 * rng.standard_normal((m, N, B))
    u = 0.3 * rng.standard_normal((m, N, B))
    return cx, cu, cxx, cxu, cuu, fx, fu, u


@pytest.mark.parametrize("impl", ["general", "dpp", "x"])
@pytest.mark.parametrize("variant", ["lti", "ltv", "tvcost"])
@pytest.mark.parametrize("regType", [1, 2])
def test_back_pass_implementations_n10m2(ddp, monkeypatch, impl, variant, regType):
    """the three kernels (general / LDS-lean 64-lane / 16-lane DPP) on the headline shape, all dispatch variants"""
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_BACKPASS", impl)
    rng = np.random.default_rng(7)
    n, m, N, B = 10, 2, 37, 9
    cx, cu, cxx, cxu, cuu, fx, fu, u = _tv_problem(rng, n, m, N, B)
    if variant == "lti":
        fx, fu, cxx, cxu, cuu = fx[:, :, 0, 0], fu[:, :, 0, 0], cxx[:, :, 0, 0], cxu[:, :, 0, 0], cuu[:, :, 0, 0]
    elif variant == "ltv":
        cxx, cxu, cuu = cxx[:, :, 0, 0], cxu[:, :, 0, 0], cuu[:, :, 0, 0]
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, None, None, u)
    assert np.array_equal(Vxx, np.transpose(Vxx, (1, 0, 2, 3)))
    for b in range(B):
        sl = lambda a, nd: a[..., b] if a.ndim == nd + 1 else a
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], sl(cxx, 3), sl(cxu, 3), sl(cuu, 3), sl(fx, 3), sl(fu, 3),
                                                  lam[b], regType, None, None, u[..., b])
        assert div[b] == d == 0
        for got, ref in ((pol.K[..., b], K), (pol.k[..., b], k), (Vx[..., b], vx), (Vxx[..., b], vxx), (dV[:, b], dv), (pol.Σi[..., b], Quu)):
            assert relerr(got, ref) < RTOL


@pytest.mark.parametrize("lims", [False, True])
@pytest.mark.parametrize("regType", [1, 2])
@pytest.mark.parametrize("N", [16, 40, 72])
def test_back_pass_q4_kernel_variants(ddp, monkeypatch, lims, regType, N):
    """n = 4, m = 1 on the 4x4x4 matrix instruction (csrc/back_pass_q4.hip): the one-step kernel, the paired-step kernel and the
    LDS-chunk kernel against the oracle per trajectory, and bit-identical to each other; one trajectory of the batch diverges"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(100 * N + 10 * regType + int(lims))
    n, m, B = 4, 1, 11
    cx, cu, cxx, cxu, cuu, fx, fu, u = _tv_problem(rng, n, m, N, B)
    cxx, cxu, cuu = cxx[:, :, 0, 0].copy(), cxu[:, :, 0, 0].copy(), cuu[:, :, 0, 0].copy()       # time-invariant cost: all three kernels apply
    fu[:, :, N // 3, 4] = 0.0; fx[:, :, N // 3, 4] *= 1e-3
    cxx_b, cxu_b, cuu_b = (np.repeat(a_[:, :, None], B, 2) for a_ in (cxx, cxu, cuu))
    cuu_b[:, :, 4] = -1.0; cxx_b[:, :, 4] *= -1.0                                                # trajectory 4: Quu < 0 somewhere -> diverges
    L = np.array([[-0.2, 0.25]]) if lims else None
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    outs = {}
    for name, env in (("single", {"DDP_Q4_SINGLE": "1"}), ("paired", {"DDP_Q4_LDS": "0"}), ("chunked", {})):
        for k_ in ("DDP_Q4_SINGLE", "DDP_Q4_LDS"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        outs[name] = ddp.back_pass(cx, cu, cxx_b, cxu_b, cuu_b, fx, fu, lam, regType, L, None, u, cost_batched=True)
    div, pol, Vx, Vxx, dV = outs["chunked"]
    assert div[4] > 0 and not np.delete(div, 4).any()
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx_b[..., b], cxu_b[..., b], cuu_b[..., b], fx[..., b], fu[..., b],
                                                  lam[b], regType, L, None, u[..., b])
        assert d == div[b]
        for got, ref in ((pol.K[..., b], K), (pol.k[..., b], k), (Vx[..., b], vx), (Vxx[..., b], vxx), (dV[:, b], dv)):
            assert relerr(got, ref) < RTOL
        if not d:
            assert relerr(pol.Σi[..., b], Quu) < RTOL
    for name in ("single", "paired"):
        d2, p2, Vx2, Vxx2, dV2 = outs[name]
        assert np.array_equal(d2, div)
        for got, ref in ((p2.K, pol.K), (p2.k, pol.k), (Vx2, Vx), (Vxx2, Vxx), (dV2, dV), (p2.Σi, pol.Σi)):
            assert np.array_equal(got, ref, equal_nan=True), name


@pytest.mark.parametrize("impl", ["general", "dpp"])
@pytest.mark.parametrize("name", ["bp_lti_n10m2_lims", "bp_ltv_pendcart_lims", "bp_ltv_pendcart_nolims"])
def test_back_pass_implementations_limits(ddp, monkeypatch, impl, name):
    monkeypatch.setenv("DDP_BACKPASS", impl)
    g = load_golden(name)
    d, pol, Vx, Vxx, dV = ddp.back_pass(g["cx"], g["cu"], g["cxx"], g["cxu"], g["cuu"], g["fx"], g["fu"], float(g["lam"]),
                                        int(g["regType"]), _lims(g), g["x"], g["u"])
    assert d == int(g["diverge"]) == 0
    for got, key in ((pol.K, "K"), (pol.k, "k"), (Vx, "Vx"), (Vxx, "Vxx"), (dV, "dV"), (pol.Σi, "Quu")):
        assert relerr(got, g[key]) < RTOL, (key, relerr(got, g[key]))


@pytest.mark.parametrize("impl", ["general", "dpp", "x"])
def test_back_pass_divergence_per_trajectory(ddp, monkeypatch, impl):
    """a non-PD Quu in ONE trajectory of a batch stops that trajectory only (diverge index, zeros before it)"""
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_BACKPASS", impl)
    rng = np.random.default_rng(11)
    n, m, N, B = 10, 2, 30, 6
    cx, cu, cxx, cxu, cuu, fx, fu, u = _tv_problem(rng, n, m, N, B)
    cuu[:, :, 12, 3] = -np.eye(m)                      # trajectory 3 fails at step 13 (1-based)
    cuu[:, :, 20, 5] = -np.eye(m)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 1e-3, 1, None, None, u)
    assert list(div) == [0, 0, 0, 13, 0, 21]
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx[..., b], cxu[..., b], cuu[..., b], fx[..., b],
                                                  fu[..., b], 1e-3, 1, None, None, u[..., b])
        assert d == div[b]
        for got, ref in ((pol.K[..., b], K), (pol.k[..., b], k), (Vx[..., b], vx), (Vxx[..., b], vxx), (dV[:, b], dv)):
            assert relerr(got, ref) < RTOL
        if d:
            assert not pol.K[:, :, : d - 1, b].any() and not Vxx[:, :, : d - 1, b].any() and not Vx[:, : d - 1, b].any()
            assert relerr(pol.Σi[:, :, d - 1, b], Quu[:, :, d - 1]) < RTOL
