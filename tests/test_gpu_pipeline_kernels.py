"""The two work-group pipelines of round 3 against the oracle AND against the one-wave kernels they replace:
   forward_pass_pipe.hip  (LQ rollout: DMA wave, two chain waves, output wave; DDP_FORWARD_PIPE=0/1)
   back_pass_mx2.hip      (chain wave + write-back wave per trajectory; DDP_MX2=0/1)
Horizons around the chunk / group sizes (12 / 8 steps), ragged batches, several step sizes, per-trajectory dynamics, the NaN-control
redo path, divergence with a writer in flight, and the step-by-step / LDS-group variants of back_pass_mx bit for bit."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    ddp_amd.default_handle()
    return ddp_amd


def _rollout_case(rng, N, B, batched_dyn=False):
    """LQ problem of the headline shape with a stabilising-ish random policy around an open-loop nominal trajectory"""
    from oracle import np_restatement as npr
    n, m = 10, 2
    P = npr.make_lq_problem(rng, n=n, m=m, T=N)
    A, Bm = P["A"], P["B"]
    if batched_dyn:
        A = np.stack([A + 0.01 * rng.standard_normal((n, n)) for _ in range(B)], -1)
        Bm = np.stack([Bm * (1 + 0.1 * rng.standard_normal()) for _ in range(B)], -1)
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
    u = 0.1 * rng.standard_normal((m, N, B))
    K = 0.3 * rng.standard_normal((m, n, N, B)); k = 0.05 * rng.standard_normal((m, N, B))
    return P, A, Bm, x0, u, K, k


@pytest.mark.parametrize("B,N,na", [(1, 1, 1), (1, 2, 1), (3, 11, 1), (4, 12, 1), (5, 13, 2), (9, 24, 3), (7, 25, 1), (6, 100, 11), (17, 37, 1),
                                    (2, 36, 16)])
@pytest.mark.parametrize("mode", ["1", "2"])
def test_forward_pipe_vs_row_kernel_and_oracle(ddp, monkeypatch, B, N, na, mode):
    """mode 2 (two rows per rollout): same statements and summation order as the row kernel for x̂ and u: bit-identical xnew / unew;
    mode 1 (one row per rollout, forward_pipe4_kernel: A x̂ formed as A x + A (x̂ - x)): to rounding; cost to rounding; oracle 1e-8"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(1000 * B + N)
    P, A, Bm, x0, u, K, k = _rollout_case(rng, N, B)
    prob = ddp.LQProblem(A, Bm, P["Q"], P["R"])
    pol = ddp.GaussianPolicy(N, 10, 2, K, k)
    monkeypatch.setenv("DDP_FORWARD_PIPE", "0")
    xnom, _, _ = ddp.forward_pass(ddp.GaussianPolicy(), x0, u, None, 1.0, prob, None)
    xnom = xnom.reshape(10, N, B)
    alphas = 10.0 ** np.linspace(0, -3, na) if na > 1 else 1.0
    row = ddp.forward_pass(pol, x0, u, xnom, alphas, prob, None)
    monkeypatch.setenv("DDP_FORWARD_PIPE", mode)
    pipe = ddp.forward_pass(pol, x0, u, xnom, alphas, prob, None)
    from ddp_amd import _lib
    assert _lib.default_handle().last_kernel(1) == ("forward_pipe4_kernel" if mode == "1" else "forward_pipe_kernel")
    if mode == "2":
        assert np.array_equal(pipe[0], row[0]) and np.array_equal(pipe[1], row[1])
    else:
        assert relerr(pipe[0], row[0]) < 1e-13 and relerr(pipe[1], row[1]) < 1e-13
    assert relerr(pipe[2], row[2]) < 1e-13
    p = oc.make_problem("lq", 10, 2, N, A=A, B=Bm, Q=P["Q"], R=P["R"])
    al = np.atleast_1d(alphas)
    for b in range(B):
        for j in sorted({0, na - 1}):
            xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], xnom[..., b], float(al[j]), None)
            g = (lambda a: a[..., b, j]) if na > 1 else (lambda a: a[..., b])
            assert relerr(g(pipe[0]), xr) < RTOL and relerr(g(pipe[1]), ur) < RTOL and relerr(g(pipe[2]), cr) < RTOL


def test_forward_pipe_per_trajectory_dynamics(ddp, monkeypatch):
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(5)
    N, B = 50, 6
    P, A, Bm, x0, u, K, k = _rollout_case(rng, N, B, batched_dyn=True)
    prob = ddp.LQProblem(A, Bm, P["Q"], P["R"], dyn_batched=True)
    monkeypatch.setenv("DDP_FORWARD_PIPE", "1")
    xnom = np.stack([oc.forward_pass(oc.make_problem("lq", 10, 2, N, A=A[..., b], B=Bm[..., b], Q=P["Q"], R=P["R"]), None, x0[:, b],
                                     u[..., b], None, 1.0, None)[0] for b in range(B)], -1)
    xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, 10, 2, K, k), x0, u, xnom, 0.5, prob, None)
    for b in range(B):
        p = oc.make_problem("lq", 10, 2, N, A=A[..., b], B=Bm[..., b], Q=P["Q"], R=P["R"])
        xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], xnom[..., b], 0.5, None)
        assert relerr(xn[..., b], xr) < RTOL and relerr(un[..., b], ur) < RTOL and relerr(cn[..., b], cr) < RTOL


@pytest.mark.parametrize("B,N,na,batched", [(1, 2, 1, True), (3, 7, 1, True), (4, 8, 1, False), (5, 9, 2, True), (9, 17, 3, False), (6, 100, 11, True),
                                            (17, 37, 1, True)])
def test_forward_pipe_time_varying_dynamics(ddp, monkeypatch, B, N, na, batched):
    """LTV rollout (A_i, B_i through the LDS image, chunks of 8 steps): the row kernel's results to rounding, oracle 1e-8; shared
    [n,n,N] and per-trajectory [n,n,N,B] dynamics, ragged batches, horizons around the chunk size, a NaN control on the redo path"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(77 * B + N)
    P, A0, B0, x0, u, K, k = _rollout_case(rng, N, B)
    sh = (10, 10, N, B) if batched else (10, 10, N)
    A = A0.reshape(10, 10, *([1] * (len(sh) - 2))) * (1.0 + 0.02 * rng.standard_normal(sh))
    Bm = B0.reshape(10, 2, *([1] * (len(sh) - 2))) * (1.0 + 0.1 * rng.standard_normal((10, 2) + sh[2:]))
    prob = ddp.LQProblem(A, Bm, P["Q"], P["R"], dyn_batched=batched)
    pol = ddp.GaussianPolicy(N, 10, 2, K, k)
    monkeypatch.setenv("DDP_FORWARD_PIPE", "0")
    xnom, _, _ = ddp.forward_pass(ddp.GaussianPolicy(), x0, u, None, 1.0, prob, None)
    xnom = xnom.reshape(10, N, B)
    alphas = 10.0 ** np.linspace(0, -3, na) if na > 1 else 1.0
    row = ddp.forward_pass(pol, x0, u, xnom, alphas, prob, None)
    monkeypatch.setenv("DDP_FORWARD_PIPE", "1")
    pipe = ddp.forward_pass(pol, x0, u, xnom, alphas, prob, None)
    # (the row kernel forms A x̂ + (B u) for time-varying dynamics, the pipeline B_1 u_1 + (B_0 u_0 + A x̂) as for time-invariant ones: rounding)
    assert relerr(pipe[0], row[0]) < 1e-13 and relerr(pipe[1], row[1]) < 1e-13 and relerr(pipe[2], row[2]) < 1e-13
    al = np.atleast_1d(alphas)
    for b in range(B):
        p = oc.make_problem("lq", 10, 2, N, A=A[..., b] if batched else A, B=Bm[..., b] if batched else Bm, Q=P["Q"], R=P["R"])
        for j in sorted({0, na - 1}):
            xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], xnom[..., b], float(al[j]), None)
            g = (lambda a: a[..., b, j]) if na > 1 else (lambda a: a[..., b])
            assert relerr(g(pipe[0]), xr) < RTOL and relerr(g(pipe[1]), ur) < RTOL and relerr(g(pipe[2]), cr) < RTOL
    if N >= 9:                                                              # u[isnan.(u)] .= 0 inside f: the step-by-step redo path of the pipeline
        k2 = k.copy(); k2[0, N // 2, 0] = np.nan
        pol2 = ddp.GaussianPolicy(N, 10, 2, K, k2)
        monkeypatch.setenv("DDP_FORWARD_PIPE", "0")
        row2 = ddp.forward_pass(pol2, x0, u, xnom, 1.0, prob, None)
        monkeypatch.setenv("DDP_FORWARD_PIPE", "1")
        pipe2 = ddp.forward_pass(pol2, x0, u, xnom, 1.0, prob, None)
        assert np.isfinite(pipe2[0]).all() and relerr(pipe2[0], row2[0]) < 1e-13 and relerr(pipe2[1], row2[1]) < 1e-13 and relerr(pipe2[2], row2[2]) < 1e-12


def test_forward_pipe_nan_control_is_redone_with_the_reference_statements(ddp, monkeypatch):
    """`u[isnan.(u)] .= 0` inside f (demo_linear.jl:43): the pipeline detects the NaN control in its output stage and recomputes that
    rollout step by step; the other rollouts of the work-group keep their pipeline results"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(9)
    N, B = 40, 6
    P, A, Bm, x0, u, K, k = _rollout_case(rng, N, B)
    u[0, 7, 1] = np.nan; u[1, 30, 4] = np.nan; k[1, 3, 4] = np.nan
    prob = ddp.LQProblem(A, Bm, P["Q"], P["R"])
    p = oc.make_problem("lq", 10, 2, N, A=A, B=Bm, Q=P["Q"], R=P["R"])
    xnom = np.stack([oc.forward_pass(p, None, x0[:, b], np.nan_to_num(u[..., b]), None, 1.0, None)[0] for b in range(B)], -1)
    pol = ddp.GaussianPolicy(N, 10, 2, K, k)
    monkeypatch.setenv("DDP_FORWARD_PIPE", "0")
    row = ddp.forward_pass(pol, x0, u, xnom, 1.0, prob, None)
    monkeypatch.setenv("DDP_FORWARD_PIPE", "1")
    pipe = ddp.forward_pass(pol, x0, u, xnom, 1.0, prob, None)
    assert pipe[1][0, 7, 1] == 0.0 and pipe[1][1, 30, 4] == 0.0 and pipe[1][1, 3, 4] == 0.0
    assert np.isfinite(pipe[0]).all() and np.isfinite(pipe[2]).all()
    for a, b_ in zip(pipe, row):
        assert relerr(a, b_) < 1e-12
    for b in range(B):
        xr, ur, cr = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], xnom[..., b], 1.0, None)
        assert relerr(pipe[0][..., b], xr) < RTOL and relerr(pipe[1][..., b], ur) < RTOL and relerr(pipe[2][..., b], cr) < RTOL


def test_ilqg_lq_same_solution_with_and_without_the_pipelines(ddp, monkeypatch):
    """whole solves (line search, accept / reject on cost differences) through the pipelines and through the one-wave kernels:
    same iteration counts, same solution"""
    from oracle import np_restatement as npr
    rng = np.random.default_rng(3)
    B, T = 12, 150
    P = npr.make_lq_problem(rng, T=T)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B)); u0 = 0.1 * rng.standard_normal((2, T, B))
    runs = []
    for pipe in ("0", "1"):
        monkeypatch.setenv("DDP_FORWARD_PIPE", pipe); monkeypatch.setenv("DDP_MX2", pipe)
        runs.append(ddp.iLQG(prob, x0, u0))
    a, b_ = runs
    assert np.array_equal(a[6]["stats"][:2], b_[6]["stats"][:2])            # status, iter
    for got, ref in zip(a[:2] + (a[2].K, a[2].k) + a[3:6], b_[:2] + (b_[2].K, b_[2].k) + b_[3:6]):
        assert relerr(got, ref) < 1e-9


# ------------------------------------------------------------------------------------------------ backward pass
def _bp_case(rng, variant, N, B):
    from test_gpu_parity import _tv_problem
    cx, cu, cxx, cxu, cuu, fx, fu, u = _tv_problem(rng, 10, 2, N, B)
    if variant in ("lti", "ltv"):
        cxx, cxu, cuu = cxx[:, :, 0, 0], cxu[:, :, 0, 0], cuu[:, :, 0, 0]
    if variant in ("lti", "tvcost_lti"):
        fx, fu = fx[:, :, 0, 0], fu[:, :, 0, 0]
    return cx, cu, cxx, cxu, cuu, fx, fu, u


@pytest.mark.parametrize("variant", ["lti", "ltv", "tvcost", "tvcost_lti"])
@pytest.mark.parametrize("regType", [1, 2])
@pytest.mark.parametrize("N", [2, 8, 9, 10, 17, 37, 100])
def test_back_pass_mx_variants(ddp, monkeypatch, variant, regType, N):
    """back_pass_mx step by step (DDP_MX_LDS=0) and in LDS groups (=1, counted vmcnt waits) agree bit for bit; back_pass_mx2 (chain +
    writer, ½(V+V') on the chain every 4th step) agrees with them to rounding, its Vxx is exactly symmetric, and all match the oracle"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(100 * N + regType)
    B = 7
    cx, cu, cxx, cxu, cuu, fx, fu, u = _bp_case(rng, variant, N, B)
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    monkeypatch.setenv("DDP_BACKPASS", "x")
    out = {}
    for tag, mx2, ldsg in (("steps", "0", "0"), ("groups", "0", "1"), ("mx2", "1", "1")):
        monkeypatch.setenv("DDP_MX2", mx2); monkeypatch.setenv("DDP_MX_LDS", ldsg)
        div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, None, None, u)
        out[tag] = (div, pol.K, pol.k, pol.Σi, Vx, Vxx, dV)
        assert np.array_equal(Vxx, np.transpose(Vxx, (1, 0, 2, 3))), tag
    for a, b_ in zip(out["steps"], out["groups"]):
        assert np.array_equal(a, b_)
    for a, b_ in zip(out["mx2"], out["groups"]):
        assert relerr(a, b_) < 1e-11
    div, K_, k_, Quu_, Vx, Vxx, dV = out["mx2"]
    for b in range(B):
        sl = lambda a, nd: a[..., b] if a.ndim == nd + 1 else a
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], sl(cxx, 3), sl(cxu, 3), sl(cuu, 3), sl(fx, 3), sl(fu, 3),
                                                  lam[b], regType, None, None, u[..., b])
        assert div[b] == d == 0
        for got, ref in ((K_[..., b], K), (k_[..., b], k), (Vx[..., b], vx), (Vxx[..., b], vxx), (dV[:, b], dv), (Quu_[..., b], Quu)):
            assert relerr(got, ref) < RTOL


@pytest.mark.parametrize("mx2", ["0", "1"])
def test_back_pass_mx_divergence_with_the_writer_in_flight(ddp, monkeypatch, mx2):
    """non-PD Quu in the middle of an LDS group, at a group boundary and in the tail below the last group: diverge index, zeros before
    the failing step (the chain waits for its writer before it zero-fills), the steps behind it as the oracle has them"""
    from oracle import oracle_ctypes as oc
    from test_gpu_parity import _tv_problem
    monkeypatch.setenv("DDP_BACKPASS", "x"); monkeypatch.setenv("DDP_MX2", mx2)
    rng = np.random.default_rng(21)
    n, m, N, B = 10, 2, 45, 9
    cx, cu, cxx, cxu, cuu, fx, fu, u = _tv_problem(rng, n, m, N, B)
    fails = {1: 41, 2: 36, 3: 35, 5: 20, 6: 3, 7: 1, 8: 43}            # trajectory -> 0-based failing step
    for b, t in fails.items():
        cuu[:, :, t, b] = -np.eye(m)
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 1e-3, 1, None, None, u)
    assert list(div) == [fails.get(b, -1) + 1 for b in range(B)]
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx[..., b], cxu[..., b], cuu[..., b], fx[..., b],
                                                  fu[..., b], 1e-3, 1, None, None, u[..., b])
        assert d == div[b]
        for got, ref in ((pol.K[..., b], K), (pol.k[..., b], k), (Vx[..., b], vx), (Vxx[..., b], vxx), (dV[:, b], dv)):
            assert relerr(got, ref) < RTOL
        if d:
            assert not pol.K[:, :, : d - 1, b].any() and not Vxx[:, :, : d - 1, b].any() and not Vx[:, : d - 1, b].any()


def test_back_pass_mx2_unstable_dynamics_keep_vxx_on_the_reference(ddp, monkeypatch):
    """open-loop unstable dynamics (rho(A) = 1.3) over a long horizon: the antisymmetric rounding residue that the chain carries
    for three steps between its exact symmetrisations grows like rho^2 per step — it must stay at rounding level"""
    from oracle import oracle_ctypes as oc
    import scipy.linalg as sla
    rng = np.random.default_rng(4)
    n, m, N, B = 10, 2, 400, 3
    A0 = rng.standard_normal((n, n))
    A = 1.3 * sla.expm(0.3 * (A0 - A0.T))
    Bm = 0.3 * rng.standard_normal((n, m)); Q = 0.1 * np.eye(n); R = 0.05 * np.eye(m)
    cx = 0.1 * rng.standard_normal((n, N, B)); cu = 0.1 * rng.standard_normal((m, N, B)); u = np.zeros((m, N, B))
    monkeypatch.setenv("DDP_BACKPASS", "x"); monkeypatch.setenv("DDP_MX2", "1")
    div, pol, Vx, Vxx, dV = ddp.back_pass(cx, cu, Q, np.zeros((n, m)), R, A, Bm, 1e-6, 1, None, None, u)
    assert np.array_equal(Vxx, np.transpose(Vxx, (1, 0, 2, 3)))
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], Q, np.zeros((n, m)), R, A, Bm, 1e-6, 1, None, None, u[..., b])
        assert d == div[b] == 0
        for got, ref in ((pol.K[..., b], K), (pol.k[..., b], k), (Vx[..., b], vx), (Vxx[..., b], vxx)):
            assert relerr(got, ref) < 1e-9


def test_ilqg_compaction_survives_a_failed_allocation(ddp, monkeypatch):
    """the block for a compacted working set cannot be allocated (simulated): compaction switches itself off BEFORE anything was
    written to the caller's arrays and the solve ends with the same results"""
    from oracle import np_restatement as npr
    rng = np.random.default_rng(17)
    B, T = 30, 120
    P = npr.make_lq_problem(rng, T=T)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B))
    u0 = 0.1 * rng.standard_normal((2, T, B)) * (1 + 3 * np.arange(B))[None, None, :]
    monkeypatch.setenv("DDP_ILQG_COMPACT", "0")
    ref = ddp.iLQG(prob, x0, u0, tol_fun=1e-6)
    monkeypatch.setenv("DDP_ILQG_COMPACT", "4"); monkeypatch.setenv("DDP_TEST_COMPACT_ALLOC_FAIL", "1")
    r = ddp.iLQG(prob, x0, u0, tol_fun=1e-6)
    assert np.array_equal(r[6]["stats"][:5], ref[6]["stats"][:5])
    for a, b_ in zip(r[:2] + (r[2].K, r[2].k) + r[3:6], ref[:2] + (ref[2].K, ref[2].k) + ref[3:6]):
        assert np.array_equal(a, b_)


@pytest.mark.parametrize("B,N,regType", [(1, 1, 1), (1, 2, 1), (3, 3, 2), (5, 4, 1), (4, 5, 1), (17, 9, 2), (16, 37, 1), (33, 64, 1), (70, 101, 2), (29, 6, 1), (61, 7, 1)])
def test_back_pass_dppw_vs_row_kernel_and_oracle(ddp, monkeypatch, B, N, regType):
    """back_pass_dppw.hip (row kernel + write-back waves, the machine-filling kernel of the LTI shape) forced on small batches: 1e-11
    against back_pass_dpp (same products; its accumulators start from the cost Hessians), Vxx exactly symmetric, the oracle at 1e-8 on every trajectory; horizons around the group
    size of 4, ragged batches (partial chain waves and work-groups), both regularisations"""
    from oracle import oracle_ctypes as oc
    from oracle import np_restatement as npr
    rng = np.random.default_rng(100 * B + N)
    P = npr.make_lq_problem(rng, T=N)
    cx = 0.05 * rng.standard_normal((10, N, B)); cu = 0.01 * rng.standard_normal((2, N, B))
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    cxu = 0.001 * rng.standard_normal((10, 2))
    outs = {}
    for tag, env in (("w", {"DDP_BACKPASS": "dpp", "DDP_DPPW": "1"}), ("d", {"DDP_BACKPASS": "dpp", "DDP_DPPW": "0"})):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        outs[tag] = ddp.back_pass(cx, cu, P["Q"], cxu, P["R"], P["A"], P["B"], lam, regType, None, None, np.zeros((2, N, B)))
    (dw, pw, vxw, vxxw, dvw), (dd, pd, vxd, vxxd, dvd) = outs["w"], outs["d"]
    assert np.array_equal(dw, dd) and not dw.any()
    for got, ref in ((pw.K, pd.K), (pw.Σi, pd.Σi), (vxxw, vxxd), (pw.k, pd.k), (vxw, vxd), (dvw, dvd)):
        assert relerr(got, ref) < 1e-11                               # same products, summed in another order (accumulators start from the cost Hessians)
    assert np.array_equal(vxxw, np.transpose(vxxw, (1, 0, 2, 3)))
    for b in range(B):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], P["Q"], cxu, P["R"], P["A"], P["B"], lam[b], regType, None, None, np.zeros((2, N)))
        assert d == 0
        for got, ref, nm in ((pw.K[..., b], K, "K"), (pw.k[..., b], k, "k"), (pw.Σi[..., b], Quu, "Quu"), (vxw[..., b], vx, "Vx"),
                             (vxxw[..., b], vxx, "Vxx"), (dvw[:, b], dv, "dV")):
            assert relerr(got, ref) < RTOL, (nm, b, relerr(got, ref))


def test_back_pass_dppw_divergence_and_inactive_trajectories(ddp, monkeypatch):
    """a trajectory whose QuuF loses positive definiteness half-way (zero-fill behind the writer's copies, the failing step's Quu kept)
    next to healthy ones, against the row kernel bit for bit and the oracle's diverge index"""
    from oracle import oracle_ctypes as oc
    from oracle import np_restatement as npr
    rng = np.random.default_rng(9)
    N, B = 30, 9
    P = npr.make_lq_problem(rng, T=N)
    cx = 0.05 * rng.standard_normal((10, N, B)); cu = 0.01 * rng.standard_normal((2, N, B))
    R = P["R"].copy()
    lam = np.full(B, 1e-3); lam[4] = -50.0 * np.abs(R).max()                 # QuuF = Quu + λI indefinite from the first step on for trajectory 4
    outs = {}
    for tag, w in (("w", "1"), ("d", "0")):
        monkeypatch.setenv("DDP_BACKPASS", "dpp"); monkeypatch.setenv("DDP_DPPW", w)
        outs[tag] = ddp.back_pass(cx, cu, P["Q"], np.zeros((10, 2)), R, P["A"], P["B"], lam, 1, None, None, np.zeros((2, N, B)))
    (dw, pw, vxw, vxxw, dvw), (dd, pd, vxd, vxxd, dvd) = outs["w"], outs["d"]
    assert np.array_equal(dw, dd) and dw[4] == N - 1 and not np.delete(dw, 4).any()
    d4 = oc.back_pass(cx[..., 4], cu[..., 4], P["Q"], np.zeros((10, 2)), R, P["A"], P["B"], lam[4], 1, None, None, np.zeros((2, N)))[0]
    assert d4 == dw[4]
    for got, ref in ((pw.K, pd.K), (pw.Σi, pd.Σi), (vxxw, vxxd), (pw.k, pd.k), (vxw, vxd), (dvw, dvd)):
        assert relerr(got, ref) < 1e-11
    assert not vxxw[:, :, : N - 2, 4].any() and vxxw[:, :, N - 1, 4].any()


def test_back_pass_dppw_is_the_default_for_large_batches(ddp, monkeypatch):
    """B >= 6144 of the shared-LTI shape goes to back_pass_dppw without any switch: against the row kernel on every trajectory, the oracle on a
    sample; N - 1 not a multiple of the group size (short first group, held Vx / k pairs on both parities)"""
    from oracle import oracle_ctypes as oc
    from oracle import np_restatement as npr
    rng = np.random.default_rng(61)
    B, N = 6144 + 37, 23
    P = npr.make_lq_problem(rng, T=N)
    cx = 0.05 * rng.standard_normal((10, N, B)); cu = 0.01 * rng.standard_normal((2, N, B))
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    u = np.zeros((2, N, B))
    monkeypatch.delenv("DDP_BACKPASS", raising=False); monkeypatch.delenv("DDP_DPPW", raising=False)
    dw, pw, vxw, vxxw, dvw = ddp.back_pass(cx, cu, P["Q"], np.zeros((10, 2)), P["R"], P["A"], P["B"], lam, 1, None, None, u)
    monkeypatch.setenv("DDP_DPPW", "0")
    dd, pd, vxd, vxxd, dvd = ddp.back_pass(cx, cu, P["Q"], np.zeros((10, 2)), P["R"], P["A"], P["B"], lam, 1, None, None, u)
    assert not dw.any() and not dd.any()
    assert np.array_equal(vxxw, np.transpose(vxxw, (1, 0, 2, 3)))
    assert not np.array_equal(vxw, vxd)                                    # (two kernels: their sums are ordered differently)
    for got, ref in ((pw.K, pd.K), (pw.Σi, pd.Σi), (vxxw, vxxd), (pw.k, pd.k), (vxw, vxd), (dvw, dvd)):
        assert relerr(got, ref) < 1e-11
    for b in list(rng.integers(0, B, 24)) + [0, B - 1, B - 37, 6143]:
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], P["Q"], np.zeros((10, 2)), P["R"], P["A"], P["B"], lam[b], 1, None, None, u[..., b])
        assert d == 0
        for got, ref in ((pw.K[..., b], K), (pw.k[..., b], k), (pw.Σi[..., b], Quu), (vxw[..., b], vx), (vxxw[..., b], vxx), (dvw[:, b], dv)):
            assert relerr(got, ref) < RTOL


def test_ilqg_with_dppw_and_finished_trajectories(ddp, monkeypatch):
    """the device-resident iLQG loop with back_pass_dppw forced: trajectories that have converged are masked out of later backward passes
    (the writer's per-trajectory store mask) — same iterations and results as with the row kernel"""
    from oracle import np_restatement as npr
    rng = np.random.default_rng(23)
    B, T = 41, 60
    P = npr.make_lq_problem(rng, T=T)
    prob = ddp.LQProblem(P["A"], P["B"], P["Q"], P["R"])
    x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B))
    u0 = 0.1 * rng.standard_normal((2, T, B)) * (1 + 3 * np.arange(B))[None, None, :]
    monkeypatch.setenv("DDP_ILQG_COMPACT", "0"); monkeypatch.setenv("DDP_BACKPASS", "dpp")
    monkeypatch.setenv("DDP_DPPW", "0")
    ref = ddp.iLQG(prob, x0, u0, tol_fun=1e-6)
    monkeypatch.setenv("DDP_DPPW", "1")
    r = ddp.iLQG(prob, x0, u0, tol_fun=1e-6)
    assert np.array_equal(r[6]["stats"][0], ref[6]["stats"][0])           # iterations per trajectory
    assert len(set(r[6]["stats"][0].tolist())) > 1                         # ... which differ: some trajectories sat out passes
    for a, b_ in zip(r[:2] + (r[2].K, r[2].k) + r[3:6], ref[:2] + (ref[2].K, ref[2].k) + ref[3:6]):
        assert relerr(a, b_) < 1e-9
