"""GPU parity tests of the KL-constrained path (BASELINE config 5) through the C ABI: back_pass_gps, ∇kl, forward_covariance,
kl_div_wiki and the single-constraint iLQGkl loop against the committed fixtures and the CPU oracle.  Tolerance 1e-8
relative (BASELINE.json)."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8
GPS = ["kl_gps_n4m2", "kl_gps_n4m2_lims", "kl_gps_n4m2_eta_per_step", "kl_gps_n4m1_lims", "kl_gps_n10m2", "kl_gps_n4m2_diverge"]


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    import ddp_amd.kl  # noqa: F401
    ddp_amd.default_handle()
    return ddp_amd


def _lims(g):
    return None if g["lims"].size == 0 else g["lims"]


@pytest.mark.parametrize("mode", ["q4", "lane", "generic"])   # back_pass_gps: matrix-core kernel (n=4, m=1) | one lane per trajectory (n=4) |
@pytest.mark.parametrize("name", GPS)                          # the run-time-sized kernel for every shape
def test_gps_chain_golden(ddp, monkeypatch, name, mode):
    lane = "0" if mode == "generic" else "1"
    monkeypatch.setenv("DDP_GPS_LANE", lane)
    monkeypatch.setenv("DDP_GPS_Q4", "1" if mode == "q4" else "0")
    monkeypatch.setenv("DDP_FCOV_Q4", lane)          # the same split for forward_covariance (n = 4: matrix-core kernel | run-time-sized kernel)
    monkeypatch.setenv("DDP_KL_LDS", lane)           # and kl_div_wiki (operands through an LDS image | straight from global memory)
    kl = ddp.kl
    g = load_golden(name)
    N = g["u"].shape[1]
    prev = ddp.GaussianPolicy(N, g["x"].shape[0], g["u"].shape[0], g["Kp"], g["kp"], g["Sp"], g["Sip"])
    terms = kl.grad_kl(prev)
    for got, key in zip(terms, ("cxkl", "cukl", "cxxkl", "cxukl", "cuukl")):
        assert relerr(got, g[key]) < RTOL, key
    d, pol, Vx, Vxx, dV = kl.back_pass_gps(g["cx"], g["cu"], g["cxx"], g["cxu"], g["cuu"], g["fx"], g["fu"], _lims(g), g["x"], g["u"],
                                           (terms, g["etab"]))
    assert d == int(g["diverge"])
    for got, key in ((pol.K, "K"), (pol.k, "k"), (pol.Σ, "Quui"), (pol.Σi, "Quu"), (Vx, "Vx"), (Vxx, "Vxx"), (dV, "dV")):
        assert relerr(got, g[key]) < RTOL, (key, relerr(got, g[key]))
    assert np.array_equal(Vxx, np.transpose(Vxx, (1, 0, 2)))
    sig = kl.forward_covariance(kl.Model(g["fx"], g["fu"], g["R1"]), g["x"], g["u"], pol)
    assert relerr(sig, g["sigmanew"]) < RTOL
    kld = kl.kl_div_wiki(g["xnew"], g["x"], sig, pol, prev)
    fin = np.isfinite(g["kldiv"])
    assert np.array_equal(np.isfinite(kld), fin) and relerr(kld[fin], g["kldiv"][fin]) < RTOL


@pytest.mark.parametrize("m,N", [(1, 70), (2, 70), (1, 72), (1, 16)])
def test_fcov_and_kl_div_fast_kernels_match_generic(ddp, monkeypatch, m, N):
    """forward_covariance on the matrix cores (n = 4) and kl_div_wiki through the LDS image against the run-time-sized kernels (which the
    goldens and the oracle pin): a ragged batch (B not a multiple of 4, N not a multiple of 64), per-trajectory model, and the oracle on
    every trajectory.  m = 1 with N a multiple of 8 takes the chunked forward_covariance kernel (same products: the same bits as the
    step-by-step one)"""
    from oracle import oracle_ctypes as oc
    kl = ddp.kl
    rng = np.random.default_rng(40 + m)
    n, B = 4, 7

    def spd(d, s=1.0):
        a = rng.standard_normal((d, d)); return s * (a @ a.T / d + 0.5 * np.eye(d))
    fx = np.stack([np.stack([np.eye(n) + 0.05 * rng.standard_normal((n, n)) for _ in range(N)], -1) for _ in range(B)], -1)
    R1 = spd(n, 1e-2)
    K = 0.3 * rng.standard_normal((m, n, N, B)); k = 0.1 * rng.standard_normal((m, N, B))
    S = np.stack([np.stack([spd(m, 0.5) for _ in range(N)], -1) for _ in range(B)], -1)
    Si = np.stack([np.stack([np.linalg.inv(S[:, :, t, b]) for t in range(N)], -1) for b in range(B)], -1)
    Kp = 0.2 * rng.standard_normal((m, n, N, B)); kp = 0.1 * rng.standard_normal((m, N, B))
    Sip = np.stack([np.stack([spd(m, 2.0) for _ in range(N)], -1) for _ in range(B)], -1)
    Sp = np.stack([np.stack([np.linalg.inv(Sip[:, :, t, b]) for t in range(N)], -1) for b in range(B)], -1)
    x = rng.standard_normal((n, N, B)); xnew = x + 0.1 * rng.standard_normal((n, N, B)); u = np.zeros((m, N, B))
    pol = ddp.GaussianPolicy(N, n, m, K, k, S, Si); prev = ddp.GaussianPolicy(N, n, m, Kp, kp, Sp, Sip)
    out = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("DDP_FCOV_Q4", fast); monkeypatch.setenv("DDP_KL_LDS", fast)
        for fxm, tag in ((fx, "batched"), (fx[..., 2], "shared")):
            sig = kl.forward_covariance(kl.Model(fxm, None, R1), x, u, pol)
            out[fast, tag] = (sig, kl.kl_div_wiki(xnew, x, sig, pol, prev))
    monkeypatch.setenv("DDP_FCOV_Q4", "1"); monkeypatch.setenv("DDP_FCOV_Q4L", "0")
    for fxm, tag in ((fx, "batched"), (fx[..., 2], "shared")):
        assert np.array_equal(kl.forward_covariance(kl.Model(fxm, None, R1), x, u, pol), out["1", tag][0]), tag
    monkeypatch.delenv("DDP_FCOV_Q4L")
    for tag in ("batched", "shared"):
        assert relerr(out["1", tag][0], out["0", tag][0]) < 1e-12 and relerr(out["1", tag][1], out["0", tag][1]) < 1e-11
        assert not out["1", tag][0][n:, n:, N - 1].any() and not out["1", tag][0][n:, :n, N - 1].any()       # last step: no policy block
    sig, kld = out["1", "batched"]
    for b in range(B):
        sr = oc.forward_covariance(fx[..., b], R1, K[..., b], S[..., b])
        assert relerr(sig[..., b], sr) < RTOL, b
        kr = oc.kl_div_wiki(xnew[..., b], x[..., b], sr, dict(K=K[..., b], k=k[..., b], S=S[..., b]), dict(K=Kp[..., b], k=kp[..., b], S=Sp[..., b], Si=Sip[..., b]))
        assert relerr(kld[:, b], kr) < RTOL, b


@pytest.mark.parametrize("N", [37, 40])
@pytest.mark.parametrize("lims", [False, True])
def test_gps_q4_batch_matches_lane_kernel_and_oracle(ddp, monkeypatch, lims, N):
    """back_pass_gps for n = 4, m = 1 on the matrix-core kernel (prepass c̃ = c/η + c_kl, 1/η on the products with V): a ragged batch with
    per-trajectory η, per-trajectory and shared cost Hessians, one trajectory failing — against the lane-per-trajectory kernel and the oracle.
    N = 40 (a multiple of 8) takes the kernel that moves chunks of eight steps through the LDS: same bits as the one-step kernel"""
    from oracle import oracle_ctypes as oc
    kl = ddp.kl
    rng = np.random.default_rng(19)
    n, m, B = 4, 1, 6

    def spd(d, s=1.0):
        a = rng.standard_normal((d, d)); return s * (a @ a.T / d + 0.5 * np.eye(d))
    fx = np.stack([np.stack([0.9 * np.eye(n) + 0.02 * rng.standard_normal((n, n)) for _ in range(N)], -1) for _ in range(B)], -1)   # (stable: with unstable random
    # dynamics and clamped controls Vxx reaches 1e10 .. 1e18, |grad| at the Newton point is rounding noise above minGrad = 1e-8, and
    # whether the QP ends with result 5 or breaks on sdotg >= 0 (result 0, boxQP.jl:133) is decided by the last bit)
    fu = 0.3 * rng.standard_normal((n, m, N, B))
    cx, cu, u, x = rng.standard_normal((n, N, B)), rng.standard_normal((m, N, B)), 0.3 * rng.standard_normal((m, N, B)), rng.standard_normal((n, N, B))
    Kp, kp = 0.2 * rng.standard_normal((m, n, N, B)), 0.1 * rng.standard_normal((m, N, B))
    Sip = 0.5 + rng.uniform(0, 2, (m, m, N, B)); Sp = 1.0 / Sip
    # (η >= 0.9: with clamped controls the value recursion is Vxx <- fx'Vxx fx / η + ..., i.e. 0.81 / η per step — η = 0.25 grew Vxx to 7e9 in
    # 20 steps and the QP's stopping tests then sit on rounding noise, see the remark on the dynamics above)
    etab = np.stack([1e-8 * np.ones(B), np.array([1.0, 0.9, 2.0, 1.0, 4.0, 1.25]), 1e16 * np.ones(B)])
    terms = kl.grad_kl(ddp.GaussianPolicy(N, n, m, Kp, kp, Sp, Sip))
    L = np.array([[-0.3, 0.25]]) if lims else None
    for batched_cost in (True, False):
        if batched_cost:
            cxx = np.stack([np.stack([spd(n) for _ in range(N)], -1) for _ in range(B)], -1)
            cuu = 0.5 + rng.uniform(0, 1, (m, m, N, B)); cxu = 0.05 * rng.standard_normal((n, m, N, B))
            cuu[:, :, 9, 3] = -30.0                                             # Quu < 0 there: trajectory 3 fails at step 10
        else:
            cxx = np.stack([spd(n) for _ in range(N)], -1); cuu = 0.5 + rng.uniform(0, 1, (m, m, N)); cxu = 0.05 * rng.standard_normal((n, m, N))
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("DDP_GPS_Q4", mode)
            res[mode] = kl.back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, L, x, u, (terms, etab))
        monkeypatch.setenv("DDP_GPS_Q4", "1"); monkeypatch.setenv("DDP_GPS_Q4L", "0")
        ds, ps, vxs, vxxs, dvs = kl.back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, L, x, u, (terms, etab))
        monkeypatch.delenv("DDP_GPS_Q4L")
        (d1, p1, vx1, vxx1, dv1), (d0, p0, vx0, vxx0, dv0) = res["1"], res["0"]
        assert np.array_equal(ds, d1)
        for got, ref in ((ps.K, p1.K), (ps.k, p1.k), (ps.Σ, p1.Σ), (ps.Σi, p1.Σi), (vxs, vx1), (vxxs, vxx1), (dvs, dv1)):
            assert np.array_equal(got, ref)                                      # chunked and one-step kernel: the same step function
        assert np.array_equal(d1, d0) and (not batched_cost or lims or d1[3] == 10)      # (with limits a clamped control skips the factorisation)
        for got, ref, nm in ((p1.K, p0.K, "K"), (p1.k, p0.k, "k"), (p1.Σ, p0.Σ, "Quui"), (p1.Σi, p0.Σi, "Quu"), (vx1, vx0, "Vx"), (vxx1, vxx0, "Vxx"),
                             (dv1, dv0, "dV")):
            assert relerr(got, ref) < 1e-10, (nm, batched_cost, relerr(got, ref))
        assert np.array_equal(vxx1, np.transpose(vxx1, (1, 0, 2, 3)))
        for b in range(B):
            sl = lambda a_: a_[..., b] if batched_cost else a_                   # noqa: E731
            tb = oc.kl_terms(Kp[..., b], kp[..., b], Sip[..., b])
            d, (K, k, Quui, Quu), vx, vxx, dv = oc.back_pass_gps(cx[..., b], cu[..., b], sl(cxx), sl(cxu), sl(cuu), fx[..., b], fu[..., b], L,
                                                                x[..., b], u[..., b], (tb, etab[:, b]))
            assert d1[b] == d
            for got, ref, nm in ((p1.K[..., b], K, "K"), (p1.k[..., b], k, "k"), (p1.Σ[..., b], Quui, "Quui"), (p1.Σi[..., b], Quu, "Quu"),
                                 (vx1[..., b], vx, "Vx"), (vxx1[..., b], vxx, "Vxx"), (dv1[:, b], dv, "dV")):
                assert relerr(got, ref) < RTOL, (nm, b, relerr(got, ref))


def test_gps_batch_matches_oracle(ddp):
    """a ragged batch with per-trajectory η, per-trajectory dynamics and cost, one trajectory diverging"""
    from oracle import oracle_ctypes as oc
    kl = ddp.kl
    rng = np.random.default_rng(11)
    n, m, N, B = 6, 3, 25, 5

    def spd(d, s=1.0):
        a = rng.standard_normal((d, d)); return s * (a @ a.T / d + 0.5 * np.eye(d))
    fx = np.stack([np.stack([np.eye(n) + 0.1 * rng.standard_normal((n, n)) for _ in range(N)], -1) for _ in range(B)], -1)
    fu = 0.3 * rng.standard_normal((n, m, N, B))
    cxx = np.stack([np.stack([spd(n) for _ in range(N)], -1) for _ in range(B)], -1)
    cuu = np.stack([np.stack([spd(m, 0.5) for _ in range(N)], -1) for _ in range(B)], -1)
    cuu[:, :, 9, 3] = -30 * np.eye(m)
    cxu = 0.05 * rng.standard_normal((n, m, N, B))
    cx, cu, u, x = rng.standard_normal((n, N, B)), rng.standard_normal((m, N, B)), 0.3 * rng.standard_normal((m, N, B)), rng.standard_normal((n, N, B))
    Kp, kp = 0.2 * rng.standard_normal((m, n, N, B)), 0.1 * rng.standard_normal((m, N, B))
    Sip = np.stack([np.stack([spd(m, 2.0) for _ in range(N)], -1) for _ in range(B)], -1)
    Sp = np.stack([np.stack([np.linalg.inv(Sip[:, :, t, b]) for t in range(N)], -1) for b in range(B)], -1)
    etab = np.stack([1e-8 * np.ones(B), np.array([1.0, 0.5, 2.0, 1.0, 4.0]), 1e16 * np.ones(B)])
    prev = ddp.GaussianPolicy(N, n, m, Kp, kp, Sp, Sip)
    terms = kl.grad_kl(prev)
    for lims in (None, np.stack([-0.3 * np.ones(m), 0.25 * np.ones(m)], 1)):
        div, pol, Vx, Vxx, dV = kl.back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, (terms, etab))
        for b in range(B):
            tb = oc.kl_terms(Kp[..., b], kp[..., b], Sip[..., b])
            d, (K, k, Quui, Quu), vx, vxx, dv = oc.back_pass_gps(cx[..., b], cu[..., b], cxx[..., b], cxu[..., b], cuu[..., b], fx[..., b],
                                                                fu[..., b], lims, x[..., b], u[..., b], (tb, etab[:, b]))
            assert div[b] == d and (d == 10) == (b == 3)
            for got, ref, nm in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (pol.Σ[..., b], Quui, "Quui"), (pol.Σi[..., b], Quu, "Quu"),
                                 (Vx[..., b], vx, "Vx"), (Vxx[..., b], vxx, "Vxx"), (dV[:, b], dv, "dV")):
                assert relerr(got, ref) < RTOL, (nm, b, relerr(got, ref))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ilqgkl_golden(ddp, tag):
    kl = ddp.kl
    g = load_golden("kl_ilqgkl_lq_" + tag)
    n, T = g["x"].shape
    m = g["u"].shape[0]
    eye = np.repeat(np.eye(m)[:, :, None], T, 2)
    prev = ddp.GaussianPolicy(T, n, m, np.zeros((m, n, T)), g["u"], eye, eye.copy())
    fx, fu = np.repeat(g["A"][:, :, None], T, 2), np.repeat(g["B"][:, :, None], T, 2)
    prob = ddp.LQProblem(g["A"], g["B"], g["Q"], g["R"])
    x, u, pol, Vx, Vxx, cost, tr = kl.iLQGkl(prob, g["x"], prev, kl.Model(fx, fu, g["R1"]), kl_step=float(g["kl_step"]), max_iter=50,
                                             cost=float(g["cost0"]))
    assert (int(tr["status"]), int(tr["iter"]), int(tr["n_backpass"])) == (int(g["status"]), int(g["iter"]), int(g["n_backpass"]))
    assert relerr(tr["η"], g["eta"]) < RTOL and abs(float(tr["divergence"]) - float(g["divergence"])) < 1e-7 * float(g["kl_step"])
    assert relerr(x, g["xnew"]) < RTOL and relerr(u, g["unew"]) < RTOL and relerr(pol.K, g["K"]) < RTOL
    assert relerr(pol.Σ, g["S"]) < RTOL and relerr(Vxx, g["Vxx"]) < RTOL and relerr(cost, g["cost"]) < RTOL
    assert np.array_equal(pol.k, u)                                       # traj_new.k = copy(u)  (iLQGkl.jl:239)
    assert np.array_equal(prev.k, g["u"])                                 # traj_prev.k restored (iLQGkl.jl:247)


@pytest.mark.parametrize("hostloop", ["0", "1"])     # device-resident loop | the host-array loop it replaced (kept as cross-check)
def test_ilqgkl_batch_independent_eta(ddp, monkeypatch, hostloop):
    monkeypatch.setenv("DDP_KL_HOSTLOOP", hostloop)
    _ilqgkl_batch_independent_eta(ddp)


def _ilqgkl_batch_independent_eta(ddp):
    """a batch of KL-constrained solves: every trajectory runs its own η bracket and matches its own oracle solve"""
    from oracle import oracle_ctypes as oc
    import scipy.linalg as sla
    kl = ddp.kl
    rng = np.random.default_rng(21)
    n, m, T, B, h = 6, 2, 50, 4, 0.01
    A0 = rng.standard_normal((n, n)); A = sla.expm(h * (A0 - A0.T)); Bm = h * rng.standard_normal((n, m))
    Q, R = h * np.eye(n), 0.1 * h * np.eye(m)
    u = 0.1 * rng.standard_normal((m, T, B)) * np.array([1.0, 2.0, 0.5, 3.0])
    x = np.zeros((n, T, B)); x[:, 0, :] = 1.0 + 0.1 * rng.standard_normal((n, B))
    for t in range(T - 1):
        x[:, t + 1, :] = A @ x[:, t, :] + Bm @ u[:, t, :]
    cost0 = 0.5 * np.einsum("itb,ij,jtb->b", x, Q, x) + 0.5 * np.einsum("itb,ij,jtb->b", u, R, u)
    eye = np.repeat(np.repeat(np.eye(m)[:, :, None, None], T, 2), B, 3)
    prev = ddp.GaussianPolicy(T, n, m, np.zeros((m, n, T, B)), u, eye, eye.copy())
    fx, fu, R1 = np.repeat(A[:, :, None], T, 2), np.repeat(Bm[:, :, None], T, 2), 1e-4 * np.eye(n)
    xo, uo, pol, Vx, Vxx, cost, tr = kl.iLQGkl(ddp.LQProblem(A, Bm, Q, R), x, prev, kl.Model(fx, fu, R1), kl_step=2e-4, max_iter=40, cost=cost0)
    p = oc.make_problem("lq", n, m, T, A=A, B=Bm, Q=Q, R=R)
    for b in range(B):
        pb = dict(K=np.zeros((m, n, T)), k=u[..., b], S=eye[..., b], Si=eye[..., b])
        xr, ur, polr, vx, vxx, cr, info = oc.ilqgkl(p, x[..., b], float(cost0[b]), pb, dict(fx=fx, R1=R1), kl_step=2e-4, max_iter=40)
        assert (tr["status"][b], tr["iter"][b], tr["n_backpass"][b]) == (info["status"], info["iter"], info["n_backpass"])
        assert relerr(tr["η"][:, b], info["eta"]) < RTOL
        assert relerr(xo[..., b], xr) < RTOL and relerr(uo[..., b], ur) < RTOL and relerr(pol.K[..., b], polr["K"]) < RTOL
        assert relerr(cost[:, b], cr) < RTOL
    assert len(set(tr["η"][1])) > 1                                       # the brackets really are per trajectory


def test_ilqgkl_bracket_exit_freezes_trajectory(ddp, monkeypatch):
    """a trajectory that leaves through `η > 0.999 ηmax` (iLQGkl.jl:174) while others stay live keeps the results of
    the η it was computed with — calc_η has already moved its bracket when the exit test fires.  Device loop == host-array loop."""
    import scipy.linalg as sla
    kl = ddp.kl
    rng = np.random.default_rng(5)
    n, m, T, B, h = 6, 2, 40, 5, 0.01
    A0 = rng.standard_normal((n, n)); A = sla.expm(h * (A0 - A0.T)); Bm = h * rng.standard_normal((n, m))
    Q, R = h * np.eye(n), 0.1 * h * np.eye(m)
    u = 0.1 * rng.standard_normal((m, T, B)) * np.array([1.0, 2.0, 0.5, 3.0, 1.5])
    x = np.zeros((n, T, B)); x[:, 0, :] = 1.0 + 0.1 * rng.standard_normal((n, B))
    for t in range(T - 1):
        x[:, t + 1, :] = A @ x[:, t, :] + Bm @ u[:, t, :]
    cost0 = 0.5 * np.einsum("itb,ij,jtb->b", x, Q, x) + 0.5 * np.einsum("itb,ij,jtb->b", u, R, u)
    eye = np.repeat(np.repeat(np.eye(m)[:, :, None, None], T, 2), B, 3)
    fx, fu, R1 = np.repeat(A[:, :, None], T, 2), np.repeat(Bm[:, :, None], T, 2), 1e-4 * np.eye(n)
    etab = np.repeat(np.array([1e-8, 1.0, 1e16])[:, None], B, 1)
    etab[:, 1] = [1e-8, 0.85, 0.9]; etab[:, 3] = [1e-8, 0.79, 0.8]         # upper ends below the η that meets the constraint (~0.97): these two
                                                                            # climb to 0.999 ηmax and leave through the bracket test after a few passes
    outs = {}
    for hostloop in ("0", "1"):
        monkeypatch.setenv("DDP_KL_HOSTLOOP", hostloop)
        prev = ddp.GaussianPolicy(T, n, m, np.zeros((m, n, T, B)), u.copy(), eye.copy(), eye.copy())
        outs[hostloop] = kl.iLQGkl(ddp.LQProblem(A, Bm, Q, R), x, prev, kl.Model(fx, fu, R1), kl_step=2e-4, max_iter=40, cost=cost0, ηbracket=etab)
    (xo, uo, pol, Vx, Vxx, cost, tr), (xh, uh, polh, Vxh, Vxxh, costh, trh) = outs["0"], outs["1"]
    assert set(tr["status"]) >= {1, 2} and np.array_equal(tr["status"], trh["status"]) and np.array_equal(tr["iter"], trh["iter"])
    first_exit = tr["iter"][tr["status"] == 2].min()
    assert (tr["iter"][tr["status"] == 1] > first_exit).any()              # others were still live after the first bracket exit
    for got, ref in ((xo, xh), (uo, uh), (pol.K, polh.K), (pol.Σ, polh.Σ), (pol.Σi, polh.Σi), (Vx, Vxh), (Vxx, Vxxh), (cost, costh), (tr["η"], trh["η"])):
        assert relerr(got, ref) < 1e-12


def test_ilqgkl_dev_entry_matches_oracle(ddp):
    """ddp_ilqgkl_f64_dev (the whole loop of iLQGkl.jl:91-178 in one call, device pointers): cost0 = NULL (costfun of the pre-rolled
    trajectory), the caller's own η brackets, every row of stats[12, B] against the oracle's solve of each trajectory"""
    import ctypes as C
    import scipy.linalg as sla
    from oracle import oracle_ctypes as oc
    _lib = ddp._lib
    L, h = _lib.lib(), ddp.default_handle()
    rng = np.random.default_rng(77)
    n, m, T, B, hh = 6, 2, 48, 6, 0.01
    A0 = rng.standard_normal((n, n)); A = sla.expm(hh * (A0 - A0.T)); Bm = hh * rng.standard_normal((n, m))
    Q, R = hh * np.eye(n), 0.1 * hh * np.eye(m)
    u = 0.1 * rng.standard_normal((m, T, B)) * np.array([1.0, 2.0, 0.5, 3.0, 1.5, 0.8])
    x = np.zeros((n, T, B)); x[:, 0, :] = 1.0 + 0.1 * rng.standard_normal((n, B))
    for t in range(T - 1):
        x[:, t + 1, :] = A @ x[:, t, :] + Bm @ u[:, t, :]
    cost0 = 0.5 * np.einsum("itb,ij,jtb->b", x, Q, x) + 0.5 * np.einsum("itb,ij,jtb->b", u, R, u)
    eye = np.repeat(np.repeat(np.eye(m)[:, :, None, None], T, 2), B, 3)
    fx, R1 = np.repeat(A[:, :, None], T, 2), 1e-4 * np.eye(n)
    etab = np.repeat(np.array([1e-8, 1.0, 1e16])[:, None], B, 1)
    etab[:, 2] = [1e-8, 0.85, 0.9]                                          # leaves through the bracket test (status 2)
    etab[1, 4] = 3.0
    bufs = []

    def up(a):
        p_ = h.to_device(_lib.f64(a)); bufs.append(p_); return p_

    def dev(*shape):
        p_ = h.malloc(int(np.prod(shape)) * 8); bufs.append(p_); return p_
    P = _lib.Problem()
    P.kind, P.n, P.m, P.N, P.B = 0, n, m, T, B
    P.A, P.Bm, P.Q, P.R, P.cost_diag = up(A), up(Bm), up(Q), up(R), 1
    o = _lib.ILQGKLOpts()
    L.ddp_ilqgkl_default_opts(C.byref(o))
    assert (o.kl_step, o.max_iter, tuple(o.etabracket), o.del0) == (1.0, 50, (1e-8, 1.0, 1e16), 1e-4)           # iLQGkl.jl:25-44
    o.kl_step, o.max_iter = 2e-4, 40
    d_etab = up(etab)
    d = dict(x=dev(n, T, B), u=dev(m, T, B), K=dev(m, n, T, B), S=dev(m, m, T, B), Si=dev(m, m, T, B), Vx=dev(n, T, B), Vxx=dev(n, n, T, B),
             cost=dev(T, B), dV=dev(2, B), st=dev(12, B))
    its = C.c_int(0)
    try:
        _lib.check(L.ddp_ilqgkl_f64_dev(h.raw, C.byref(P), C.byref(o), up(x), None, up(np.zeros((m, n, T, B))), up(u), up(eye), up(eye),
                                        up(fx), 0, up(R1), None, d_etab, d["x"], d["u"], d["K"], d["S"], d["Si"], d["Vx"], d["Vxx"], d["cost"],
                                        d["dV"], d["st"], C.byref(its)))
        st = h.to_host(d["st"], (12, B)); eb = h.to_host(d_etab, (3, B))
        xo, uo, Ko, So, Vxxo = (h.to_host(d[k_], sh) for k_, sh in (("x", (n, T, B)), ("u", (m, T, B)), ("K", (m, n, T, B)), ("S", (m, m, T, B)),
                                                                      ("Vxx", (n, n, T, B))))
        co, dV = h.to_host(d["cost"], (T, B)), h.to_host(d["dV"], (2, B))
    finally:
        for p_ in bufs:
            h.free(p_)
    p = oc.make_problem("lq", n, m, T, A=A, B=Bm, Q=Q, R=R)
    assert {1, 2} <= set(st[0].astype(int)) and its.value == int(st[1].max())
    for b in range(B):
        pb = dict(K=np.zeros((m, n, T)), k=u[..., b], S=eye[..., b], Si=eye[..., b])
        xr, ur, polr, vx, vxx, cr, info = oc.ilqgkl(p, x[..., b], float(cost0[b]), pb, dict(fx=fx, R1=R1), kl_step=2e-4, max_iter=40,
                                                    etab=etab[:, b])
        assert (int(st[0, b]), int(st[1, b]), int(st[2, b]), bool(st[3, b])) == (info["status"], info["iter"], info["n_backpass"], info["satisfied"]), b
        assert relerr(st[4:7, b], info["eta"]) < RTOL and relerr(eb[:, b], info["eta"]) < RTOL
        assert abs(st[7, b] - info["divergence"]) < 1e-7 * 2e-4
        assert abs(st[8, b] - cr.sum()) < 1e-10 * abs(cr.sum()) and abs(st[9, b] - (cost0[b] - cr.sum())) < 1e-9 * abs(cost0[b])
        assert abs(st[10, b] + info["dV"].sum()) < 1e-8 * np.abs(info["dV"]).max() and abs(st[11, b] - info["g_norm"]) < 1e-8 * info["g_norm"]
        assert relerr(dV[:, b], info["dV"]) < RTOL
        assert relerr(xo[..., b], xr) < RTOL and relerr(uo[..., b], ur) < RTOL and relerr(Ko[..., b], polr["K"]) < RTOL
        assert relerr(So[..., b], polr["S"]) < RTOL and relerr(Vxxo[..., b], vxx) < RTOL and relerr(co[:, b], cr) < RTOL


def test_ilqgkl_pendcart_c5_shape(ddp):
    """BASELINE config 5 = config 3 (pendcart, n=4, m=1, control limits) + the KL constraint; reduced N and B"""
    from oracle import oracle_ctypes as oc
    kl = ddp.kl
    rng = np.random.default_rng(31)
    N, B = 80, 3
    prob = ddp.PendcartProblem()
    lims = np.array([[-5.0, 5.0]])
    u = (1.5 * np.sin(np.arange(N) / 9.0))[None, :, None] * np.array([1.0, 0.7, 1.3]) + 0.05 * rng.standard_normal((1, N, B))
    x0 = np.array([np.pi - 0.6, 0, 0, 0])[:, None] + 0.05 * rng.standard_normal((4, B))
    x, _, c0 = ddp.forward_pass(None, x0, u, None, 1.0, prob, lims)
    cost0 = c0.sum(axis=0)
    fx, fu = ddp.df(prob, x, u)[:2]
    R1 = 1e-3 * np.eye(4)
    eye = np.ones((1, 1, N, B))
    prev = ddp.GaussianPolicy(N, 4, 1, np.zeros((1, 4, N, B)), u, eye, eye.copy())
    xo, uo, pol, Vx, Vxx, cost, tr = kl.iLQGkl(prob, x, prev, kl.Model(fx, fu, R1), kl_step=0.05, lims=lims, max_iter=30, cost=cost0)
    pend = dict(g=prob.g, l=prob.l, h=prob.h, d=prob.d, goal=prob.goal)
    p = oc.make_problem("pendcart", 4, 1, N, Q=prob.Q, R=prob.R, pend=pend)
    for b in range(B):
        pb = dict(K=np.zeros((1, 4, N)), k=u[..., b], S=eye[..., b], Si=eye[..., b])
        xr, ur, polr, vx, vxx, cr, info = oc.ilqgkl(p, x[..., b], float(cost0[b]), pb, dict(fx=fx[..., b], R1=R1), kl_step=0.05, lims=lims,
                                                    max_iter=30)
        assert (tr["status"][b], tr["iter"][b], tr["n_backpass"][b]) == (info["status"], info["iter"], info["n_backpass"]), b
        assert relerr(tr["η"][:, b], info["eta"]) < 1e-7
        assert relerr(xo[..., b], xr) < 1e-7 and relerr(uo[..., b], ur) < 1e-7 and relerr(pol.K[..., b], polr["K"]) < 1e-7
        assert np.abs(uo[..., b]).max() <= 5.0


def test_demo_linear_kl_smoke(ddp):
    """demo_linear_kl (src/demo_linear.jl:63-136) — the reference's own test is a smoke run (test/runtests.jl:9)"""
    out = ddp.kl.demo_linear_kl(kl_step=100.0, rng=np.random.default_rng(3), T=80, outer=2, max_iter=20)      # the upstream call
    assert out[0].shape == (10, 80) and out[1].shape == (2, 80) and np.isfinite(out[5]).all()
    # the η bracket meets the constraint with EQUALITY (iLQGkl.jl:143-158): a step of 100 overshoots, a small one descends
    x, u, traj, Vx, Vxx, cost, tr = ddp.kl.demo_linear_kl(kl_step=0.01, rng=np.random.default_rng(3), T=80, outer=4, max_iter=20)
    oc_ = tr["outercosts"]
    assert np.isfinite(oc_).all() and np.all(np.diff(oc_) < 0) and abs(float(tr["divergence"]) - 0.01) < 0.1 * 0.01 + 1e-12


def test_kl_dual_kernels_match_calc_eta(ddp):
    """ddp_kl_dual_begin / retry / update (csrc/kl.hip) against calc_η and the loop bookkeeping of the host mirror (klutils.jl:112-133,
    iLQGkl.jl:91-122,169-177), bit for bit, on random brackets and divergences incl. NaN / Inf and kl_step <= 0"""
    import ctypes as C
    _lib = ddp._lib
    kl = ddp.kl
    h = ddp.default_handle()
    L = _lib.lib()
    rng = np.random.default_rng(3)
    B = 777
    for kl_step in (0.5, 0.0):
        lo = 10.0 ** rng.uniform(-8, -2, B); hi = lo * 10.0 ** rng.uniform(0.001, 12, B)
        etab = np.asfortranarray(np.stack([lo, np.sqrt(lo * hi), hi]))
        etab[1, ::7] = 0.9995 * etab[2, ::7]                               # some sit right at the bracket exit
        del0 = 10.0 ** rng.uniform(-5, -1, B)
        mean = kl_step + rng.standard_normal(B) * np.where(rng.random(B) < 0.4, 0.03, 1.0)
        mean[5], mean[6], mean[7] = np.nan, np.inf, 0.0
        live = (rng.random(B) < 0.8).astype(np.int32)
        div = (rng.random(B) < 0.3).astype(np.int32)
        i32z = lambda: np.zeros(B, dtype=np.int32)                        # noqa: E731
        host = dict(etab=etab.copy(order="F"), eta=np.full(B, -1.0), del_=del0.copy(), divergence=np.zeros(B), satisfied=i32z(), status=i32z(),
                    live=live.copy(), pend=i32z(), iters=i32z(), nback=i32z())
        devp = {k_: h.to_device(v) if v.dtype == np.float64 else None for k_, v in host.items()}
        for k_, v in host.items():
            if devp[k_] is None:
                devp[k_] = h.malloc(v.nbytes)
                _lib.check(L.ddp_memcpy_h2d(h.raw, devp[k_], _lib.ptr(v), C.c_size_t(v.nbytes)))
        dual = _lib.KLDual(*[devp[k_] for k_ in ("etab", "eta", "del_", "divergence", "satisfied", "status", "live", "pend", "iters", "nback")])
        d_div, d_mean = h.malloc(4 * B), h.to_device(mean)
        _lib.check(L.ddp_memcpy_h2d(h.raw, d_div, _lib.ptr(div), C.c_size_t(4 * B)))
        cnt = C.c_int(-1)
        # --- begin
        _lib.check(L.ddp_kl_dual_begin_f64_dev(h.raw, B, 9, C.byref(dual), C.byref(cnt)))
        assert cnt.value == live.sum()
        host["pend"] = live.copy(); host["iters"][live == 1] = 9; host["eta"][live == 1] = host["etab"][1, live == 1]
        # --- retry
        _lib.check(L.ddp_kl_dual_retry_f64_dev(h.raw, B, C.byref(dual), d_div, C.byref(cnt)))
        bad = (host["pend"] == 1) & (div > 0)
        assert cnt.value == bad.sum()
        host["nback"][host["pend"] == 1] += 1
        host["etab"][1, bad] += host["del_"][bad]; host["del_"][bad] *= 2; host["eta"][bad] = host["etab"][1, bad]
        host["pend"] = bad.astype(np.int32)
        # --- update
        _lib.check(L.ddp_kl_dual_update_f64_dev(h.raw, B, C.c_double(kl_step), C.byref(dual), d_mean, C.byref(cnt)))
        for b in np.flatnonzero(live):
            eb, sat, dv = kl.calc_η(None, None, None, host["etab"][:, b], None, None, kl_step, _mean=mean[b])
            host["divergence"][b], host["satisfied"][b] = dv, int(sat)
            if sat:
                host["status"][b], host["live"][b] = 1, 0
            elif host["etab"][1, b] > 0.999 * host["etab"][2, b]:
                host["status"][b], host["live"][b] = 2, 0
        assert cnt.value == host["live"].sum()
        for k_, v in host.items():
            got = h.to_host(devp[k_], v.shape, v.dtype)
            assert np.array_equal(got, v, equal_nan=(v.dtype == np.float64)), (k_, kl_step)
        assert {0, 1, 2} <= set(host["status"]) or kl_step == 0.0
        for p_ in list(devp.values()) + [d_div, d_mean]:
            h.free(p_)
