"""CPU tests of the KL-path oracle (oracle/ddp_oracle_kl.c) against the committed fixtures (NumPy restatement,
tests/golden/make_golden_kl.py) and against analytic known answers derived from the cited reference lines
(src/backward_pass.jl:259-350, src/klutils.jl, src/forward_pass.jl:37-56, src/iLQGkl.jl)."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import oracle_ctypes as oc

TOL = 1e-10
GPS = ["kl_gps_n4m2", "kl_gps_n4m2_lims", "kl_gps_n4m2_eta_per_step", "kl_gps_n4m1_lims", "kl_gps_n10m2", "kl_gps_n4m2_diverge"]


def _lims(g):
    return None if g["lims"].size == 0 else g["lims"]


@pytest.mark.parametrize("name", GPS)
def test_gps_golden(name):
    g = load_golden(name)
    kl = oc.kl_terms(g["Kp"], g["kp"], g["Sip"])
    for got, key in zip(kl, ("cxkl", "cukl", "cxxkl", "cxukl", "cuukl")):
        assert relerr(got, g[key]) < TOL, key
    d, (K, k, Quui, Quu), Vx, Vxx, dV = oc.back_pass_gps(g["cx"], g["cu"], g["cxx"], g["cxu"], g["cuu"], g["fx"], g["fu"], _lims(g),
                                                       g["x"], g["u"], (kl, g["etab"]))
    assert d == int(g["diverge"])
    for got, key in ((K, "K"), (k, "k"), (Quui, "Quui"), (Quu, "Quu"), (Vx, "Vx"), (Vxx, "Vxx"), (dV, "dV")):
        assert relerr(got, g[key]) < TOL, key
    sig = oc.forward_covariance(g["fx"], g["R1"], K, Quui)
    assert relerr(sig, g["sigmanew"]) < TOL
    kld = oc.kl_div_wiki(g["xnew"], g["x"], sig, dict(K=K, k=k, S=Quui), dict(K=g["Kp"], k=g["kp"], S=g["Sp"], Si=g["Sip"]))
    fin = np.isfinite(g["kldiv"])                      # after a divergence Σ = 0 before the failing step: logdet = -Inf there
    assert np.array_equal(np.isfinite(kld), fin) and relerr(kld[fin], g["kldiv"][fin]) < 1e-9


def test_gps_reduces_to_back_pass_when_kl_terms_vanish():
    """η = 1 and zero KL terms: back_pass_gps is back_pass with λ = 0 (backward_pass.jl:286-299 vs :203-210)"""
    g = load_golden("kl_gps_n4m2")
    m, N = g["u"].shape
    n = g["fx"].shape[0]
    zero = (np.zeros((n, N)), np.zeros((m, N)), np.zeros((n, n, N)), np.zeros((m, n, N)), np.zeros((m, m, N)))
    for lims in (None, np.array([[-0.2, 0.3], [-0.25, 0.2]])):
        d, (K, k, Quui, Quu), Vx, Vxx, dV = oc.back_pass_gps(g["cx"], g["cu"], g["cxx"], g["cxu"], g["cuu"], g["fx"], g["fu"], lims, g["x"],
                                                           g["u"], (zero, [1e-8, 1.0, 1e16]))
        d0, (K0, k0, Quu0), Vx0, Vxx0, dV0 = oc.back_pass(g["cx"], g["cu"], g["cxx"], g["cxu"], g["cuu"], g["fx"], g["fu"], 0.0, 1, lims,
                                                        g["x"], g["u"])
        assert d == d0 == 0
        assert relerr(K, K0) < 1e-12 and relerr(k, k0) < 1e-12 and relerr(Vxx, Vxx0) < 1e-12 and relerr(Vx, Vx0) < 1e-12
        assert relerr(dV, dV0) < 1e-12
        # Quu of back_pass is the unsymmetrised, unregularised block; gps stores its symmetric part (:301)
        assert relerr(Quu[:, :, :-1], 0.5 * (Quu0 + Quu0.transpose(1, 0, 2))[:, :, :-1]) < 1e-12
        for t in range(N):                                                     # Quui = inv(Quu)  (:283,346)
            assert np.allclose(Quui[:, :, t] @ Quu[:, :, t], np.eye(m), atol=1e-10)


def test_eta_scales_the_cost_terms():
    """Q• <- Q•/η + c•kl (:294-299): with zero KL terms, doubling η halves Quu, Vx and Vxx (except the terminal Vx, Vxx)
    and leaves the gains unchanged"""
    g = load_golden("kl_gps_n10m2")
    m, N = g["u"].shape
    n = g["fx"].shape[0]
    zero = (np.zeros((n, N)), np.zeros((m, N)), np.zeros((n, n, N)), np.zeros((m, n, N)), np.zeros((m, m, N)))
    # a one-step problem isolates the scaling: N = 2
    sl = lambda a: a[..., -2:]
    args = [sl(g[key]) for key in ("cx", "cu", "cxx", "cxu", "cuu", "fx", "fu")]
    z2 = tuple(sl(a) for a in zero)
    r1 = oc.back_pass_gps(*args, None, sl(g["x"]), sl(g["u"]), (z2, [1e-8, 1.0, 1e16]))
    r2 = oc.back_pass_gps(*args, None, sl(g["x"]), sl(g["u"]), (z2, [1e-8, 2.0, 1e16]))
    assert relerr(r2[1][0], r1[1][0]) < 1e-12 and relerr(r2[1][1], r1[1][1]) < 1e-12          # K, k
    assert relerr(2 * r2[1][3], r1[1][3]) < 1e-12                                              # Quu
    assert relerr(2 * r2[2][:, 0], r1[2][:, 0]) < 1e-12 and relerr(2 * r2[3][:, :, 0], r1[3][:, :, 0]) < 1e-12


def test_kl_terms_known_answers():
    rng = np.random.default_rng(3)
    m, n, T = 2, 3, 5
    k = rng.standard_normal((m, T))
    a = rng.standard_normal((m, m)); Si = np.repeat((a @ a.T + np.eye(m))[:, :, None], T, 2)
    cx, cu, cxx, cxu, cuu = oc.kl_terms(np.zeros((m, n, T)), k, Si)                            # K = 0  (klutils.jl:16-20)
    assert not cx.any() and not cxx.any() and not cxu.any()
    assert relerr(cu, -np.einsum("abt,bt->at", Si, k)) < 1e-14 and relerr(cuu, Si) == 0.0
    K = rng.standard_normal((m, n, T))
    cx, cu, cxx, cxu, cuu = oc.kl_terms(K, k, Si)
    # [cxx cxu'; cxu cuu] is M of KLmv (klutils.jl:28-35): PSD with the null space {(x, Kx)}
    for t in range(T):
        M = np.block([[cxx[:, :, t], cxu[:, :, t].T], [cxu[:, :, t], cuu[:, :, t]]])
        x = rng.standard_normal(n)
        assert np.allclose(M @ np.concatenate([x, K[:, :, t] @ x]), 0, atol=1e-12)
        assert np.linalg.eigvalsh(0.5 * (M + M.T)).min() > -1e-12


def test_kl_div_wiki_known_answers():
    rng = np.random.default_rng(4)
    n, m, T = 3, 2, 4
    K = rng.standard_normal((m, n, T)); k = rng.standard_normal((m, T))
    a = rng.standard_normal((m, m)); S = np.repeat((a @ a.T + np.eye(m))[:, :, None], T, 2)
    Si = np.stack([np.linalg.inv(S[:, :, t]) for t in range(T)], -1)
    x = rng.standard_normal((n, T)); sig = np.zeros((n + m, n + m, T))
    pol = dict(K=K, k=k, S=S, Si=Si)
    assert np.abs(oc.kl_div_wiki(x, x, sig, pol, pol)).max() < 1e-12                           # identical policies
    # same gains, μ = 0, Σt = 0: the Gaussian KL  ½(tr(Σp⁻¹Σn) + Δk'Σp⁻¹Δk − m + ln|Σp| − ln|Σn|)   (klutils.jl:92)
    new = dict(K=K, k=k + 0.3, S=2.0 * S)
    want = 0.5 * (2.0 * m + np.einsum("a,abt,b->t", 0.3 * np.ones(m), Si, 0.3 * np.ones(m)) - m - m * np.log(2.0))
    assert relerr(oc.kl_div_wiki(x, x, sig, new, pol), want) < 1e-12
    # a negative-determinant covariance makes logdet throw -> the reference returns Inf (:95-99)
    bad = dict(K=K, k=k, S=np.repeat(np.array([[1.0, 2.0], [2.0, 1.0]])[:, :, None], T, 2))
    assert np.isscalar(oc.kl_div_wiki(x, x, sig, bad, pol)) and oc.kl_div_wiki(x, x, sig, bad, pol) == np.inf


def test_forward_covariance_is_the_lyapunov_iteration():
    rng = np.random.default_rng(5)
    n, m, N = 3, 1, 200
    A = 0.5 * np.eye(n) + 0.1 * rng.standard_normal((n, n))
    R1 = 0.1 * np.eye(n)
    K = rng.standard_normal((m, n, N)); Sg = np.ones((m, m, N))
    S = oc.forward_covariance(np.repeat(A[:, :, None], N, 2), R1, K, Sg)
    import scipy.linalg as sla
    X = sla.solve_discrete_lyapunov(A, R1)                                                    # fixed point of Σ⁺ = AΣA' + R1  (:49)
    assert relerr(S[:n, :n, -1], X) < 1e-10
    t = 7
    assert relerr(S[n:, :n, t], K[:, :, t] @ S[:n, :n, t]) < 1e-14                            # (:50)
    assert relerr(S[n:, n:, t], K[:, :, t] @ S[:n, :n, t] @ K[:, :, t].T + Sg[:, :, t]) < 1e-14   # (:52)
    assert not S[n:, :, -1].any()                                                             # `undef` upstream, zero here


def test_calc_eta_bracket():
    e, s = oc.calc_eta([1e-8, 1.0, 1e16], 0.5, 1.0)              # constraint slack: η too big -> upper end moves (klutils.jl:121-124)
    assert not s and e[2] == 1.0 and e[1] == max(np.sqrt(1e-8 * 1.0), 0.1)
    e, s = oc.calc_eta([1e-8, 1.0, 1e16], 3.0, 1.0)              # violated: η too small -> lower end moves, at most x10 (:125-128)
    assert not s and e[0] == 1.0 and e[1] == 10.0
    e, s = oc.calc_eta([1e-8, 1.0, 1e16], 1.05, 1.0)             # within 10 % -> satisfied, bracket untouched (:117)
    assert s and list(e) == [1e-8, 1.0, 1e16]
    e, s = oc.calc_eta([1e-8, 1.0, 1e16], 5.0, 0.0)              # kl_step <= 0 -> always satisfied (:113)
    assert s


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ilqgkl_golden_and_constraint(tag):
    g = load_golden("kl_ilqgkl_lq_" + tag)
    n, T = g["x"].shape
    m = g["u"].shape[0]
    p = oc.make_problem("lq", n, m, T, A=g["A"], B=g["B"], Q=g["Q"], R=g["R"])
    eye = np.repeat(np.eye(m)[:, :, None], T, 2)
    prev = dict(K=np.zeros((m, n, T)), k=g["u"], S=eye, Si=eye)
    model = dict(fx=np.repeat(g["A"][:, :, None], T, 2), R1=g["R1"])
    x, u, pol, Vx, Vxx, cost, info = oc.ilqgkl(p, g["x"], float(g["cost0"]), prev, model, kl_step=float(g["kl_step"]), max_iter=50)
    assert (info["status"], info["iter"], info["n_backpass"]) == (int(g["status"]), int(g["iter"]), int(g["n_backpass"]))
    assert relerr(info["eta"], g["eta"]) < 1e-10 and abs(info["divergence"] - float(g["divergence"])) < 1e-9 * float(g["kl_step"])
    assert relerr(x, g["xnew"]) < 1e-9 and relerr(u, g["unew"]) < 1e-9 and relerr(pol["K"], g["K"]) < 1e-9
    assert relerr(pol["k"], g["unew"]) == 0.0 or relerr(pol["k"], g["unew"]) < 1e-9            # traj_new.k = copy(u)  (iLQGkl.jl:239)
    # SUCCESS means the mean divergence sits within 10 % of kl_step (iLQGkl.jl:169, klutils.jl:117), and the step is an improvement
    assert info["status"] == 1 and abs(info["divergence"] - float(g["kl_step"])) < 0.1 * float(g["kl_step"])
    assert cost.sum() < float(g["cost0"])
