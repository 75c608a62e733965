"""boxQP beyond the backward pass's sizes (csrc/boxqp_big.hip: one work-group per problem, 8 < m <= DDP_QP_MAX_M) against the oracle's
boxQP (oracle/ddp_oracle.c:81, a restatement of src/boxQP.jl:29-188).  Upstream's own large case is demoQP (boxQP.jl:190-199, m = 500)."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    return ddp_amd


def _qp(rng, m, scale=1.0, box=1.0):
    M = rng.standard_normal((m, m))
    H = M @ M.T + 0.5 * m * np.eye(m)                           # well conditioned: the comparison is about the algorithm, not cond(H)
    g = scale * m * rng.standard_normal(m)
    return H, g, -box * np.ones(m), box * np.ones(m), rng.standard_normal(m)


@pytest.mark.parametrize("m", [9, 12, 16, 33, 64, 100, 257, 500])
def test_boxqp_big_matches_the_oracle(ddp, m):
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(100 + m)
    for scale in (0.05, 1.0):                                   # few / many coordinates at their bounds
        H, g, lo, up, x0 = _qp(rng, m, scale)
        x, res, Hf, free = ddp.boxQP(H, g, lo, up, x0)
        xr, rr, Hfr, fr, it = oc.boxqp(H, g, lo, up, x0)
        assert res == rr and (free == fr).all(), (m, scale, res, rr)
        assert relerr(x, xr) < RTOL
        assert Hf.shape == Hfr.shape and relerr(Hf, Hfr) < RTOL
        assert 0 < free.sum() < m or scale < 0.1


def test_boxqp_big_batch_and_layout(ddp):
    """a batch of 5 problems of m = 40: every problem against the oracle; Hfree is zero outside the leading nfree x nfree upper triangle"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(7)
    m, cnt = 40, 5
    P = [_qp(rng, m, 0.3) for _ in range(cnt)]
    H = np.stack([p[0] for p in P], 2); g, lo, up, x0 = (np.stack([p[k] for p in P], 1) for k in (1, 2, 3, 4))
    x, res, Hf, free = ddp.boxQP(H, g, lo, up, x0)
    for c in range(cnt):
        xr, rr, Hfr, fr, it = oc.boxqp(*P[c])
        nf = int(fr.sum())
        assert res[c] == rr and (free[:, c] == fr).all() and relerr(x[:, c], xr) < RTOL and relerr(Hf[:nf, :nf, c], Hfr) < RTOL
        rest = Hf[:, :, c].copy(); rest[:nf, :nf] = 0.0
        assert not rest.any() and not np.tril(Hf[:nf, :nf, c], -1).any()


def test_boxqp_big_result_codes(ddp):
    """all clamped (6), unconstrained Newton point in one step (gradient below tolerance, 5 or 4), not positive definite (0, the
    swallowed PosDefException of backward_pass.jl:48-52) — the oracle's codes"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(3)
    m = 24
    H, g, lo, up, x0 = _qp(rng, m)
    cases = [(H, 1e4 * np.ones(m), lo, up, x0),                 # everything pushed to the lower bound
             (H, g, -1e6 * np.ones(m), 1e6 * np.ones(m), x0),   # bounds far away
             (H - 3.0 * m * np.eye(m), g, lo, up, x0)]          # indefinite
    for c in cases:
        x, res, Hf, free = ddp.boxQP(*c)
        xr, rr, Hfr, fr, it = oc.boxqp(*c)
        assert res == rr and (free == fr).all(), (res, rr)
        if rr != 0:
            assert relerr(x, xr) < RTOL
    assert ddp.boxQP(*cases[0])[1] == 6 and ddp.boxQP(*cases[2])[1] == 0


def test_demoqp(ddp):
    """upstream's demoQP (m = 500, bounds ±1): solves, and the solution satisfies the KKT conditions of the box-constrained problem"""
    rng = np.random.default_rng(11)
    M = rng.standard_normal((500, 500)); H = M @ M.T; g = rng.standard_normal(500)
    x, res, Hf, free = ddp.boxQP(H, g, -np.ones(500), np.ones(500), rng.standard_normal(500))
    assert res in (4, 5)
    grad = g + H @ x
    assert np.abs(grad[free]).max() < 1e-6 * np.abs(H).max()
    assert ((x[~free] == -1) & (grad[~free] > 0) | (x[~free] == 1) & (grad[~free] < 0)).all()
    out = ddp.demoQP(n=64, rng=rng)
    assert len(out) == 5 and out[1] in (4, 5)


def test_m_limit_is_reported(ddp):
    with pytest.raises(ddp.DDPError):
        ddp.boxQP(np.eye(1025), np.zeros(1025), -np.ones(1025), np.ones(1025), np.zeros(1025))


@pytest.mark.parametrize("opts", [dict(maxIter=1), dict(maxIter=2), dict(maxIter=3), dict(minRelImprove=0.5), dict(minGrad=1e3),
                                  dict(Armijo=0.9999, stepDec=0.5, minStep=0.3)])
def test_boxqp_big_options_and_exit_codes(ddp, opts):
    """the solver's keyword options (boxQP.jl:30-35): iteration cap (result 1 incl. the reference's `iter == maxIter` quirk), relative
    improvement (4), gradient tolerance (5), a line search that gives up (2) — whatever the oracle returns, with the same iterate"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(17)
    m = 30
    H, g, lo, up, x0 = _qp(rng, m, 0.4)
    x, res, Hf, free = ddp.boxQP(H, g, lo, up, x0, **opts)
    xr, rr, Hfr, fr, it = oc.boxqp(H, g, lo, up, x0, opts=dict(dict(maxIter=100, minGrad=1e-8, minRelImprove=1e-8, stepDec=0.6, minStep=1e-22, Armijo=0.1), **opts))
    assert res == rr and (free == fr).all(), (opts, res, rr)
    assert relerr(x, xr) < RTOL and relerr(Hf, Hfr) < RTOL


def test_boxqp_big_degenerate_bounds_and_start_outside(ddp):
    """coordinates with lower == upper (always at a bound), a start far outside the box, one free coordinate only"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(23)
    m = 20
    H, g, lo, up, x0 = _qp(rng, m, 0.2)
    lo[:5] = up[:5] = 0.25
    x0 = 50.0 * rng.standard_normal(m)
    for case in ((H, g, lo, up, x0), (H, 1e3 * np.r_[-1.0, np.ones(m - 1)], -np.ones(m), np.r_[1e6, np.ones(m - 1)], x0)):
        x, res, Hf, free = ddp.boxQP(*case)
        xr, rr, Hfr, fr, it = oc.boxqp(*case)
        assert res == rr and (free == fr).all() and relerr(x, xr) < RTOL and Hf.shape == Hfr.shape
        if Hf.size:
            assert relerr(Hf, Hfr) < RTOL
