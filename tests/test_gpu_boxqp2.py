"""The straight-line m = 2 box-QP of the backward kernels (csrc/boxqp_dev.h, boxqp_dev2; src/boxQP.jl:58-169) through the stand-alone
entry `ddp_boxqp_f64` against the C oracle: solutions, result codes, free sets and factors on problems built to reach every one of the
9 clamp patterns of the solution (each coordinate free / at its lower / at its upper bound), every exit code the routine can leave
through (2, 4, 5, 6 — and 0 for a matrix that is not positive definite), warm starts on and off the bounds, degenerate boxes (lower ==
upper, as the padded control of an m = 1 problem inside the 2 x 2 system), and the cases it hands to the generic loop (back-tracking
line searches, more than three iterations)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    ddp_amd.default_handle()
    return ddp_amd


def _problems(rng, cnt):
    """H (2,2,cnt) SPD over six decades of conditioning, g, boxes and warm starts such that every clamp pattern occurs"""
    th = rng.uniform(0, np.pi, cnt); c, s = np.cos(th), np.sin(th)
    l1 = 10.0 ** rng.uniform(-3, 2, cnt); l2 = l1 * 10.0 ** rng.uniform(0, 4, cnt)
    H = np.empty((2, 2, cnt))
    H[0, 0] = c * c * l1 + s * s * l2; H[1, 1] = s * s * l1 + c * c * l2; H[0, 1] = H[1, 0] = c * s * (l1 - l2)
    xs = rng.standard_normal((2, cnt)) * 10.0 ** rng.uniform(-2, 1, (1, cnt))          # unconstrained minimiser
    g = -np.einsum("ijc,jc->ic", H, xs)
    w = 10.0 ** rng.uniform(-2, 1, (2, cnt))
    ctr = xs + rng.standard_normal((2, cnt)) * w * rng.choice([0.0, 0.5, 2.0, 10.0], (2, cnt))     # boxes around / beside / far from it
    lo, up = ctr - w * rng.uniform(0, 1, (2, cnt)), ctr + w * rng.uniform(0, 1, (2, cnt))
    deg = rng.uniform(size=(2, cnt)) < 0.05
    up = np.where(deg, lo, up)                                                              # degenerate boxes: lower == upper
    pick = rng.integers(0, 5, (2, cnt))
    x0 = np.where(pick == 0, lo, np.where(pick == 1, up, np.where(pick == 2, xs, np.where(pick == 3, 0.5 * (lo + up), rng.standard_normal((2, cnt)) * 5))))
    return H, g, lo, up, x0


def test_boxqp2_every_clamp_pattern_and_exit_code(ddp):
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(2026)
    cnt = 6000
    H, g, lo, up, x0 = _problems(rng, cnt)
    # not positive definite (exit 0 through the swallowed PosDefException), and a tiny gradient at a free warm start (exit 5)
    H[:, :, :40] *= -1.0
    x0[:, 40:140] = -np.linalg.solve(H[:, :, 40:140].transpose(2, 0, 1), g[:, 40:140].T[..., None])[..., 0].T
    lo[:, 40:140] = x0[:, 40:140] - 1.0; up[:, 40:140] = x0[:, 40:140] + 1.0
    x, res, Hf, free = ddp.boxQP(H, g, lo, up, x0)
    pats, codes, iters = set(), {}, {}
    for c in range(cnt):
        xr, rr, Hfr, fr, it = oc.boxqp(H[..., c], g[:, c], lo[:, c], up[:, c], x0[:, c])
        assert res[c] == rr, (c, res[c], rr)
        codes[rr] = codes.get(rr, 0) + 1
        iters[it] = iters.get(it, 0) + 1
        if rr < 1:
            continue
        assert np.array_equal(free[:, c], fr), (c, free[:, c], fr)
        sc = max(1.0, float(np.max(np.abs(xr))))
        assert np.max(np.abs(x[:, c] - xr)) < 1e-9 * sc, (c, x[:, c], xr)
        nf = int(fr.sum())
        if nf:
            assert np.max(np.abs(Hf[:nf, :nf, c] - Hfr)) < 1e-8 * max(1.0, float(np.max(np.abs(Hfr)))), c
        pats.add(tuple(0 if f else (1 if xr[i] == lo[i, c] else 2) for i, f in enumerate(fr)))
    assert len(pats) == 9, sorted(pats)                                   # free / lower / upper for each of the two coordinates
    for code in (0, 4, 5, 6):
        assert codes.get(code, 0) > 0, codes
    assert any(it >= 4 for it in iters), iters                            # ... some of them finished by the generic loop (a fourth iteration)


def test_boxqp2_backtracking_and_minstep_exit(ddp):
    """a line search that has to back off (Armijo fails at step 1: the unrolled part gives the problem to the loop) and exit code 2
    (`step < minStep`, boxQP.jl:148-151) reached with an Armijo constant no step can satisfy"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(5)
    cnt = 400
    H, g, lo, up, x0 = _problems(rng, cnt)
    opts = dict(maxIter=100, minGrad=1e-8, minRelImprove=1e-8, stepDec=0.6, minStep=1e-3, Armijo=1.5)     # ratio < 1.5 almost always: back off to minStep
    x, res, Hf, free = ddp.boxQP(H, g, lo, up, x0, maxIter=100, minGrad=1e-8, minRelImprove=1e-8, stepDec=0.6, minStep=1e-3, Armijo=1.5)
    seen = {}
    for c in range(cnt):
        xr, rr, Hfr, fr, it = oc.boxqp(H[..., c], g[:, c], lo[:, c], up[:, c], x0[:, c], opts)
        assert res[c] == rr, (c, res[c], rr)
        seen[rr] = seen.get(rr, 0) + 1
        if rr >= 1:
            assert np.array_equal(free[:, c], fr) and np.max(np.abs(x[:, c] - xr)) < 1e-9 * max(1.0, float(np.max(np.abs(xr))))
    assert seen.get(2, 0) > 0, seen


@pytest.mark.parametrize("maxit", [1, 2, 3, 4])
def test_boxqp2_small_iteration_limits(ddp, maxit):
    """`maxIter` below what the unrolled part assumes (iterations 1..4): the generic loop's own exits, result 1 included (boxQP.jl:167-169)"""
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(70 + maxit)
    cnt = 300
    H, g, lo, up, x0 = _problems(rng, cnt)
    opts = dict(maxIter=maxit, minGrad=1e-8, minRelImprove=1e-8, stepDec=0.6, minStep=1e-22, Armijo=0.1)
    x, res, Hf, free = ddp.boxQP(H, g, lo, up, x0, maxIter=maxit)
    for c in range(cnt):
        xr, rr, Hfr, fr, it = oc.boxqp(H[..., c], g[:, c], lo[:, c], up[:, c], x0[:, c], opts)
        assert res[c] == rr, (c, res[c], rr, it)
        if rr >= 1:
            assert np.array_equal(free[:, c], fr) and np.max(np.abs(x[:, c] - xr)) < 1e-9 * max(1.0, float(np.max(np.abs(xr))))
