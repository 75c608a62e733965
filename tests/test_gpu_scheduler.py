"""The slot scheduler of the device-resident driver (csrc/ilqg.hip: ddp_ilqg_queue_f64, ddp_ilqg_mpc_f64): more problems than resident
trajectories, and closed-loop MPC without returning to the host.  Reference: the iteration logic of src/iLQG.jl:143-341 per solve (the
scheduler only decides WHICH problem a slot works on); the MPC hook upstream is the warm start of :193-197."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    return ddp_amd


def _pend_batch(rng, P, T=80):
    x0 = np.tile(np.array([np.pi - 0.6, 0.0, 0.0, 0.0])[:, None], (1, P))
    x0[0] += rng.uniform(-0.4, 0.4, P); x0[1] += rng.uniform(-0.3, 0.3, P)
    u0 = 0.05 * rng.standard_normal((1, T, P))
    return x0, u0


PEND = dict(lims=np.array([[-5.0, 5.0]]), regType=2, α=10.0 ** np.linspace(0.2, -3, 6), λmax=1e15, tol_fun=1e-8, tol_grad=1e-8, max_iter=60)


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_queue_pendcart_equals_standalone_batches_bit_for_bit(ddp):
    """48 pendulum solves (control limits, regType 2: boxQP path) through 16 slots.  A slot does the launches of a stand-alone solve at
    batch size 16, so every problem must come out with the BITS of `iLQG` on the batch of 16 it would have been part of — whichever
    slot it ran on and whoever its neighbours were — and the same call twice gives the same bits."""
    rng = np.random.default_rng(3)
    prob = ddp.PendcartProblem()
    P, S = 48, 16
    x0, u0 = _pend_batch(rng, P)
    q1 = ddp.iLQG_queue(prob, x0, u0, slots=S, **PEND)
    q2 = ddp.iLQG_queue(prob, x0, u0, slots=S, **PEND)
    for a, b in zip(q1[:2] + (q1[2].K, q1[2].k, q1[2].Σi) + q1[3:6] + (q1[6]["stats"],), q2[:2] + (q2[2].K, q2[2].k, q2[2].Σi) + q2[3:6] + (q2[6]["stats"],)):
        assert _same(a, b)
    iters = q1[6]["iter"]
    assert iters.max() > 1.5 * np.median(iters) or iters.max() > iters.min() + 5          # a batch worth scheduling: unequal solve lengths
    # stand-alone: any partition into batches of 16 gives the same per-trajectory bits (trajectories do not interact)
    order = rng.permutation(P)
    for c in range(0, P, S):
        sel = order[c:c + S]
        r = ddp.iLQG(prob, x0[:, sel], u0[:, :, sel], timing=False, **PEND)
        for a, b in ((q1[0], r[0]), (q1[1], r[1]), (q1[2].K, r[2].K), (q1[2].k, r[2].k), (q1[2].Σi, r[2].Σi), (q1[3], r[3]), (q1[4], r[4]), (q1[5], r[5])):
            assert _same(a[..., sel], b), c
        assert _same(q1[6]["stats"][:, sel], r[6]["stats"])
    assert q1[6]["global_iters"] < 0.75 * sum(sorted(iters)[-1:] * (P // S))              # fewer batch iterations than three lock-step batches


def test_queue_lq_matches_the_oracle(ddp):
    """LQ family (the shared-LTI backward kernel serves the slots whose λ coincide): 40 problems through 8 slots, every solve against the
    oracle's iLQG — status, iteration count, solution"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(8)
    T, P, S = 60, 40, 8
    Pm = npr.make_lq_problem(rng, T=T)
    prob = ddp.LQProblem(Pm["A"], Pm["B"], Pm["Q"], Pm["R"])
    x0 = np.ones((10, P)) + 0.1 * rng.standard_normal((10, P))
    u0 = 0.1 * rng.standard_normal((2, T, P)) * (1 + np.arange(P) % 5)[None, None, :]
    x, u, pol, Vx, Vxx, cost, tr = ddp.iLQG_queue(prob, x0, u0, slots=S)
    p = oc.make_problem("lq", 10, 2, T, A=Pm["A"], B=Pm["B"], Q=Pm["Q"], R=Pm["R"])
    for b in range(P):
        xr, ur, (Kr, kr, Quur), vxr, vxxr, cr, info = oc.ilqg(p, x0[:, b], u0[..., b])
        st = tr["stats"][:, b]
        assert (int(st[0]), int(st[1]), int(st[3]), int(st[4])) == (info["status"], info["iter"], info["n_backpass"], info["n_forward"]), b
        assert relerr(x[..., b], xr) < RTOL and relerr(u[..., b], ur) < RTOL and relerr(Vxx[..., b], vxxr) < RTOL and relerr(pol.K[..., b], Kr) < RTOL
        assert abs(cost[:, b].sum() - cr.sum()) < 1e-9 * cr.sum()


def test_queue_with_initially_diverging_problems(ddp):
    """problems whose initial rollout leaves the bound for every step size end with DDP_EXIT_INIT_DIVERGED (iLQG.jl:205-210), zero
    outputs, and hand their slot on; problems that need a smaller α for the initial rollout (iLQG.jl:181-192) take it"""
    rng = np.random.default_rng(2)
    import scipy.linalg as sla
    n, m, T, P, S = 10, 2, 40, 12, 4
    A0 = rng.standard_normal((n, n)); A = 1.3 * sla.expm(0.3 * (A0 - A0.T)); Bm = 0.5 * rng.standard_normal((n, m))
    prob = ddp.LQProblem(A, Bm, 0.01 * np.eye(n), 0.001 * np.eye(m))
    x0 = 0.01 * rng.standard_normal((n, P))
    u0 = 0.01 * rng.standard_normal((m, T, P))
    x0[:, [1, 6, 7]] *= 1e11                                        # |x_1| > 1e8: diverges whatever α (iLQG.jl:187)
    u0[:, :, [2, 9]] *= 1e6                                         # 1.3^40 amplifies: bounded only from the fifth step size on
    q = ddp.iLQG_queue(prob, x0, u0, slots=S, max_iter=5)
    st = q[6]["stats"]
    assert list(np.where(st[0] == -1)[0]) == [1, 6, 7]
    for b in (1, 6, 7):
        assert not q[0][..., b].any() and not q[1][..., b].any() and not q[2].K[..., b].any() and not q[4][..., b].any()
    for c in range(0, P, S):
        r = ddp.iLQG(prob, x0[:, c:c + S], u0[:, :, c:c + S], max_iter=5, timing=False)
        assert _same(r[6]["stats"], st[:, c:c + S]) and _same(r[0], q[0][..., c:c + S]) and _same(r[1], q[1][..., c:c + S])


def test_mpc_closed_loop_on_device(ddp):
    """5 receding-horizon steps of 6 pendulums on the device against the same loop driven from the host (iLQG, apply u_0, x_1 as the
    next initial state, mpc_shift, iLQG again): closed-loop states, applied controls and the summary of every solve, bit for bit"""
    rng = np.random.default_rng(5)
    prob = ddp.PendcartProblem()
    B, T, steps = 6, 60, 5
    x0, u0 = _pend_batch(rng, B, T)
    kw = dict(PEND, max_iter=25)
    xcl, ucl, scl, xp, up, git = ddp.iLQG_mpc(prob, x0, u0, steps, **kw)
    xs, us = x0.copy(), u0.copy()
    assert _same(xcl[:, 0], x0)
    for t in range(steps):
        r = ddp.iLQG(prob, xs, us, timing=False, **kw)
        assert _same(scl[:, t], r[6]["stats"]), t
        assert _same(xcl[:, t], r[0][:, 0]) and _same(ucl[:, t], r[1][:, 0]) and _same(xcl[:, t + 1], r[0][:, 1]), t
        xs = np.ascontiguousarray(r[0][:, 1])
        us = ddp.mpc_shift(r[1])
    assert _same(xp, r[0]) and _same(up, r[1])
    # the loop did something: every solve ran, the pendulums moved
    assert (scl[0] > 0).all() and (scl[1] > 1).all() and np.abs(xcl[:, -1] - xcl[:, 0]).max() > 1e-3
