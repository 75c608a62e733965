"""forward_pend_row_kernel (csrc/forward_pass_dpp.hip; src/forward_pass.jl:9-30, src/system_pendcart.jl:83-106) moves whole 16-step chunks of
its operand and result streams through LDS from 3 584 rollouts on (DDP_PEND_CHUNK=1 / 0: always / never).  Both paths run the same step
function: every output must agree bit for bit — N below, at and off multiples of 16, with and without a policy, limits, several step
sizes, a wrapped angle difference — and the chunked path against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    return ddp_amd


def _both(ddp, fn):
    out = {}
    for v in ("0", "1"):
        os.environ["DDP_PEND_CHUNK"] = v                         # (the handle re-reads its DDP_* switches when they change)
        try:
            out[v] = fn()
        finally:
            del os.environ["DDP_PEND_CHUNK"]
    return out["0"], out["1"]


@pytest.mark.parametrize("N", [5, 16, 17, 31, 32, 37, 100])
@pytest.mark.parametrize("policy", [False, True])
@pytest.mark.parametrize("lims", [None, 2.0])
def test_chunked_streams_are_bit_identical_to_the_element_wise_path(ddp, N, policy, lims):
    from ddp_amd import _lib
    rng = np.random.default_rng(100 * N + 10 * policy + (lims is not None))
    n, m, B = 4, 1, 7
    prob = ddp.PendcartProblem()
    x0 = np.array([np.pi - 0.5, 0.0, 0.0, 0.0])[:, None] + 0.1 * rng.standard_normal((n, B))
    u = 1.5 * rng.standard_normal((m, N, B))
    L = None if lims is None else np.array([[-lims, lims]])
    if policy:
        K = 0.3 * rng.standard_normal((m, n, N, B))
        k = 0.2 * rng.standard_normal((m, N, B))
        x = np.cumsum(0.05 * rng.standard_normal((n, N, B)), axis=1) + x0[:, None, :]
        pol = ddp.GaussianPolicy(N, n, m, K, k)
        alphas = np.array([1.0, 0.5, 0.1])
    else:
        pol, x, alphas = ddp.GaussianPolicy(), None, 1.0
    for diff in ((None, ddp.WrappedDiff(1)) if policy else (None,)):
        a, b_ = _both(ddp, lambda: ddp.forward_pass(pol, x0, u, x, alphas, prob, L, diff))
        assert _lib.default_handle().last_kernel(1) == "forward_dpp_kernel"      # (the name of the family forward_pend_row_kernel belongs to)
        for p, q in zip(a, b_):
            assert np.array_equal(p, q)
        assert np.all(np.isfinite(b_[0]))


def test_chunked_path_against_the_oracle(ddp):
    """the chunked path itself (not only its agreement with the other one) against the C restatement: one batch, limits, three step sizes"""
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(77)
    n, m, N, B = 4, 1, 53, 5
    prob = ddp.PendcartProblem()
    x0 = np.array([np.pi - 0.4, 0.0, 0.0, 0.0])[:, None] + 0.1 * rng.standard_normal((n, B))
    u = rng.standard_normal((m, N, B))
    K = 0.3 * rng.standard_normal((m, n, N, B)); k = 0.2 * rng.standard_normal((m, N, B))
    x = np.cumsum(0.05 * rng.standard_normal((n, N, B)), axis=1) + x0[:, None, :]
    L = np.array([[-1.5, 1.5]])
    os.environ["DDP_PEND_CHUNK"] = "1"
    try:
        xn, un, cn = ddp.forward_pass(ddp.GaussianPolicy(N, n, m, K, k), x0, u, x, np.array([1.0, 0.3]), prob, L)
    finally:
        del os.environ["DDP_PEND_CHUNK"]
    P = npr.PENDCART
    p = oc.make_problem("pendcart", 4, 1, N, Q=P["Q"], R=P["R"], pend=P)
    for b in range(B):
        for ai, al in enumerate((1.0, 0.3)):
            xo, uo, co = oc.forward_pass(p, (K[..., b], k[..., b]), x0[:, b], u[..., b], x[..., b], al, L)
            assert np.max(np.abs(xn[:, :, b, ai] - xo)) < 1e-11 and np.max(np.abs(un[:, :, b, ai] - uo)) < 1e-11     # (same arithmetic; sin / cos: pend_math.h vs libm)
            assert np.max(np.abs(cn[:, b, ai] - co)) < 1e-10 * max(1.0, np.max(np.abs(co)))
