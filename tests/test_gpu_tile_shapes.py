"""The fp64-matrix-core tile kernel with RUN-TIME sizes (csrc/back_pass_mx.hip, `RT`): any n <= 10, m <= 2 without control limits inside the
(10, 2) tile layout — padded state rows are exact zeros, the unused control of m = 1 is an identity entry of the 2x2 system.  The reference's
back_pass is size-generic (src/backward_pass.jl:162-252 + :28-42, :64-76): every case against the C oracle on every trajectory, for the three
rank dispatches, both regularisations, per-trajectory operands, inactive trajectories, a diverging λ, short horizons; the row kernel and the
exact-shape tile kernel give second opinions."""
import numpy as np
import pytest

from conftest import relerr
from test_gpu_row_shapes import _problem, _check, _run

pytestmark = pytest.mark.gpu
NAME = "back_pass_mx_kernel<RT>"
SHAPES = [(1, 1), (1, 2), (2, 1), (3, 2), (4, 1), (4, 2), (5, 2), (6, 2), (7, 1), (7, 2), (8, 2), (9, 1), (9, 2), (10, 1), (10, 2)]


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    ddp_amd.default_handle()
    return ddp_amd


@pytest.mark.parametrize("n,m", SHAPES)
@pytest.mark.parametrize("kind", ["lti", "ltv", "tv"])
def test_tile_kernel_every_shape_vs_oracle(ddp, n, m, kind):
    rng = np.random.default_rng(2000 * n + 10 * m + len(kind))
    N, B = 27, 7                                      # N - 1 not a multiple of the prefetch ring (8)
    args = _problem(rng, n, m, N, B, kind)
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    for regType in (1, 2):
        out, name = _run(ddp, args, lam, regType, None, "tile")
        assert name == NAME, name
        _check(ddp, out, args, lam, regType, None, False)


@pytest.mark.parametrize("n,m", [(3, 1), (6, 2), (9, 2), (10, 1)])
def test_tile_kernel_is_the_default_dispatch_without_limits(ddp, n, m):
    from ddp_amd import _lib
    rng = np.random.default_rng(77 * n + m)
    N, B = 40, 9
    args = _problem(rng, n, m, N, B, "ltv")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.3, 1, None, x, u)
    assert _lib.default_handle().last_kernel(0) == NAME
    ref, name = _run(ddp, args, 0.3, 1, None, "row")
    assert name == "back_pass_row_kernel"
    assert np.array_equal(out[0], ref[0])
    for a_, b_ in ((out[1].K, ref[1].K), (out[1].k, ref[1].k), (out[2], ref[2]), (out[3], ref[3]), (out[4], ref[4]), (out[1].Σi, ref[1].Σi)):
        assert relerr(a_, b_) < 1e-10
    # with limits a small batch goes to the wide tile kernel (the box-QP as a wave-uniform solve), DDP_BACKPASS=row keeps the row kernel
    L = np.stack([-0.3 * np.ones(m), 0.3 * np.ones(m)], 1)
    ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.3, 1, L, x, u)
    assert _lib.default_handle().last_kernel(0) == "back_pass_mxg_kernel"


def test_tile_kernel_at_the_exact_shape_agrees_with_the_exact_kernel(ddp):
    rng = np.random.default_rng(4)
    args = _problem(rng, 10, 2, 61, 5, "tv")
    a, na = _run(ddp, args, 0.05, 2, None, "tile")
    b, nb = _run(ddp, args, 0.05, 2, None, "x")
    assert na == NAME and nb.startswith("back_pass_mx"), (na, nb)
    for p_, q_ in ((a[1].K, b[1].K), (a[1].k, b[1].k), (a[2], b[2]), (a[3], b[3]), (a[4], b[4]), (a[1].Σi, b[1].Σi)):
        assert relerr(p_, q_) < 1e-12


@pytest.mark.parametrize("n,m", [(5, 2), (7, 1), (8, 2)])
def test_tile_kernel_per_trajectory_operands_inactive_and_divergence(ddp, n, m):
    from ddp_amd import _lib
    rng = np.random.default_rng(131 * n + m)
    N, B = 19, 10
    args = _problem(rng, n, m, N, B, "btv")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    lam = np.full(B, 0.2); lam[[1, 6]] = -50.0
    for regType in (1, 2):
        out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, None, x, u)
        assert _lib.default_handle().last_kernel(0) == NAME
        if regType == 1:
            assert out[0][1] == N - 1 and out[0][6] == N - 1 and out[0][0] == 0
        _check(ddp, out, args, lam, regType, None, True)
    # a λ that fails in the middle of the horizon for some trajectories
    lam = np.full(B, 0.2); lam[[2, 5]] = -0.12
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, 1, None, x, u)
    _check(ddp, out, args, lam, 1, None, True)


@pytest.mark.parametrize("N", [1, 2, 3, 8, 9, 10, 17])
@pytest.mark.parametrize("n,m", [(3, 2), (6, 1), (9, 2)])
def test_tile_kernel_short_horizons(ddp, n, m, N):
    rng = np.random.default_rng(17 * n + m + N)
    args = _problem(rng, n, m, N, 5, "tv")
    out, name = _run(ddp, args, 0.1, 1, None, "tile")
    assert name == NAME
    _check(ddp, out, args, 0.1, 1, None, False)


def test_tile_kernel_full_size_off_shape(ddp):
    """the shape of bench.py's off-shape B without its limits at a batch the tile kernel takes by default (n=6, m=2, N=1000, B=2048, LTI):
    a sample of trajectories"""
    from ddp_amd import _lib
    rng = np.random.default_rng(6262)
    n, m, N, B = 6, 2, 1000, 2048
    args = _problem(rng, n, m, N, B, "lti")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.1, 1, None, x, u)
    assert _lib.default_handle().last_kernel(0) == NAME
    who = sorted({0, 1, 2, 3, B - 1, B - 2} | set(int(v) for v in rng.integers(0, B, 18)))
    _check(ddp, out, args, 0.1, 1, None, False, who=who)


# ------------------------------------------------------------------------------------- back_pass_mxg.hip: n <= 12, m <= 4 in one tile
WNAME = "back_pass_mxg_kernel"
# every padded size NP = 4, 8, 12, every m = 1 .. 4 (m <= 3 above n = 8), odd and even n, the corners (4,4), (8,4), (12,3)
WSHAPES = [(1, 1), (2, 3), (3, 4), (4, 2), (4, 4), (5, 3), (6, 3), (6, 4), (7, 3), (8, 1), (8, 4), (9, 3), (10, 3), (11, 2), (11, 3), (12, 1), (12, 2),
           (12, 3)]


@pytest.mark.parametrize("n,m", WSHAPES)
@pytest.mark.parametrize("kind", ["lti", "ltv", "tv"])
def test_wide_tile_kernel_every_shape_vs_oracle(ddp, n, m, kind):
    rng = np.random.default_rng(3000 * n + 10 * m + len(kind))
    N, B = 27, 7
    args = _problem(rng, n, m, N, B, kind)
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    for regType in (1, 2):
        out, name = _run(ddp, args, lam, regType, None, "wtile")
        assert name == WNAME, name
        _check(ddp, out, args, lam, regType, None, False)


@pytest.mark.parametrize("n,m", [(7, 4), (12, 3), (11, 1), (3, 3)])
def test_wide_tile_kernel_is_the_default_dispatch_without_limits(ddp, n, m):
    from ddp_amd import _lib
    rng = np.random.default_rng(177 * n + m)
    N, B = 40, 9
    args = _problem(rng, n, m, N, B, "ltv")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.3, 1, None, x, u)
    assert _lib.default_handle().last_kernel(0) == WNAME
    ref, name = _run(ddp, args, 0.3, 1, None, "row")
    assert name == "back_pass_row_kernel"
    assert np.array_equal(out[0], ref[0])
    for a_, b_ in ((out[1].K, ref[1].K), (out[1].k, ref[1].k), (out[2], ref[2]), (out[3], ref[3]), (out[4], ref[4]), (out[1].Σi, ref[1].Σi)):
        assert relerr(a_, b_) < 1e-10
    L = np.stack([-0.3 * np.ones(m), 0.3 * np.ones(m)], 1)
    ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.3, 1, L, x, u)
    assert _lib.default_handle().last_kernel(0) == "back_pass_mxg_kernel"
    _, name = _run(ddp, args, 0.3, 1, L, "row")
    assert name == "back_pass_row_kernel"


def test_wide_tile_kernel_agrees_with_the_exact_tile_kernel(ddp):
    rng = np.random.default_rng(5)
    args = _problem(rng, 10, 2, 61, 5, "tv")
    for regType in (1, 2):
        a, na = _run(ddp, args, 0.05, regType, None, "wtile")
        b, nb = _run(ddp, args, 0.05, regType, None, "x")
        assert na == WNAME and nb.startswith("back_pass_mx"), (na, nb)
        for p_, q_ in ((a[1].K, b[1].K), (a[1].k, b[1].k), (a[2], b[2]), (a[3], b[3]), (a[4], b[4]), (a[1].Σi, b[1].Σi)):
            assert relerr(p_, q_) < 1e-11


@pytest.mark.parametrize("n,m", [(5, 3), (8, 4), (12, 3)])
def test_wide_tile_kernel_per_trajectory_operands_inactive_and_divergence(ddp, n, m):
    from ddp_amd import _lib
    rng = np.random.default_rng(231 * n + m)
    N, B = 19, 10
    args = _problem(rng, n, m, N, B, "btv")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    lam = np.full(B, 0.2); lam[[1, 6]] = -50.0
    for regType in (1, 2):
        out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, None, x, u)
        assert _lib.default_handle().last_kernel(0) == WNAME
        if regType == 1:
            assert out[0][1] == N - 1 and out[0][6] == N - 1 and out[0][0] == 0
        _check(ddp, out, args, lam, regType, None, True)
    lam = np.full(B, 0.2); lam[[2, 5]] = -0.12
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, 1, None, x, u)
    _check(ddp, out, args, lam, 1, None, True)


@pytest.mark.parametrize("N", [1, 2, 3, 8, 9, 10, 17])
@pytest.mark.parametrize("n,m", [(3, 3), (7, 4), (12, 3)])
def test_wide_tile_kernel_short_horizons(ddp, n, m, N):
    rng = np.random.default_rng(27 * n + m + N)
    args = _problem(rng, n, m, N, 5, "tv")
    out, name = _run(ddp, args, 0.1, 1, None, "wtile")
    assert name == WNAME
    _check(ddp, out, args, 0.1, 1, None, False)


@pytest.mark.parametrize("n,m", [(1, 1), (3, 3), (5, 3), (7, 4), (8, 4), (9, 3), (11, 1), (11, 3), (12, 3)])
@pytest.mark.parametrize("kind", ["lti", "ltv", "tv", "btv"])
def test_wide_tile_kernel_coalesced_io_is_bit_identical_to_the_element_per_lane_path(ddp, n, m, kind):
    """DDP_MXG_COAL=0 reads and writes one 8-byte element per lane in the tile layout; the default path moves the same numbers in
    contiguous 16-byte pieces through LDS images (blocks of odd length: 8-byte tails, over-reads into the next time step) — the
    arithmetic in between is the same, so every output must agree bit for bit, odd and even block lengths, N not a multiple of the ring"""
    import os
    rng = np.random.default_rng(4000 * n + 10 * m + len(kind))
    for N, B in ((2, 3), (9, 3), (37, 6)):
        args = _problem(rng, n, m, N, B, kind)
        lam = 10.0 ** rng.uniform(-2, 0.3, B)
        lam[0] = -40.0                                          # one diverging trajectory: the zero-fill below the failing step
        for regType in (1, 2):
            out, name = _run(ddp, args, lam, regType, None, "wtile")
            os.environ["DDP_MXG_COAL"] = "0"                     # (the handle re-reads its DDP_* switches when they change)
            try:
                ref, _ = _run(ddp, args, lam, regType, None, "wtile")
            finally:
                del os.environ["DDP_MXG_COAL"]
            assert name == WNAME
            assert np.array_equal(out[0], ref[0])
            for a_, b_ in ((out[1].K, ref[1].K), (out[1].k, ref[1].k), (out[2], ref[2]), (out[3], ref[3]), (out[4], ref[4]), (out[1].Σi, ref[1].Σi)):
                assert np.array_equal(a_, b_)


# ------------------------------------------------------------------------------------------------- control limits on the wide tile kernel
@pytest.mark.parametrize("n,m", [(1, 1), (2, 2), (3, 1), (4, 4), (5, 2), (6, 2), (6, 3), (7, 4), (8, 4), (9, 2), (10, 2), (11, 3), (12, 3), (12, 1)])
@pytest.mark.parametrize("kind", ["lti", "ltv", "tv"])
def test_wide_tile_kernel_with_limits_vs_oracle(ddp, n, m, kind):
    """backward_pass.jl:43-62 on back_pass_mxg.hip: the box-QP runs as ONE wave-uniform solve per trajectory (boxqp_dev.h on LDS-broadcast
    data), clamped rows of K are zero, k comes from the QP; tight bounds (most controls clamped), loose bounds (none), both regularisations,
    the default dispatch of a small batch — every trajectory against the C oracle"""
    from ddp_amd import _lib
    rng = np.random.default_rng(5000 * n + 10 * m + len(kind))
    N, B = 27, 6
    args = _problem(rng, n, m, N, B, kind)
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    for regType, width in ((1, 0.05), (2, 0.3), (1, 50.0)):
        L = np.stack([-width * np.ones(m), 1.2 * width * np.ones(m)], 1)
        out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, L, x, u)
        assert _lib.default_handle().last_kernel(0) == "back_pass_mxg_kernel"
        _check(ddp, out, args, lam, regType, L, False)


@pytest.mark.parametrize("n,m", [(6, 2), (10, 2), (12, 3)])
def test_wide_tile_kernel_with_limits_inactive_divergence_inverted_bounds(ddp, n, m):
    """per-trajectory operands, an inactive trajectory, λ < 0 (no positive pivot: diverge at the first step, :54-55), inverted bounds
    (lims[1,1] > lims[1,2]: the Cholesky branch, :31) — against the oracle and the row kernel"""
    from ddp_amd import _lib
    rng = np.random.default_rng(77 * n + m)
    N, B = 19, 7
    args = _problem(rng, n, m, N, B, "btv")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    lam = np.full(B, 0.2); lam[[2, 5]] = -50.0
    L = np.stack([-0.2 * np.ones(m), 0.25 * np.ones(m)], 1)
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, 1, L, x, u)
    assert _lib.default_handle().last_kernel(0) == "back_pass_mxg_kernel"
    assert out[0][2] == N - 1 and out[0][5] == N - 1 and out[0][0] == 0
    _check(ddp, out, args, lam, 1, L, True)
    Li = np.stack([0.3 * np.ones(m), -0.3 * np.ones(m)], 1)
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, np.abs(lam), 1, Li, x, u)
    assert _lib.default_handle().last_kernel(0) == "back_pass_mxg_kernel"
    ref, name = _run(ddp, args, np.abs(lam), 1, Li, "row")
    assert name == "back_pass_row_kernel"
    for a_, b_ in ((out[1].K, ref[1].K), (out[1].k, ref[1].k), (out[2], ref[2]), (out[3], ref[3]), (out[4], ref[4])):
        assert relerr(a_, b_) < 1e-10


def test_wide_tile_kernel_with_limits_full_size_c2(ddp):
    """the C2 shape (n=10, m=2, N=1000, B=1024) with control limits ±0.05 on the default dispatch: a sample of trajectories against the
    oracle, the rest against the row kernel"""
    from ddp_amd import _lib
    rng = np.random.default_rng(99)
    n, m, N, B = 10, 2, 1000, 1024
    args = _problem(rng, n, m, N, B, "lti")
    cx, cu, cxx, cxu, cuu, fx, fu, x, u = args
    L = 0.05 * np.stack([-np.ones(m), np.ones(m)], 1)
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 0.1, 1, L, x, u)
    assert _lib.default_handle().last_kernel(0) == "back_pass_mxg_kernel"
    _check(ddp, out, args, 0.1, 1, L, False, who=[0, 1, 511, 1023])
    ref, name = _run(ddp, args, 0.1, 1, L, "row")
    assert name in ("back_pass_row_kernel", "back_pass_dpp_kernel")
    assert np.array_equal(out[0], ref[0])
    for a_, b_ in ((out[1].K, ref[1].K), (out[1].k, ref[1].k), (out[2], ref[2]), (out[3], ref[3]), (out[4], ref[4])):
        assert relerr(a_, b_) < 1e-9
