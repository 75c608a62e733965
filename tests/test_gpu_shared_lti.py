"""The shared-LTI backward pass (csrc/back_pass_sh.hip): the matrix recursion once per distinct λ, an affine chain per trajectory.
Reference: src/backward_pass.jl:217-252 (LTI method), :28-79 (@end_backward_pass).  Every case is compared with the C oracle on
EVERY trajectory; the per-trajectory kernels (DDP_SH_MIN_B above the batch) give the second opinion."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu
RTOL = 1e-8     # BASELINE.json: Vx/Vxx/L within 1e-8 relative


@pytest.fixture(scope="module")
def ddp():
    import ddp_amd
    return ddp_amd


def _lti(rng, N, B, cxu_scale=0.01):
    import scipy.linalg as sla
    n, m = 10, 2
    A0 = rng.standard_normal((n, n))
    A = sla.expm(0.05 * (A0 - A0.T)) * rng.uniform(0.97, 1.03)
    Bm = 0.1 * rng.standard_normal((n, m))
    def spd(d, s):
        a = rng.standard_normal((d, d)); return s * (a @ a.T / d + 0.5 * np.eye(d))
    cxx, cuu = spd(n, 0.1), spd(m, 0.05)
    cxu = cxu_scale * rng.standard_normal((n, m))
    cx = 0.1 * rng.standard_normal((n, N, B)); cu = 0.1 * rng.standard_normal((m, N, B))
    u = np.zeros((m, N, B))
    return cx, cu, cxx, cxu, cuu, A, Bm, u


def _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, regType, u, who=None):
    from oracle import oracle_ctypes as oc
    div, pol, Vx, Vxx, dV = out
    B = cx.shape[-1]
    lam = np.broadcast_to(np.asarray(lam, float), (B,))
    worst = 0.0
    for b in (range(B) if who is None else who):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx, cxu, cuu, A, Bm, lam[b], regType, None, None, u[..., b])
        assert d == div[b], (b, d, div[b])
        for got, ref, name in ((pol.K[..., b], K, "K"), (pol.k[..., b], k, "k"), (Vx[..., b], vx, "Vx"), (Vxx[..., b], vxx, "Vxx"),
                               (dV[:, b], dv, "dV")):
            e = relerr(got, ref)
            assert e < RTOL, (name, b, e)
            worst = max(worst, e)
        if d == 0:
            assert relerr(pol.Σi[..., b], Quu) < RTOL, ("Quu", b)
        else:
            assert not pol.K[:, :, : d, b].any() and not Vxx[:, :, : d, b].any() and not Vx[:, : d, b].any() and not pol.k[:, : d, b].any()
    return worst


@pytest.mark.parametrize("regType", [1, 2])
@pytest.mark.parametrize("N", [16, 17, 24, 37, 100])
def test_uniform_lambda_every_trajectory(ddp, monkeypatch, regType, N):
    """one λ for the whole batch: one group, tiles of 4 / 8 / 16 trajectories incl. a ragged last tile; N not a multiple of 8"""
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    rng = np.random.default_rng(10 * N + regType)
    B = 37
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    for lam in (0.0, 0.37, 1.0):
        out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, regType, None, None, u)
        assert np.array_equal(out[3], np.transpose(out[3], (1, 0, 2, 3)))            # Vxx exactly symmetric
        _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, regType, u)


def test_matches_the_per_trajectory_kernels(ddp, monkeypatch):
    """the same call through the per-trajectory MFMA-tile kernel: agreement far below the oracle tolerance"""
    rng = np.random.default_rng(5)
    N, B = 200, 64
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    a = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, 0.5, 1, None, None, u)
    monkeypatch.setenv("DDP_SH_MIN_B", "1000000")
    b_ = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, 0.5, 1, None, None, u)
    for x, y in ((a[1].K, b_[1].K), (a[1].k, b_[1].k), (a[2], b_[2]), (a[3], b_[3]), (a[4], b_[4]), (a[1].Σi, b_[1].Σi)):
        assert relerr(x, y) < 1e-11
    assert np.array_equal(a[0], b_[0])


@pytest.mark.parametrize("regType", [1, 2])
def test_lambda_groups_singletons_and_inactive(ddp, monkeypatch, regType):
    """per-trajectory λ drawn from 5 values (groups) + values that occur once (handed to the per-trajectory kernels)"""
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    rng = np.random.default_rng(77 + regType)
    N, B = 61, 150
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    vals = np.array([0.0, 1.0, 1.0 / 1.6, 1.0 / 1.6 ** 3, 1e-6])
    lam = vals[rng.integers(0, len(vals), B)]
    lam[[3, 50, 99]] = [0.123, 4.5, 7e-3]                       # singletons
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, regType, None, None, u)
    _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, regType, u)


def test_all_distinct_lambda_falls_back(ddp, monkeypatch):
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    rng = np.random.default_rng(3)
    N, B = 40, 33
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    lam = 10.0 ** rng.uniform(-3, 0.5, B)
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
    _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, u)


def test_more_distinct_values_than_groups(ddp, monkeypatch):
    """40 distinct λ values with 3 trajectories each: 16 become groups, the rest goes to the per-trajectory kernels"""
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    rng = np.random.default_rng(8)
    N, B = 33, 120
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    lam = np.repeat(10.0 ** np.linspace(-3, 0.3, 40), 3)
    rng.shuffle(lam)
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
    _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, u)


def _lti_failing(rng, N, B, c=0.003):
    """negative-definite cxx: Vxx sinks backwards in time, Quu = cuu + fu'Vxx fu with it, and QuuF = Quu + λI stops being positive
    definite some way below the horizon — λ chooses the failing step"""
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B, cxu_scale=0.0)
    Bm = Bm / np.linalg.norm(Bm, axis=0) * 0.3
    return cx, cu, -c * np.eye(10), cxu, 0.02 * np.eye(2), A, Bm, u


def test_divergence_follows_the_group(ddp, monkeypatch):
    """three groups failing at different steps next to a healthy one: diverge index, zeros at and below the failing step, the steps
    above it as the oracle has them"""
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    rng = np.random.default_rng(12)
    N, B = 77, 41
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti_failing(rng, N, B)
    lam = np.array([0.02, 0.04, 0.06, 50.0])[np.arange(B) % 4]
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
    assert len(set(out[0][:4])) == 4 and out[0][3] == 0 and all(out[0][:3] > 0), out[0][:4]
    _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, u)


def test_divergence_at_every_position_of_a_chunk(ddp, monkeypatch):
    """the failing step walks through the chunks of 8 steps (every slot, chunk boundaries) and the top chunk.  A recursion that is
    about to lose positive definiteness amplifies rounding (the value function is indefinite here): per failing step the λ with the
    smallest amplification — measured on the oracle by perturbing cxx by 1e-12 — is used, and only draws that amplify by less than
    1e5 are compared (the tolerance stays 1e-8)"""
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    N, B = 60, 6
    ds = set()
    for c in (0.003, 0.01):
        cx, cu, cxx, cxu, cuu, A, Bm, u = _lti_failing(np.random.default_rng(13), N, B, c)
        best = {}
        for lam in np.linspace(-0.019, 0.06, 80):
            d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., 0], cu[..., 0], cxx, cxu, cuu, A, Bm, lam, 1, None, None, u[..., 0])
            if d == 0:
                continue
            d2, (K2, _, _), _, vxx2, _ = oc.back_pass(cx[..., 0], cu[..., 0], cxx * (1 + 1e-12), cxu, cuu, A, Bm, lam, 1, None, None, u[..., 0])
            amp = max(relerr(K2, K), relerr(vxx2, vxx)) / 1e-12 if d2 == d else np.inf
            if d not in best or amp < best[d][0]:
                best[d] = (amp, lam)
        for d, (amp, lam) in best.items():
            if amp > 1e5:
                continue
            ds.add(d)
            out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
            assert out[0][0] == d
            _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, u, who=range(2))
    assert len({d % 8 for d in ds}) == 8 and max(ds) >= 57, sorted(ds)


def test_inactive_trajectories_are_untouched(ddp, monkeypatch):
    """the activity mask of the C ABI (device-resident entry): masked trajectories keep what was in the output arrays"""
    import ctypes as C
    from ddp_amd import _lib
    from oracle import oracle_ctypes as oc
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    rng = np.random.default_rng(99)
    n, m, N, B = 10, 2, 48, 50
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    act = (rng.random(B) < 0.7).astype(np.int32)
    lam = np.where(np.arange(B) % 2 == 0, 1.0, 0.25)
    L = _lib.lib(); h = _lib.default_handle()
    d_in = [h.to_device(x) for x in (cx, cu, cxx, cxu, cuu, A, Bm, lam, act)]
    shapes = {"K": (m, n, N, B), "k": (m, N, B), "Quu": (m, m, N, B), "Vx": (n, N, B), "Vxx": (n, n, N, B), "dV": (2, B)}
    d_out = {kk: h.to_device(np.full(sh, 7.0)) for kk, sh in shapes.items()}
    d_div = h.to_device(np.full(B, 7, np.int32))
    desc = _lib.BPDesc(n, m, N, B, 0, 0, 0, 0, 1, 0)
    _lib.check(L.ddp_back_pass_f64_dev(h.raw, C.byref(desc), *d_in[:8], None, None, d_in[8], d_out["K"], d_out["k"], d_out["Quu"],
                                       d_out["Vx"], d_out["Vxx"], d_out["dV"], d_div))
    h.sync()
    got = {kk: h.to_host(d_out[kk], sh) for kk, sh in shapes.items()}
    div = h.to_host(d_div, (B,), np.int32)
    for p_ in d_in + list(d_out.values()) + [d_div]:
        h.free(p_)
    for b in range(B):
        if not act[b]:
            assert all(np.all(got[kk][..., b] == 7.0) for kk in shapes) and div[b] == 7
            continue
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx, cxu, cuu, A, Bm, lam[b], 1, None, None, u[..., b])
        assert d == div[b] == 0
        for g_, r_ in ((got["K"][..., b], K), (got["k"][..., b], k), (got["Vx"][..., b], vx), (got["Vxx"][..., b], vxx), (got["dV"][:, b], dv),
                       (got["Quu"][..., b], Quu)):
            assert relerr(g_, r_) < RTOL


def test_full_size_c2_groups_of_lambda(ddp):
    """BASELINE config 2 at B = 2048 (the default dispatch takes the shared path from 2 048 trajectories): λ from 4 distinct values,
    every trajectory against the oracle (all host cores)"""
    from conftest import par_map
    from oracle import np_restatement as npr
    rng = np.random.default_rng(4321)
    n, m, N, B = 10, 2, 1000, 2048
    P = npr.make_lq_problem(rng)
    cx = 0.01 * rng.standard_normal((n, N, B)); cu = 0.001 * rng.standard_normal((m, N, B))
    u = np.zeros((m, N, B))
    vals = np.array([1.0, 1.0 / 1.6, 1.0 / 1.6 ** 3, 0.0])
    lam = vals[rng.integers(0, 4, B)]
    out = ddp.back_pass(cx, cu, P["Q"], np.zeros((n, m)), P["R"], P["A"], P["B"], lam, 1, None, None, u)
    assert not out[0].any()
    assert np.array_equal(out[3], np.transpose(out[3], (1, 0, 2, 3)))
    par_map(lambda b: _check_all(out, cx, cu, P["Q"], np.zeros((n, m)), P["R"], P["A"], P["B"], lam, 1, u, who=[b]), range(B))


def test_large_batch_grouping_path(ddp):
    """B = 9 000 > 8 192: the grouping kernel parks the table slots in global memory between its passes (sh_group_kernel<false>) instead
    of keeping them in registers; 6 λ groups in random order + singletons; a sample of trajectories (incl. the singletons and the ends of
    the batch) against the oracle, and the whole batch against the per-trajectory kernels (DDP_BACKPASS forces them)"""
    import os
    from ddp_amd import _lib
    rng = np.random.default_rng(2024)
    N, B = 24, 9000
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    vals = np.array([0.0, 1.0, 1.0 / 1.6, 1.6, 1e-6, 2.56])
    lam = vals[rng.integers(0, len(vals), B)]
    single = [0, 4097, 8191, 8192, 8999]
    lam[single] = [0.321, 3.3, 0.047, 9e-4, 0.77]
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
    assert _lib.default_handle().last_kernel(0) == "sh_back_kernel"
    who = sorted(set(single + [1, 2, 63, 64, 1023, 1024, 4096, 8190, 8193, 8998] + list(rng.integers(0, B, 40))))
    _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, u, who=who)
    os.environ["DDP_BACKPASS"] = "x"                              # mx: the per-trajectory kernel
    try:
        _lib.default_handle().raw
        ref = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
        assert _lib.default_handle().last_kernel(0) != "sh_back_kernel"
    finally:
        del os.environ["DDP_BACKPASS"]
        _lib.default_handle().raw
    assert (out[0] == ref[0]).all()
    for a_, b_ in ((out[1].K, ref[1].K), (out[1].k, ref[1].k), (out[2], ref[2]), (out[3], ref[3]), (out[4], ref[4])):
        assert relerr(a_, b_) < 1e-10


@pytest.mark.parametrize("G,B", [(16, 7681), (16, 8000), (16, 15361), (8, 7937), (8, 8100), (2, 8129), (2, 8160), (2, 16300)])
def test_tile_count_at_the_edges_of_a_round(ddp, G, B):
    """B just above k * 32 * (ncu - G) with G equal λ groups: the device then chooses MORE, smaller tiles than one group holding the
    whole batch would get (round 4 sized the work-item list for that single case and overran it — ADVICE r4).  Whole batch against the
    per-trajectory kernels, a sample against the oracle; no tile may have timed out."""
    import os
    from ddp_amd import _lib
    rng = np.random.default_rng(100 * G + B)
    N = 16
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    vals = 1.6 ** -np.arange(G, dtype=float)
    lam = vals[np.arange(B) % G]
    rng.shuffle(lam)
    h = _lib.default_handle()
    t0 = h.sh_timeouts()
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
    assert h.last_kernel(0) == "sh_back_kernel"
    # no retry: a tile that gave up is a defect of the hand-off protocol, and the control block says who waited for what
    assert h.sh_timeouts() == t0, "tiles timed out: %r" % (h.sh_timeout_info(),)
    os.environ["DDP_BACKPASS"] = "x"
    try:
        ref = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
        assert h.last_kernel(0) != "sh_back_kernel"
    finally:
        del os.environ["DDP_BACKPASS"]
        h.raw
    assert (out[0] == ref[0]).all()
    for a_, b_ in ((out[1].K, ref[1].K), (out[1].k, ref[1].k), (out[2], ref[2]), (out[3], ref[3]), (out[4], ref[4]), (out[1].Σi, ref[1].Σi)):
        assert relerr(a_, b_) < 1e-10
    who = sorted({0, 1, B // 2, B - 2, B - 1} | set(int(v) for v in rng.integers(0, B, 12)))
    _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, u, who=who)


def test_a_timed_out_tile_hands_its_trajectories_to_the_per_trajectory_kernels(ddp, monkeypatch):
    """DDP_TEST_SH_ABORT makes every consumer tile give up at its second chunk (what a 4 s time-out does): the results must still be
    those of the oracle — the tile flags its trajectories for the kernels launched behind the shared pass — and the event is counted
    (ddp_sh_timeouts) instead of vanishing (the flag was written and never read)."""
    from ddp_amd import _lib
    rng = np.random.default_rng(77)
    N, B = 40, 70
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    lam = np.where(np.arange(B) % 3 == 0, 0.5, 1.0)
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    h = _lib.default_handle()
    good = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
    t0 = h.sh_timeouts()
    monkeypatch.setenv("DDP_TEST_SH_ABORT", "1")
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
    assert h.sh_timeouts() > t0
    info = h.sh_timeout_info()                                   # ... and says who waited for what (ddp_sh_timeout_info)
    assert info["records"] and any(r["chunk"] == 1 and 0 <= r["group"] < 2 and r["groups"] == 2 and r["waited_ms"] < 4000 for r in info["records"]), info
    monkeypatch.delenv("DDP_TEST_SH_ABORT")
    _check_all(out, cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, u)
    for a_, b_ in ((out[1].K, good[1].K), (out[2], good[2]), (out[3], good[3]), (out[4], good[4])):
        assert relerr(a_, b_) < 1e-10
    again = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)       # and the path is healthy afterwards
    assert np.array_equal(again[3], good[3]) and h.sh_timeouts() == h.sh_timeouts()


@pytest.mark.parametrize("regType", [1, 2])
def test_long_horizon_wide_lambda_range_tight_tolerance(ddp, monkeypatch, regType):
    """The shared chain symmetrises V only once per chunk of 8 steps (SH_SYM = 8) and, for regType 1, uses K'(T + Qux) for K'T + Qux'K
    (backward_pass.jl:71-72 applies ½(V + V') at every step): over the full horizon N = 1 000 and 16 values of λ across nine
    decades — the small ones make the closed loop stiff and the recursion badly conditioned — every output of a sample of trajectories
    agrees with the oracle to 1e-10 (the suite's bar is 1e-8), and Vxx is exactly symmetric."""
    monkeypatch.setenv("DDP_SH_MIN_B", "1")
    rng = np.random.default_rng(300 + regType)
    N, B = 1000, 64
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    vals = 10.0 ** np.linspace(-6, 3, 16)
    lam = vals[np.arange(B) % 16]
    out = ddp.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, regType, None, None, u)
    from ddp_amd import _lib
    assert _lib.default_handle().last_kernel(0) == "sh_back_kernel"
    assert not out[0].any()
    assert np.array_equal(out[3], np.transpose(out[3], (1, 0, 2, 3)))
    from oracle import oracle_ctypes as oc
    worst = 0.0
    for b in range(0, B, 2):                                     # both trajectories of every λ value appear over the two regTypes' seeds
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx, cxu, cuu, A, Bm, lam[b], regType, None, None, u[..., b])
        assert d == 0
        for got, ref in ((out[1].K[..., b], K), (out[1].k[..., b], k), (out[2][..., b], vx), (out[3][..., b], vxx), (out[4][:, b], dv)):
            worst = max(worst, relerr(got, ref))
    assert worst < 1e-10, worst
