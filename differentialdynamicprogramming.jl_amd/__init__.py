"""MI355X-native iLQG hot path behind the call signatures of DifferentialDynamicProgramming.jl.

Host-side mirror of the reference's interface for this path (same names, argument order and error
behaviour), written in Python because no Julia toolchain exists in the build image; the Julia
`@ccall` wrapper a maintainer would use lives in ``julia/DDPAmd.jl``.

    back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u)   src/backward_pass.jl:162-252
    boxQP(H,g,lower,upper,x0)                               src/boxQP.jl:29-188
    forward_pass(traj_new,x0,u,x,α,problem,lims)            src/forward_pass.jl:9-33
    iLQG(problem,x0,u0; lims, ...)                          src/iLQG.jl:143-341
    GaussianPolicy                                          src/iLQG.jl:39-53

All compute runs in libddp_amd.so (HIP, gfx950) through the C ABI of include/ddp_amd.h.  There is no
CPU fallback: without the library or without a GPU every call raises ``DDPError``.

Arrays follow the reference's shapes; an extra trailing axis is the batch of independent
trajectories (``K[m,n,N,B]``), which the reference does not have.
"""
from __future__ import annotations

import ctypes as _C
import time as _time
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import DDPError, Handle, default_handle  # noqa: F401

__all__ = ["GaussianPolicy", "LQProblem", "PendcartProblem", "back_pass", "boxQP", "forward_pass", "iLQG", "print_timing", "mpc_shift", "demo_linear", "demo_pendcart", "demoQP",
           "df", "Handle", "DDPError", "DEFAULT_ALPHA", "WrappedDiff"]

DEFAULT_ALPHA = 10.0 ** np.linspace(0, -3, 11)     # iLQG.jl:145


@dataclass
class GaussianPolicy:
    """src/iLQG.jl:39-53.  ``K[m,n,T]`` (quirk Q19b: the docstring of the reference says n×m),
    ``k[m,T]``, ``Σ`` = Quui (never written by back_pass in the reference — zeros here),
    ``Σi`` = the unregularised Quu."""
    T: int = 0
    n: int = 0
    m: int = 0
    K: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 0)))
    k: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    Σ: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 0)))
    Σi: np.ndarray = field(default_factory=lambda: np.zeros((0, 0, 0)))

    def isempty(self):
        return self.T == self.n == self.m == 0

    def __len__(self):
        return self.T


# ------------------------------------------------------------------ registered problem families
@dataclass
class LQProblem:
    """x⁺ = A x + B u, cost = ½Σ x∘(Qx) + ½Σ u∘(Ru) — the closures of src/demo_linear.jl:30-50.
    A/B may be [n,n]/[n,m] (LTI), [..,N] (LTV) and, with ``dyn_batched``, carry a trailing batch axis."""
    A: np.ndarray
    B: np.ndarray
    Q: np.ndarray
    R: np.ndarray
    dyn_batched: bool = False
    kind = 0

    @property
    def n(self):
        return self.A.shape[0]

    @property
    def m(self):
        return self.B.shape[1]

    @property
    def dyn_tv(self):
        return (self.A.ndim - (1 if self.dyn_batched else 0)) == 3


@dataclass
class PendcartProblem:
    """src/system_pendcart.jl:42-59,83-106 (explicit Euler step, quadratic cost of length N+1)."""
    g: float = 9.82
    l: float = 0.35
    h: float = 0.01
    d: float = 0.99
    Q: np.ndarray = field(default_factory=lambda: np.diag([10.0, 1.0, 2.0, 1.0]))
    R: np.ndarray = field(default_factory=lambda: np.array([[1.0]]))
    goal: np.ndarray = field(default_factory=lambda: np.array([np.pi, 0.0, 0.0, 0.0]))
    kind = 1
    n = 4
    m = 1
    dyn_tv = False
    dyn_batched = False


class WrappedDiff:
    """What stands in for a user ``diff_fun`` (forward_pass.jl:5,19; iLQG.jl:156): subtraction with the listed state coordinates
    (0-based) wrapped to [-π, π], i.e. ``rem2pi(a[j] - b[j], RoundNearest)`` — ddp_problem::diff_wrap.  ``None`` / ``np.subtract``
    mean the reference's default ``-``.  A Python closure cannot run on the device."""

    def __init__(self, *coords):
        self.coords = tuple(int(c) for c in coords)
        if any(c < 0 or c >= 32 for c in self.coords):
            raise ValueError("WrappedDiff: coordinates must be in 0..31")

    @property
    def mask(self):
        m = 0
        for c in self.coords:
            m |= 1 << c
        return m

    def __call__(self, a, b):                                   # the same function on host arrays
        d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
        for c in self.coords:
            d[c] = np.remainder(d[c] + np.pi, 2 * np.pi) - np.pi
        return d


def _diff_mask(diff, n):
    if diff is None or diff is np.subtract:
        return 0
    if isinstance(diff, WrappedDiff):
        if any(c >= n for c in diff.coords):
            raise ValueError("WrappedDiff names coordinate %d of a state of length %d" % (max(diff.coords), n))
        return diff.mask
    raise TypeError("diff_fun must be None / np.subtract (the reference's `-`) or a WrappedDiff: a Python closure cannot run on the device")


class _DevProblem:
    """ddp_problem struct + the arrays it points at (host arrays for the host-pointer flavours)."""

    def __init__(self, prob, N, B, diff=None):
        P = _lib.Problem()
        P.diff_wrap = _diff_mask(diff, prob.n)
        P.kind, P.n, P.m, P.N, P.B = prob.kind, prob.n, prob.m, N, B
        self.Q, self.R = _lib.f64(prob.Q), _lib.f64(np.atleast_2d(prob.R))
        P.Q, P.R = _lib.ptr(self.Q), _lib.ptr(self.R)
        if prob.kind == 0:
            self.A, self.B = _lib.f64(prob.A), _lib.f64(prob.B)
            P.A, P.Bm = _lib.ptr(self.A), _lib.ptr(self.B)
            P.dyn_tv, P.dyn_batched = int(prob.dyn_tv), int(prob.dyn_batched)
        else:
            P.g, P.l, P.h, P.d = prob.g, prob.l, prob.h, prob.d
            for i in range(4):
                P.goal[i] = float(prob.goal[i])
        # Q and R diagonal (the reference's demos): the rollout kernels then evaluate the cost themselves (ddp_problem::cost_diag)
        P.cost_diag = int(not np.any(self.Q - np.diag(np.diag(self.Q))) and not np.any(self.R - np.diag(np.diag(self.R))))
        self.struct = P
        self.cost_len = N + 1 if prob.kind == 1 else N


def _lims(lims):
    if lims is None or np.size(lims) == 0:
        return None
    return _lib.f64(lims)


# ---------------------------------------------------------------------------------- back_pass
def back_pass(cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u, *, fx_batched=None, cost_batched=None,
              batched_dynamics=None, batched_cost=None, handle=None):
    """Drop-in for ``back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u)`` (backward_pass.jl:217).

    Dispatch on array rank like the reference (``fx`` 2-D → LTI :217, 3-D → LTV :162, ``cxx`` 3-D →
    time-varying cost :179).  ``cx`` of rank 3 means a batch ``cx[n,N,B]``; then ``λ`` may be a vector
    of length B and ``fx``/``cxx`` of rank 4 are per-trajectory.
    Returns ``(diverge, GaussianPolicy, Vx, Vxx, dV)``; with a batch every output carries a trailing
    batch axis and ``diverge`` is an int32 vector.

    Per-trajectory TIME-INVARIANT operands (``fx[n,n,B]``, ``cxx[n,n,B]``) have the rank of the reference's time-varying ones and are
    never guessed from ``shape[2] == B``: pass ``batched_dynamics=True`` / ``batched_cost=True`` (``DDPAmd.back_pass`` has the same
    keywords; ``fx_batched`` / ``cost_batched`` are the older names of the same switches)."""
    h = handle or default_handle()
    if batched_dynamics is not None:
        assert fx_batched is None or bool(fx_batched) == bool(batched_dynamics), "fx_batched and batched_dynamics disagree"
        fx_batched = bool(batched_dynamics)
    if batched_cost is not None:
        assert cost_batched is None or bool(cost_batched) == bool(batched_cost), "cost_batched and batched_cost disagree"
        cost_batched = bool(batched_cost)
    cx, cu, u = _lib.f64(cx), _lib.f64(cu), _lib.f64(u)
    batched = cx.ndim == 3
    n, N = cx.shape[0], cx.shape[1]
    m = cu.shape[0]
    B = cx.shape[2] if batched else 1
    fx, fu, cxx, cxu, cuu = map(_lib.f64, (fx, fu, cxx, np.reshape(cxu, np.shape(cxu)), cuu))
    if fx_batched is None:
        fx_batched = fx.ndim == 4
    if cost_batched is None:
        cost_batched = cxx.ndim == 4
    fx_tv = (fx.ndim - int(fx_batched)) == 3
    cost_tv = (cxx.ndim - int(cost_batched)) == 3
    # the reference's @asserts (backward_pass.jl:221-225 / :8-11 / :183-187)
    assert cu.shape[:2] == (m, N), "size(cu) should be (m, N)"
    assert fx.shape[:2] == (n, n) and fu.shape[:2] == (n, m), "size(fx), size(fu)"
    assert cxx.shape[:2] == (n, n), "size(cxx) should be (n, n)"
    assert cxu.shape[:2] == (n, m), "size(cxu) should be (n, m)"
    assert cuu.reshape((m, m) + cuu.shape[2:] if cuu.ndim >= 2 else (m, m)).shape[:2] == (m, m), "size(cuu)"
    cuu = cuu.reshape((m, m) + (cuu.shape[2:] if cuu.ndim >= 2 else ()))
    # full extents (the reference asserts size(cxx) == (n,n,N) etc.; the C side would read out of bounds otherwise)
    tail_f = ((N,) if fx_tv else ()) + ((B,) if fx_batched else ())
    tail_c = ((N,) if cost_tv else ()) + ((B,) if cost_batched else ())
    assert cu.shape == (m, N) + ((B,) if batched else ()), "size(cu) should be (m, N[, B])"
    assert u.shape == cu.shape, "size(u) should be (m, N[, B])"
    assert fx.shape == (n, n) + tail_f and fu.shape == (n, m) + tail_f, "size(fx), size(fu): time / batch extents"
    assert cxx.shape == (n, n) + tail_c and cxu.shape == (n, m) + tail_c and cuu.shape == (m, m) + tail_c, "size(cxx), size(cxu), size(cuu): time / batch extents"
    L = _lims(lims)
    assert L is None or L.shape == (m, 2), "lims should be (m, 2)"
    d = _lib.BPDesc(n, m, N, B, int(fx_tv), int(fx_batched), int(cost_tv), int(cost_batched), int(regType),
                    int(L is not None))
    lam = np.ascontiguousarray(np.broadcast_to(np.asarray(λ, dtype=np.float64), (B,)))
    K = _lib.result_array((m, n, N, B)); k = _lib.result_array((m, N, B))
    Quu = _lib.result_array((m, m, N, B)); Vx = _lib.result_array((n, N, B))
    Vxx = _lib.result_array((n, n, N, B)); dV = np.zeros((2, B), order="F")
    div = np.zeros(B, dtype=np.int32)
    _lib.check(_lib.lib().ddp_back_pass_f64(h.raw, _C.byref(d), *map(_lib.ptr, (cx, cu, cxx, cxu, cuu, fx, fu, lam, L, u,
                                                                              K, k, Quu, Vx, Vxx, dV)),
                                            div.ctypes.data_as(_lib.i32p)))
    if not batched:
        pol = GaussianPolicy(N, n, m, K[..., 0], k[..., 0], np.zeros((m, m, N)), Quu[..., 0])
        return int(div[0]), pol, Vx[..., 0], Vxx[..., 0], dV[:, 0]
    pol = GaussianPolicy(N, n, m, K, k, np.zeros((m, m, N, B)), Quu)
    return div, pol, Vx, Vxx, dV


# -------------------------------------------------------------------------------------- boxQP
def boxQP(H, g, lower, upper, x0, *, maxIter=100, minGrad=1e-8, minRelImprove=1e-8, stepDec=0.6, minStep=1e-22,
          Armijo=0.1, handle=None):
    """Drop-in for ``boxQP(H,g,lower,upper,x0; ...)`` (boxQP.jl:29-36): returns ``(x, result, Hfree, free)``.
    ``H`` of rank 3 / vectors of rank 2 solve a batch (trailing axis)."""
    h = handle or default_handle()
    H, g, lower, upper, x0 = map(_lib.f64, (H, g, lower, upper, x0))
    batched = H.ndim == 3
    m = H.shape[0]
    cnt = H.shape[2] if batched else 1
    x = np.zeros((m, cnt), order="F"); Hf = np.zeros((m, m, cnt), order="F")
    res = np.zeros(cnt, dtype=np.int32); fr = np.zeros((m, cnt), dtype=np.uint8, order="F")
    o = _lib.QPOpts(maxIter, minGrad, minRelImprove, stepDec, minStep, Armijo)
    _lib.check(_lib.lib().ddp_boxqp_f64(h.raw, m, cnt, *map(_lib.ptr, (H, g, lower, upper, x0)), _C.byref(o),
                                        _lib.ptr(x), res.ctypes.data_as(_lib.i32p), _lib.ptr(Hf),
                                        fr.ctypes.data_as(_lib.u8p)))
    if not batched:
        free = fr[:, 0].astype(bool)
        nf = int(free.sum())
        return x[:, 0], int(res[0]), Hf[:nf, :nf, 0], free
    return x, res, Hf, fr.astype(bool)


def demoQP(*, n=500, rng=None, **kwargs):
    """``demoQP(;kwargs...)`` (src/boxQP.jl:190-199): ``H = M·M'`` with ``M = randn(n,n)``, ``g = randn(n)``, bounds ±1, a random
    start; returns ``boxQP``'s tuple and the wall time of the call in seconds."""
    import time as _time
    rng = rng if rng is not None else np.random.default_rng()
    M = rng.standard_normal((n, n))
    H, g = M @ M.T, rng.standard_normal(n)
    t0 = _time.perf_counter()
    out = boxQP(H, g, -np.ones(n), np.ones(n), rng.standard_normal(n), **kwargs)
    return out + (_time.perf_counter() - t0,)


# ------------------------------------------------------------------------------- demos (problem generators + solver settings)
def demo_linear(*, B=None, rng=None, T=1000, n=10, m=2, h=0.01, **kwargs):
    """``demo_linear(;kwargs...)`` (src/demo_linear.jl:5-60): random stable LTI system ``A = exp(h(A0 - A0'))``, ``B = h·randn``,
    ``Q = h·I``, ``R = 0.1h·I``, ``x0 = ones(n)``, ``u0 = 0.1·randn(m,T)``, no control limits; runs ``iLQG`` with the given keyword
    arguments and returns its tuple.  ``B`` solves a batch of independent problems (same system, own ``u0``) — the reference
    runs one.  NumPy's generator replaces Julia's (the draws differ, the distribution does not)."""
    import scipy.linalg as _sla
    rng = rng if rng is not None else np.random.default_rng()
    A0 = rng.standard_normal((n, n))
    A = _sla.expm(h * (A0 - A0.T))                             # skew-symmetric generator: pure imaginary eigenvalues
    Bm = h * rng.standard_normal((n, m))
    Q, R = h * np.eye(n), 0.1 * h * np.eye(m)
    x0 = np.ones(n) if B is None else np.ones((n, B))
    u0 = 0.1 * rng.standard_normal((m, T) if B is None else (m, T, B))
    return iLQG(LQProblem(A, Bm, Q, R), x0, u0, **kwargs)


def demo_pendcart(*, x0=(np.pi - 0.6, 0.0, 0.0, 0.0), goal=(np.pi, 0.0, 0.0, 0.0), Q=None, R=1.0, lims=None, T=600, B=None, **kwargs):
    """``demo_pendcart(;x0, goal, Q, R, lims, T)`` (src/system_pendcart.jl:42-212) with the reference's solver settings
    (``regType=2, α=exp10.(range(0.2,stop=-3,length=6)), λmax=1e15, tol_fun=tol_grad=1e-8, max_iter=1000``) from ``u0 = 0``
    (the reference passes ``0*u00``: its LQR simulation only feeds the comparison plot).  ``B`` replicates the problem."""
    Q = np.diag([10.0, 1.0, 2.0, 1.0]) if Q is None else np.asarray(Q, dtype=np.float64)
    lims = 5.0 * np.array([[-1.0, 1.0]]) if lims is None else lims
    prob = PendcartProblem(Q=Q, R=np.array([[float(R)]]), goal=np.asarray(goal, dtype=np.float64))
    x0 = np.asarray(x0, dtype=np.float64)
    if B is not None:
        x0 = np.tile(x0[:, None], (1, B))
    u0 = np.zeros((1, T) if B is None else (1, T, B))
    kw = dict(regType=2, α=10.0 ** np.linspace(0.2, -3, 6), λmax=1e15, tol_fun=1e-8, tol_grad=1e-8, max_iter=1000)
    kw.update(kwargs)
    return iLQG(prob, x0, u0, lims=lims, **kw)


# ------------------------------------------------------------------------------- MPC warm start
def mpc_shift(a, shift=1, *, zero_tail=False, batched=None, handle=None):
    """Receding-horizon shift of a time-major array ``a[...,N(,B)]`` (controls, nominal states, gains; ``batched`` says whether the
    last axis is the batch — default: yes for more than two axes) by ``shift`` steps through
    ``ddp_mpc_shift_f64_dev``: ``out[:,i] = a[:,i+shift]``, the vacated tail repeats the last column (or is zero).  New — the
    reference has no MPC loop; its hook is the pre-rolled ``x0[n,N]`` + ``cost`` warm start (iLQG.jl:193-197)."""
    h = handle or default_handle()
    a = _lib.f64(a)
    if batched is None:
        batched = a.ndim > 2
    lead = a.shape[:-2] if batched else a.shape[:-1]
    N, B = (a.shape[-2], a.shape[-1]) if batched else (a.shape[-1], 1)
    d = int(np.prod(lead))
    src = h.to_device(a)
    dst = h.malloc(a.nbytes)
    try:
        _lib.check(_lib.lib().ddp_mpc_shift_f64_dev(h.raw, d, N, B, int(shift), int(bool(zero_tail)), src, dst))
        out = h.to_host(dst, a.shape)
    finally:
        h.free(src); h.free(dst)
    return out


def _check_problem(problem, n, m, N, B):
    """extents of a registered family's arrays against the call's sizes (the C side trusts them)"""
    if problem.kind == 1:
        if (n, m) != (4, 1):
            raise ValueError("PendcartProblem has n = 4, m = 1")
        return
    tail = ((N,) if problem.dyn_tv else ()) + ((B,) if problem.dyn_batched else ())
    A, Bm = np.asarray(problem.A), np.asarray(problem.B)
    if A.shape != (n, n) + tail or Bm.shape != (n, m) + tail:
        raise ValueError("LQProblem: A should be (n,n%s), B (n,m%s)" % ((",".join([""] + ["N"] * problem.dyn_tv + ["B"] * problem.dyn_batched),) * 2))
    if np.shape(problem.Q) != (n, n) or np.shape(np.atleast_2d(problem.R)) != (m, m):
        raise ValueError("LQProblem: Q should be (n,n), R (m,m)")


# ------------------------------------------------------------------------------- forward_pass
def forward_pass(traj_new, x0, u, x, α, problem, lims, diff=None, *, handle=None):
    """Drop-in for ``forward_pass(traj_new,x0,u,x,α,f,costfun,lims,diff)`` (forward_pass.jl:9) with a
    registered ``problem`` standing in for the closures ``f``/``costfun``; ``diff``: ``None`` (``-``) or a ``WrappedDiff``.
    ``traj_new`` may be an empty ``GaussianPolicy`` (then ``x`` is ignored, iLQG.jl:185).
    A vector ``α`` rolls all step sizes out concurrently (outputs get a trailing α axis).
    Returns ``(xnew, unew, cnew)``."""
    h = handle or default_handle()
    u, x0 = _lib.f64(u), _lib.f64(x0)
    batched = u.ndim == 3
    m, N = u.shape[:2]
    n = x0.shape[0]
    B = u.shape[2] if batched else 1
    dp = _DevProblem(problem, N, B, diff)
    if x0.shape != ((n, B) if batched else (n,)) and not (not batched and x0.shape == (n, 1)):
        raise ValueError("x0 should be (n,) — (n, B) with a batched u")
    _check_problem(problem, n, m, N, B)
    alphas = np.atleast_1d(np.asarray(α, dtype=np.float64))
    na = len(alphas)
    empty = traj_new is None or traj_new.isempty()
    K = None if empty else _lib.f64(traj_new.K)
    k = None if empty else _lib.f64(traj_new.k)
    xx = None if empty else _lib.f64(x)
    if not empty:
        tb = (B,) if batched else ()
        if K.shape != (m, n, N) + tb or k.shape != (m, N) + tb or xx.shape != (n, N) + tb:
            raise ValueError("traj_new.K, traj_new.k, x should be (m,n,N), (m,N), (n,N) [+ batch axis]")
    L = _lims(lims)
    CL = dp.cost_len
    xnew = _lib.result_array((n, N, B, na)); unew = _lib.result_array((m, N, B, na))
    cnew = _lib.result_array((CL, B, na)); csum = np.zeros((B, na), order="F")
    _lib.check(_lib.lib().ddp_forward_pass_f64(h.raw, _C.byref(dp.struct), _lib.ptr(K), _lib.ptr(k), _lib.ptr(x0),
                                               _lib.ptr(u), _lib.ptr(xx), _lib.ptr(alphas), na, _lib.ptr(L),
                                               _lib.ptr(xnew), _lib.ptr(unew), _lib.ptr(cnew), _lib.ptr(csum)))
    if not batched:
        xnew, unew, cnew = xnew[:, :, 0], unew[:, :, 0], cnew[:, 0]
    if np.ndim(α) == 0:
        xnew, unew, cnew = xnew[..., 0], unew[..., 0], cnew[..., 0]
    return xnew, unew, cnew


# ------------------------------------------------------------------------------------------ df
def df(problem, x, u, *, handle=None):
    """The ``df`` closure of the registered families (STEP 1, iLQG.jl:225-229).  Returns
    ``(fx,fu,fxx,fxu,fuu,cx,cu,cxx,cxu,cuu)`` like the reference (second-order terms are ``[]``)."""
    h = handle or default_handle()
    x, u = _lib.f64(x), _lib.f64(u)
    batched = u.ndim == 3
    m, N = u.shape[:2]
    n = x.shape[0]
    B = u.shape[2] if batched else 1
    dp = _DevProblem(problem, N, B)
    cx = _lib.result_array((n, N, B)); cu = _lib.result_array((m, N, B))
    pend = problem.kind == 1
    fx = _lib.result_array((n, n, N, B)) if pend else None
    fu = _lib.result_array((n, m, N, B)) if pend else None
    _lib.check(_lib.lib().ddp_df_f64(h.raw, _C.byref(dp.struct), _lib.ptr(x), _lib.ptr(u), _lib.ptr(cx), _lib.ptr(cu),
                                     _lib.ptr(fx), _lib.ptr(fu)))
    if not pend:
        fx, fu = dp.A, dp.B
    elif not batched:
        fx, fu = fx[..., 0], fu[..., 0]
    if not batched:
        cx, cu = cx[..., 0], cu[..., 0]
    e = np.zeros((0,))
    return fx, fu, e, e, e, cx, cu, dp.Q, np.zeros((n, m)), dp.R


# ---------------------------------------------------------------------------------------- iLQG
def print_timing(trace):
    """The timing summary of iLQG.jl:343-366 from the trace keys ``time_derivs``, ``time_backward``, ``time_forward``."""
    if "time_derivs" not in trace:
        raise KeyError("print_timing needs the time_* keys: call iLQG(..., timing=True)")
    parts = [float(np.nansum(trace[k])) for k in ("time_derivs", "time_backward", "time_forward")]
    total = float(trace.get("time_total", sum(parts)))
    it = max(int(trace.get("global_iters", 1)), 1)
    pct = [100.0 * t / total for t in parts] + [100.0 * (total - sum(parts)) / total]
    print("\n iterations:   %-3d\n time / iter:  %-5.2f ms\n total time:   %-5.3f seconds, of which\n derivs:     %-4.1f%%\n"
          " back pass:  %-4.1f%%\n fwd pass:   %-4.1f%%\n other:      %-4.1f%% (host, transfers)\n =========== end iLQG ==========="
          % (it, 1e3 * total / it, total, pct[0], pct[1], pct[2], pct[3]))


STATUS = {1: "SUCCESS: gradient norm < tol_grad", 2: "SUCCESS: cost change < tol_fun", 3: "EXIT: λ > λmax",
          4: "EXIT: Maximum iterations reached", -1: "EXIT: Initial control sequence caused divergence",
          5: "EXIT: the driver's bound on batch iterations ran out (not a state of the reference)"}


def iLQG(problem, x0, u0, *, lims=None, α=DEFAULT_ALPHA, tol_fun=1e-7, tol_grad=1e-4, max_iter=500, λ=1.0, dλ=1.0,
         λfactor=1.6, λmax=1e10, λmin=1e-6, regType=1, reduce_ratio_min=0.0, verbosity=0, trace_cap=None, cost=None,
         timing=True, diff_fun=None, handle=None):
    """Drop-in for ``iLQG(f,costfun,df,x0,u0; lims, α, tol_fun, ...)`` (iLQG.jl:143-163) with a registered
    ``problem`` standing in for the three closures (``diff_fun``: ``None`` = ``-``, or a ``WrappedDiff``).  ``u0[m,N,B]`` / ``x0[n,B]`` solve a batch of
    independent problems, each with its own λ schedule, line search and termination.
    Returns ``(x, u, traj_new, Vx, Vxx, cost, trace)``; ``trace`` is a dict with the reference's trace
    keys that survive batching (``:cost`` per iteration) plus the per-trajectory summary ``stats``.
    ``x0[n,N]`` (``x0[n,N,B]`` with a batch) is a PRE-ROLLED initial trajectory (iLQG.jl:193-197: no initial rollout;
    ``cost`` as given or ``costfun(x0,u0)``) — the warm start of an MPC loop.
    ``timing=False`` drops the ``time_*`` trace keys: the driver then synchronises with the host every fourth batch iteration only.
    ``trace_cap``: rows of the per-iteration history kept per trajectory; the default shrinks with the batch,
    ``min(4 max_iter + 64, 4096, max(64, 256e6 / (56 B)))`` (136 rows at B = 32768), so that ``history[7, cap, B]`` stays under 256 MB.
    ``trace["trace_cap"]`` is the cap used and ``trace["truncated"]`` says, per trajectory, whether it took more iterations than rows
    were kept (its later rows are missing, ``stats`` / ``iter`` are complete).
    Returns ``None`` when the initial control sequence diverges (iLQG.jl:205-210) in the unbatched case."""
    h = handle or default_handle()
    u0, x0 = _lib.f64(u0), _lib.f64(x0)
    batched = u0.ndim == 3
    m, N = u0.shape[:2]
    n = x0.shape[0]
    B = u0.shape[2] if batched else 1
    prerolled = False
    if x0.ndim == (3 if batched else 2):                      # x0 has a time axis: single column or pre-rolled (iLQG.jl:181,193)
        if x0.shape[1] == N:
            prerolled = True
        elif x0.shape[1] == 1:
            x0 = _lib.f64(x0[:, 0])
        else:
            raise ValueError("pre-rolled initial trajectory must be of correct length (size(x0,2) == N)")     # iLQG.jl:199
    if x0.shape != (((n, N) if prerolled else (n,)) + ((B,) if batched else ())):
        raise ValueError("x0 should be (n,) / pre-rolled (n, N) — with a batched u0: (n, B) / (n, N, B)")
    _check_problem(problem, n, m, N, B)
    dp = _DevProblem(problem, N, B, diff_fun)
    o = _lib.ILQGOpts()
    _lib.lib().ddp_ilqg_default_opts(_C.byref(o))
    o.lambda_, o.dlambda, o.lambda_factor, o.lambda_max, o.lambda_min = λ, dλ, λfactor, λmax, λmin
    o.tol_fun, o.tol_grad, o.max_iter, o.regType, o.reduce_ratio_min = tol_fun, tol_grad, max_iter, regType, reduce_ratio_min
    alphas = np.asarray(α, dtype=np.float64)
    if len(alphas) > 16:
        raise ValueError("at most 16 line-search step sizes are supported")
    o.n_alpha = len(alphas)
    for i, a in enumerate(alphas):
        o.alpha[i] = a
    L = _lims(lims)
    CL = dp.cost_len
    x = _lib.result_array((n, N, B)); u = _lib.result_array((m, N, B))
    K = _lib.result_array((m, n, N, B)); k = _lib.result_array((m, N, B)); Quu = _lib.result_array((m, m, N, B))
    Vx = _lib.result_array((n, N, B)); Vxx = _lib.result_array((n, n, N, B))
    stats = np.zeros((8, B), order="F")
    # rows of the per-iteration trace kept per trajectory; by default bounded so that trace7[7, cap, B] stays under 256 MB
    # (a batch of 4096 pendcart solves with cap = 4 max_iter + 64 moved 0.93 GB of mostly zeros)
    cap = trace_cap if trace_cap is not None else min(4 * max_iter + 64, 4096, max(64, int(256e6 / (56 * B))))
    cap = min(cap, 4096)
    git = _C.c_int(0)
    c0 = None if (not prerolled or cost is None or np.size(cost) == 0) else _lib.f64(np.reshape(cost, (CL, B), order="F"))
    cost = _lib.result_array((CL, B))
    tr7 = _lib.result_array((7, cap, B))
    tcap = 4 * max_iter + 1000                                 # the driver's bound on global iterations
    timing_on = bool(timing)
    timing = np.full((3, tcap), np.nan)
    t_start = _time.time()
    _lib.check(_lib.lib().ddp_ilqg_set_timing(h.raw, _lib.ptr(timing) if timing_on else None, tcap if timing_on else 0))
    try:
        _lib.check(_lib.lib().ddp_ilqg_ex_f64(h.raw, _C.byref(dp.struct), _C.byref(o), _lib.ptr(x0), int(prerolled), _lib.ptr(u0),
                                              _lib.ptr(c0), _lib.ptr(L), *map(_lib.ptr, (x, u, K, k, Quu, Vx, Vxx, cost, stats)), cap,
                                              _lib.ptr(tr7), _C.byref(git)))
    finally:
        _lib.lib().ddp_ilqg_set_timing(h.raw, None, 0)
    total_t = _time.time() - t_start
    tr = tr7[4]
    # the reference's trace keys (iLQG.jl:257,325-330), one row per iteration: trace["history"][key][iteration-1(, b)]
    hist = {key: tr7[c] for c, key in enumerate(("λ", "dλ", "α", "improvement", "cost", "reduce_ratio", "grad_norm"))}
    trace = dict(stats=stats, status=stats[0].astype(int), iter=stats[1].astype(int), λ=stats[5], grad_norm=stats[6],
                 cost=tr, global_iters=git.value, history=hist, trace_cap=cap, truncated=(stats[1] - 1 > cap))
    if trace["truncated"].any():
        import warnings
        warnings.warn("iLQG: %d of %d trajectories took more iterations than the %d history rows kept (trace_cap); their later rows are "
                      "missing from trace['history'] / trace['cost']" % (int(trace["truncated"].sum()), B, cap))
    # time_derivs / time_backward / time_forward (iLQG.jl:227,241,281): GPU seconds per GLOBAL iteration of the batch
    for r, key in enumerate(("time_derivs", "time_backward", "time_forward")):
        if timing_on:
            trace[key] = timing[r, : git.value].copy()
    trace["time_total"] = total_t
    if verbosity > 0:
        for b in range(min(B, 8)):
            print("[%d] %s after %d iterations, cost %.6g" % (b, STATUS.get(int(stats[0, b]), "?"), int(stats[1, b]), stats[7, b]))
        if timing_on:
            print_timing(trace)
    if not batched:
        if int(stats[0, 0]) == -1:
            return None
        it = int(stats[1, 0])
        trace["cost"] = tr[: max(it - 1, 0), 0]
        trace["history"] = {key: v[: max(it - 1, 0), 0] for key, v in hist.items()}
        pol = GaussianPolicy(N, n, m, K[..., 0], k[..., 0], np.zeros((m, m, N)), Quu[..., 0])
        return x[..., 0], u[..., 0], pol, Vx[..., 0], Vxx[..., 0], cost[:, 0], trace
    pol = GaussianPolicy(N, n, m, K, k, np.zeros((m, m, N, B)), Quu)
    return x, u, pol, Vx, Vxx, cost, trace


def _ilqg_opts(α, tol_fun, tol_grad, max_iter, λ, dλ, λfactor, λmax, λmin, regType, reduce_ratio_min):
    o = _lib.ILQGOpts()
    _lib.lib().ddp_ilqg_default_opts(_C.byref(o))
    o.lambda_, o.dlambda, o.lambda_factor, o.lambda_max, o.lambda_min = λ, dλ, λfactor, λmax, λmin
    o.tol_fun, o.tol_grad, o.max_iter, o.regType, o.reduce_ratio_min = tol_fun, tol_grad, max_iter, regType, reduce_ratio_min
    alphas = np.asarray(α, dtype=np.float64)
    if len(alphas) > 16:
        raise ValueError("at most 16 line-search step sizes are supported")
    o.n_alpha = len(alphas)
    for i, a in enumerate(alphas):
        o.alpha[i] = a
    return o


def iLQG_queue(problem, x0, u0, *, slots=0, lims=None, α=DEFAULT_ALPHA, tol_fun=1e-7, tol_grad=1e-4, max_iter=500, λ=1.0, dλ=1.0,
               λfactor=1.6, λmax=1e10, λmin=1e-6, regType=1, reduce_ratio_min=0.0, diff_fun=None, handle=None):
    """``P = u0.shape[2]`` independent solves of ``iLQG`` through ``slots`` resident trajectories (``ddp_ilqg_queue_f64``): a slot whose
    solve has ended is flushed and armed with the next problem on the device, instead of idling until the slowest trajectory of a
    lock-step batch has ended.  Every solve is the solve ``iLQG`` performs at batch size ``slots``.
    Returns ``(x, u, traj_new, Vx, Vxx, cost, trace)`` with P columns; ``trace`` holds ``stats[8,P]``, ``status``, ``iter``, ``global_iters``."""
    h = handle or default_handle()
    u0, x0 = _lib.f64(u0), _lib.f64(x0)
    if u0.ndim != 3 or x0.ndim != 2:
        raise ValueError("iLQG_queue: x0[n,P], u0[m,N,P]")
    m, N, P = u0.shape
    n = x0.shape[0]
    _check_problem(problem, n, m, N, P)
    dp = _DevProblem(problem, N, P, diff_fun)
    o = _ilqg_opts(α, tol_fun, tol_grad, max_iter, λ, dλ, λfactor, λmax, λmin, regType, reduce_ratio_min)
    L = _lims(lims)
    CL = dp.cost_len
    x = _lib.result_array((n, N, P)); u = _lib.result_array((m, N, P))
    K = _lib.result_array((m, n, N, P)); k = _lib.result_array((m, N, P)); Quu = _lib.result_array((m, m, N, P))
    Vx = _lib.result_array((n, N, P)); Vxx = _lib.result_array((n, n, N, P)); cost = _lib.result_array((CL, P))
    stats = np.zeros((8, P), order="F")
    git = _C.c_int(0)
    t0 = _time.time()
    _lib.check(_lib.lib().ddp_ilqg_queue_f64(h.raw, _C.byref(dp.struct), _C.byref(o), int(slots), _lib.ptr(x0), _lib.ptr(u0), _lib.ptr(L),
                                             *map(_lib.ptr, (x, u, K, k, Quu, Vx, Vxx, cost, stats)), _C.byref(git)))
    trace = dict(stats=stats, status=stats[0].astype(int), iter=stats[1].astype(int), λ=stats[5], grad_norm=stats[6],
                 global_iters=git.value, time_total=_time.time() - t0)
    return x, u, GaussianPolicy(N, n, m, K, k, np.zeros((m, m, N, P)), Quu), Vx, Vxx, cost, trace


def iLQG_mpc(problem, x0, u0, steps, *, zero_tail=False, lims=None, α=DEFAULT_ALPHA, tol_fun=1e-7, tol_grad=1e-4, max_iter=500, λ=1.0,
             dλ=1.0, λfactor=1.6, λmax=1e10, λmin=1e-6, regType=1, reduce_ratio_min=0.0, diff_fun=None, handle=None):
    """Closed loop on the device (``ddp_ilqg_mpc_f64``): every trajectory of the batch is solved ``steps`` times; after each solve the
    first control is applied (model = plant: the next initial state is ``x[:,1]`` of the solution), the control sequence is shifted by
    one step (``mpc_shift``) and the problem is solved again without returning to the host.
    Returns ``(xcl[n,steps+1,B], ucl[m,steps,B], stats[8,steps,B], x_plan[n,N,B], u_plan[m,N,B], global_iters)``."""
    h = handle or default_handle()
    u0, x0 = _lib.f64(u0), _lib.f64(x0)
    if u0.ndim != 3 or x0.ndim != 2:
        raise ValueError("iLQG_mpc: x0[n,B], u0[m,N,B]")
    m, N, B = u0.shape
    n = x0.shape[0]
    _check_problem(problem, n, m, N, B)
    dp = _DevProblem(problem, N, B, diff_fun)
    o = _ilqg_opts(α, tol_fun, tol_grad, max_iter, λ, dλ, λfactor, λmax, λmin, regType, reduce_ratio_min)
    L = _lims(lims)
    steps = int(steps)
    xcl = np.zeros((n, steps + 1, B), order="F"); ucl = np.zeros((m, steps, B), order="F"); scl = np.zeros((8, steps, B), order="F")
    x = _lib.result_array((n, N, B)); u = _lib.result_array((m, N, B))
    git = _C.c_int(0)
    _lib.check(_lib.lib().ddp_ilqg_mpc_f64(h.raw, _C.byref(dp.struct), _C.byref(o), steps, int(bool(zero_tail)), _lib.ptr(x0), _lib.ptr(u0),
                                           _lib.ptr(L), *map(_lib.ptr, (xcl, ucl, scl, x, u)), _C.byref(git)))
    return xcl, ucl, scl, x, u, git.value


def runtime_info():
    """which HIP runtime the process ended up with and why (``_lib._share_torch_hip``), the library version, the devices it sees"""
    L = _lib.lib()
    return dict(version=L.ddp_version().decode(), devices=int(L.ddp_device_count()), hip_runtime=_lib.hip_runtime_note)
