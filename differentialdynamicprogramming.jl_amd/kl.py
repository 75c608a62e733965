"""KL-constrained path (BASELINE config 5) behind the reference's call signatures.

    ∇kl(traj_prev)                                                   src/klutils.jl:8-23
    back_pass_gps(cx,cu,cxx,cxu,cuu,fx,fu,lims,x,u,kl_cost_terms)    src/backward_pass.jl:259-350
    forward_covariance(model,x,u,traj)                               src/forward_pass.jl:37-56
    kl_div_wiki(xnew,xold,Σ_new,traj_new,traj_prev)                  src/klutils.jl:70-103
    calc_η(xnew,xold,sigmanew,ηbracket,traj_new,traj_prev,kl_step)   src/klutils.jl:112-133   (scalar kl_step)
    iLQGkl(problem,x0,traj_prev,model; kl_step, ...)                 src/iLQGkl.jl:25-178,234-252 (single KL constraint)

All array work runs in libddp_amd.so (HIP kernels of csrc/back_pass.hip [GPS variant] and csrc/kl.hip), and so does the
outer loop of iLQGkl with its per-trajectory η bracket (``ddp_ilqgkl_f64``).  A trailing
axis is the batch of independent trajectories.  No CPU fallback.

`model`: the reference calls `df(model,x,u)` and `covariance(model,x,u)` of the un-vendored LinearTimeVaryingModelsBase;
here the model is ``Model(fx[n,n,N(,B)], fu[n,m,N(,B)], R1[n,n])`` — the arrays those calls would return — and
``model_covariance`` is this build's documented choice for `covariance` (empirical covariance of the one-step
prediction residuals, the inline comment at forward_pass.jl:42).  The per-time-step branch (`constrain_per_step`,
iLQGkl.jl:180-232) is not offloaded.
"""
from __future__ import annotations

import ctypes as _C
from dataclasses import dataclass

import numpy as np

from . import _lib
from . import GaussianPolicy, _DevProblem, _lims, default_handle, df, forward_pass

__all__ = ["Model", "grad_kl", "∇kl", "back_pass_gps", "forward_covariance", "kl_div_wiki", "calc_η", "geom", "iLQGkl",
           "model_covariance", "demo_linear_kl"]


@dataclass
class Model:
    fx: np.ndarray          # [n,n,N] or [n,n,N,B]
    fu: np.ndarray          # [n,m,N] or [n,m,N,B]  (only used by model_covariance)
    R1: np.ndarray          # [n,n]


def _b(a, nd):
    """append a unit batch axis to an unbatched array of rank nd"""
    a = _lib.f64(a)
    return a.reshape(a.shape + (1,)) if a.ndim == nd else a


def grad_kl(traj_prev, *, handle=None):
    """``∇kl(traj_prev)`` -> ``(cx,cu,cxx,cxu,cuu)`` with ``cxu`` of shape [m,n,T] like the reference (klutils.jl:20);
    ``(0,0,0,0,0)`` for an empty policy (:9)."""
    if traj_prev is None or traj_prev.isempty():
        return (0, 0, 0, 0, 0)
    h = handle or default_handle()
    batched = np.ndim(traj_prev.K) == 4
    K, k, Si = _b(traj_prev.K, 3), _b(traj_prev.k, 2), _b(traj_prev.Σi, 3)
    m, n, T, B = K.shape
    cx = _lib.result_array((n, T, B)); cu = _lib.result_array((m, T, B)); cxx = _lib.result_array((n, n, T, B))
    cxu = _lib.result_array((m, n, T, B)); cuu = _lib.result_array((m, m, T, B))
    _lib.check(_lib.lib().ddp_kl_terms_f64(h.raw, n, m, T, B, *map(_lib.ptr, (K, k, Si, cx, cu, cxx, cxu, cuu))))
    out = (cx, cu, cxx, cxu, cuu)
    return out if batched else tuple(a[..., 0] for a in out)


globals()["∇kl"] = grad_kl


def back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, kl_cost_terms, *, handle=None):
    """Drop-in for ``back_pass_gps(cx,cu,cxx,cxu,cuu,fx,fu,lims,x,u,kl_cost_terms)`` (backward_pass.jl:259).
    ``kl_cost_terms = ((cxkl,cukl,cxxkl,cxukl,cuukl), ηbracket)``; ``ηbracket`` is a 3-vector, a [3,N] matrix (per-step η),
    or with a batch [3,B] / [3,N,B].  Returns ``(diverge, GaussianPolicy(N,n,m,K,k,Quui,Quu), Vx, Vxx, dV)``."""
    h = handle or default_handle()
    cxa = _lib.f64(cx)
    batched = cxa.ndim == 3
    cx, cu, u = _b(cx, 2), _b(cu, 2), _b(u, 2)
    n, N, B = cx.shape
    m = cu.shape[0]
    fx, fu, cxx, cxu, cuu = map(_lib.f64, (fx, fu, cxx, cxu, cuu))
    assert fx.ndim in (3, 4) and cxx.ndim in (3, 4), "back_pass_gps needs 3-D fx/fu and cxx/cxu/cuu (backward_pass.jl:259)"
    assert cx.shape[:2] == (n, N) and cu.shape[:2] == (m, N) and cxx.shape[:3] == (n, n, N) and cxu.shape[:3] == (n, m, N)
    assert cuu.shape[:3] == (m, m, N)                                                     # the reference's @asserts :266-270
    kl, etab = kl_cost_terms
    kl = [_b(a, nd) for a, nd in zip(kl, (2, 2, 3, 3, 3))]
    etab = np.asarray(etab, dtype=np.float64)
    # ηbracket: [3] one η; [3,N] per-step η of an unbatched call (ηbracket[2,i], :262,293); batch: [3,B] or [3,N,B]
    if etab.ndim == 1:
        eta_tv, eta = False, np.full(B, etab[1])
    elif etab.ndim == 2 and batched:
        eta_tv, eta = False, np.ascontiguousarray(etab[1])
    elif etab.ndim == 2:
        eta_tv, eta = True, np.asfortranarray(etab[1].reshape(N, 1))
    else:
        eta_tv, eta = True, np.asfortranarray(etab[1])
    assert eta.size == (N * B if eta_tv else B), "ηbracket does not match the batch / horizon"
    L = _lims(lims)
    d = _lib.BPDesc(n, m, N, B, 1, int(fx.ndim == 4), 1, int(cxx.ndim == 4), 1, int(L is not None))
    t = _lib.KLCostTerms(*[_lib.ptr(a) for a in kl], _lib.ptr(eta), int(eta_tv))
    K = _lib.result_array((m, n, N, B)); k = _lib.result_array((m, N, B)); Quu = _lib.result_array((m, m, N, B))
    Quui = _lib.result_array((m, m, N, B)); Vx = _lib.result_array((n, N, B)); Vxx = _lib.result_array((n, n, N, B))
    dV = np.zeros((2, B), order="F"); div = np.zeros(B, dtype=np.int32)
    _lib.check(_lib.lib().ddp_back_pass_gps_f64(h.raw, _C.byref(d), *map(_lib.ptr, (cx, cu, cxx, cxu, cuu, fx, fu)), _C.byref(t),
                                                _lib.ptr(L), _lib.ptr(u), *map(_lib.ptr, (K, k, Quu, Quui, Vx, Vxx, dV)),
                                                div.ctypes.data_as(_lib.i32p)))
    if not batched:
        return int(div[0]), GaussianPolicy(N, n, m, K[..., 0], k[..., 0], Quui[..., 0], Quu[..., 0]), Vx[..., 0], Vxx[..., 0], dV[:, 0]
    return div, GaussianPolicy(N, n, m, K, k, Quui, Quu), Vx, Vxx, dV


def forward_covariance(model, x, u, traj, *, handle=None):
    """``forward_covariance(model,x,u,traj)`` -> ``sigmanew[(n+m),(n+m),N(,B)]`` (forward_pass.jl:37-56); ``x``,``u`` are
    only what the reference hands to `df(model,·)`/`covariance(model,·)` — the model here already holds those arrays."""
    h = handle or default_handle()
    batched = np.ndim(traj.K) == 4
    K, Sg = _b(traj.K, 3), _b(traj.Σ, 3)
    m, n, N, B = K.shape
    fx, R1 = _lib.f64(model.fx), _lib.f64(model.R1)
    S = _lib.result_array((n + m, n + m, N, B))
    _lib.check(_lib.lib().ddp_forward_covariance_f64(h.raw, n, m, N, B, _lib.ptr(fx), int(fx.ndim == 4), _lib.ptr(R1), _lib.ptr(K),
                                                     _lib.ptr(Sg), _lib.ptr(S)))
    return S if batched else S[..., 0]


def model_covariance(model, x, u):
    """this build's `covariance(model,x,u)`: Julia `cov` of the residuals x[:,t+1] - fx_t x[:,t] - fu_t u[:,t] (host, n x n)"""
    x, u = np.asarray(x, float), np.asarray(u, float)
    E = np.stack([x[:, t + 1] - model.fx[:, :, t] @ x[:, t] - model.fu[:, :, t] @ u[:, t] for t in range(x.shape[1] - 1)], axis=1)
    return np.atleast_2d(np.cov(E))


def _kl_div(xnew, xold, Σ_new, traj_new, traj_prev, handle=None):
    h = handle or default_handle()
    batched = np.ndim(traj_new.K) == 4
    Kn, kn, Sn = _b(traj_new.K, 3), _b(traj_new.k, 2), _b(traj_new.Σ, 3)
    Kp, kp, Sp, Sip = _b(traj_prev.K, 3), _b(traj_prev.k, 2), _b(traj_prev.Σ, 3), _b(traj_prev.Σi, 3)
    xnew, xold, S = _b(xnew, 2), _b(xold, 2), _b(Σ_new, 3)
    m, n, T, B = Kn.shape
    kld = np.zeros((T, B), order="F"); mean = np.zeros(B)
    _lib.check(_lib.lib().ddp_kl_div_f64(h.raw, n, m, T, B, *map(_lib.ptr, (xnew, xold, S, Kn, kn, Sn, Kp, kp, Sp, Sip, kld, mean))))
    return (kld, mean) if batched else (kld[:, 0], mean[0])


def kl_div_wiki(xnew, xold, Σ_new, traj_new, traj_prev, *, handle=None):
    """``kl_div_wiki`` (klutils.jl:70-103): the clipped per-step divergences; ``Inf`` when a logdet throws (unbatched call)"""
    kld, mean = _kl_div(xnew, xold, Σ_new, traj_new, traj_prev, handle)
    if np.ndim(mean) == 0 and np.isinf(mean) and np.all(np.isfinite(kld)):
        return np.inf
    return kld


def geom(ηbracket):
    ηbracket = np.asarray(ηbracket, float)
    return np.sqrt(ηbracket[0] * ηbracket[2])                                             # klutils.jl:154-155


def calc_η(xnew, xold, sigmanew, ηbracket, traj_new, traj_prev, kl_step, *, handle=None, _mean=None):
    """scalar-``kl_step`` method (klutils.jl:112-133): returns ``(ηbracket, satisfied, divergence)``; mutates ``ηbracket``"""
    if not kl_step > 0:
        return ηbracket, True, 0
    divergence = _kl_div(xnew, xold, sigmanew, traj_new, traj_prev, handle)[1] if _mean is None else _mean
    viol = divergence - kl_step
    satisfied = abs(viol) < 0.1 * kl_step
    if not satisfied:
        if viol < 0:                                                                      # η was too big
            ηbracket[2] = ηbracket[1]
            ηbracket[1] = max(geom(ηbracket), 0.1 * ηbracket[2])
        else:                                                                             # η was too small
            ηbracket[0] = ηbracket[1]
            ηbracket[1] = min(geom(ηbracket), 10.0 * ηbracket[0])
    return ηbracket, bool(satisfied), divergence


def iLQGkl(problem, x0, traj_prev, model, *, kl_step=1.0, lims=None, max_iter=50, cost=None, ηbracket=(1e-8, 1.0, 1e16),
           del0=1e-4, constrain_per_step=False, diff_fun=None, handle=None):
    """``iLQGkl(dynamics,costfun,derivs,x0,traj_prev,model; kl_step, lims, max_iter, cost, ηbracket, del0)`` with a registered
    ``problem`` standing in for the three closures (single KL constraint, iLQGkl.jl:91-178).  ``x0[n,N(,B)]`` is the
    pre-rolled trajectory (the reference errors otherwise, :71-72) and ``cost`` its cost (:69).
    Returns ``(x, u, traj_new, Vx, Vxx, cost, trace)``; ``trace`` is a dict of per-trajectory arrays
    (status 1 SUCCESS :169 / 2 η > ηmax :174 / 3 max_iter :234, iter, η bracket, divergence, n_backpass).
    The loop runs inside ONE library call (``ddp_ilqgkl_f64``); ``DDP_KL_HOSTLOOP=1`` selects the loop on host arrays instead."""
    if constrain_per_step:
        raise NotImplementedError("constrain_per_step (iLQGkl.jl:180-232) is not offloaded")
    if cost is None or np.size(cost) == 0:
        raise ValueError("Initial trajectory supplied, initial cost must also be supplied")                     # :69
    h = handle or default_handle()
    x0 = _lib.f64(x0)
    batched = x0.ndim == 3
    x = _b(x0, 2)
    n, N, B = x.shape
    u = _b(traj_prev.k, 2).copy(order="F")                                                                       # :45
    m = u.shape[0]
    if x.shape[1] != u.shape[1]:
        raise ValueError("pre-rolled initial trajectory must be of correct length (size(x0,2) == N)")            # :72
    prev0 = GaussianPolicy(N, n, m, _b(traj_prev.K, 3), np.zeros_like(u), _b(traj_prev.Σ, 3), _b(traj_prev.Σi, 3))   # k *= 0 (:51)
    etab = np.asarray(ηbracket, dtype=np.float64)
    etab = etab.copy() if etab.shape == (3, B) else np.repeat(etab.reshape(3)[:, None], B, 1)                    # copy (:52); [3,B]: one bracket per trajectory
    del0 = np.full(B, float(del0))
    import os as _os
    if _os.environ.get("DDP_KL_HOSTLOOP") != "1":
        return _ilqgkl_call(h, problem, model, prev0, lims, kl_step, max_iter, x, u, cost, etab, float(del0[0]), batched, diff_fun)
    # ---- DDP_KL_HOSTLOOP=1: the loop of the reference on host arrays, one library call per array operation (cross-check in the tests)
    # STEP 1 (:86): the KL demos hand 3-D arrays to back_pass_gps (demo_linear.jl:91-101)
    fx, fu, _, _, _, cx, cu, cxx, cxu, cuu = df(problem, x, u, handle=h)
    dynb = bool(getattr(problem, "dyn_batched", False))
    fx, fu = _tv(fx, N, dynb), _tv(fu, N, dynb)
    cxx, cxu, cuu = _tv(cxx, N), _tv(cxu, N), _tv(cuu, N)
    kl = grad_kl(prev0, handle=h)                                                                               # :90
    status = np.zeros(B, dtype=int); iters = np.zeros(B, dtype=int); nback = np.zeros(B, dtype=int)
    divergence = np.zeros(B); satisfied = np.zeros(B, dtype=bool)
    live = np.ones(B, dtype=bool)
    out = None
    for it in range(1, max_iter + 1):                                                                           # :91
        idx = np.flatnonzero(live)
        if idx.size == 0:
            break
        iters[idx] = it
        # back_pass until the KL-regularised Quu is positive definite everywhere (:95-122); each trajectory owns its η
        pend = idx.copy()
        guard = 0
        allB = idx.size == B                                   # nothing to gather or scatter while every trajectory is live
        acc = None                                             # results of this iteration, indexed like `idx`
        pos = np.full(B, -1); pos[idx] = np.arange(idx.size)
        while pend.size:
            whole = pend.size == B
            sub = (lambda a: a) if whole else (lambda a: a[..., pend])                                          # noqa: E731
            div, pol, Vx, Vxx, dV = back_pass_gps(sub(cx), sub(cu), cxx if cxx.ndim == 3 else sub(cxx), cxu if cxu.ndim == 3 else sub(cxu),
                                                  cuu if cuu.ndim == 3 else sub(cuu), fx if fx.ndim == 3 else sub(fx),
                                                  fu if fu.ndim == 3 else sub(fu), lims, sub(x), sub(u),
                                                  (tuple(sub(a) for a in kl), etab[:, pend]), handle=h)
            nback[pend] += 1
            good = div == 0
            got = (pol.K, pol.k, pol.Σ, pol.Σi, Vx, Vxx, dV)
            if acc is None and good.all() and pend.size == idx.size:
                acc = list(got)                                # the common case: one back pass, no copies
            else:
                if acc is None:
                    acc = [np.zeros(a_.shape[:-1] + (idx.size,), order="F") for a_ in got]
                tgt = pos[pend[good]]
                for dst, src in zip(acc, got):
                    dst[..., tgt] = src[..., good]
            bad = pend[~good]
            etab[1, bad] += del0[bad]                                                                            # :103-105
            del0[bad] *= 2
            pend = bad
            guard += 1
            if guard > 200:
                raise RuntimeError("back_pass_gps keeps diverging (the reference would loop forever)")
        new = GaussianPolicy(N, n, m, acc[0], acc[1], acc[2], acc[3])
        sel = (lambda a: a) if allB else (lambda a: a[..., idx])                                                # noqa: E731
        pb = problem if allB else _SubProblem(problem, idx, B)
        xs = sel(x)
        xnew, unew, cnew = forward_pass(new, xs[:, 0, :], sel(u), xs, 1.0, pb, lims, diff_fun, handle=h)                  # :132
        mdl = Model(model.fx if (np.ndim(model.fx) == 3 or allB) else model.fx[..., idx], model.fu, model.R1)
        sig = forward_covariance(mdl, xs, sel(u), new, handle=h)                                                # :133
        pv = prev0 if allB else GaussianPolicy(N, n, m, sel(prev0.K), sel(prev0.k), sel(prev0.Σ), sel(prev0.Σi))
        _, mean = _kl_div(xnew, xs, sig, new, pv, h)
        if out is None:
            out = dict(x=np.zeros((n, N, B)), u=np.zeros((m, N, B)), K=np.zeros((m, n, N, B)), S=np.zeros((m, m, N, B)),
                       Si=np.zeros((m, m, N, B)), Vx=np.zeros((n, N, B)), Vxx=np.zeros((n, n, N, B)), cost=np.zeros((cnew.shape[0], B)),
                       dV=np.zeros((2, B)))
        if allB:
            out.update(x=xnew.reshape(n, N, B), u=unew.reshape(m, N, B), cost=cnew.reshape(-1, B), K=new.K, S=new.Σ, Si=new.Σi,
                       Vx=acc[4], Vxx=acc[5], dV=acc[6])
        else:
            out["x"][..., idx], out["u"][..., idx], out["cost"][..., idx] = xnew.reshape(n, N, -1), unew.reshape(m, N, -1), cnew.reshape(-1, idx.size)
            out["K"][..., idx], out["S"][..., idx], out["Si"][..., idx] = new.K, new.Σ, new.Σi
            out["Vx"][..., idx], out["Vxx"][..., idx], out["dV"][..., idx] = acc[4], acc[5], acc[6]
        for j, b in enumerate(idx):                                                                             # :141, :169-177
            eb, sat, dv = calc_η(None, None, None, etab[:, b], None, None, kl_step, _mean=mean[j])
            etab[:, b] = eb
            divergence[b], satisfied[b] = dv, sat
            if sat:
                status[b], live[b] = 1, False
            elif etab[1, b] > 0.999 * etab[2, b]:
                status[b], live[b] = 2, False
    status[live] = 3                                                                                            # :234
    traj_new = GaussianPolicy(N, n, m, out["K"], out["u"].copy(), out["S"], out["Si"])                          # traj_new.k = copy(u) (:239)
    trace = dict(status=status, iter=iters, η=etab, divergence=divergence, satisfied=satisfied, n_backpass=nback, dV=out["dV"])
    if not batched:
        traj_new = GaussianPolicy(N, n, m, out["K"][..., 0], out["u"][..., 0].copy(), out["S"][..., 0], out["Si"][..., 0])
        trace = {k_: (v[..., 0] if isinstance(v, np.ndarray) else v) for k_, v in trace.items()}
        return out["x"][..., 0], out["u"][..., 0], traj_new, out["Vx"][..., 0], out["Vxx"][..., 0], out["cost"][..., 0], trace
    return out["x"], out["u"], traj_new, out["Vx"], out["Vxx"], out["cost"], trace



def _ilqgkl_call(h, problem, model, prev0, lims, kl_step, max_iter, x, u, cost, etab, del0, batched, diff_fun=None):
    """the whole loop of iLQGkl (iLQGkl.jl:91-178) as ONE library call: ``ddp_ilqgkl_f64`` (csrc/kl.hip)"""
    n, N, B = x.shape
    m = u.shape[0]
    dp = _DevProblem(problem, N, B, diff_fun)
    CL = dp.cost_len
    c0 = np.asarray(cost, dtype=np.float64)                                                      # only sum(cost) enters (:74,135)
    c0 = (c0.sum(axis=0) if c0.ndim == 2 else c0.reshape(-1)) if batched else np.array([c0.sum()])   # batch: [CL,B] per-step costs or [B] sums
    c0 = np.ascontiguousarray(np.broadcast_to(c0, (B,)))
    mfx, R1 = _lib.f64(model.fx), _lib.f64(model.R1)
    Kp, Sp, Sip = _lib.f64(prev0.K), _lib.f64(prev0.Σ), _lib.f64(prev0.Σi)
    Lh = _lims(lims)
    o = _lib.ILQGKLOpts()
    _lib.lib().ddp_ilqgkl_default_opts(_C.byref(o))
    o.kl_step, o.max_iter, o.del0 = float(kl_step), int(max_iter), float(del0)
    eb = np.asfortranarray(etab)
    xo = _lib.result_array((n, N, B)); uo = _lib.result_array((m, N, B)); K = _lib.result_array((m, n, N, B))
    S = _lib.result_array((m, m, N, B)); Si = _lib.result_array((m, m, N, B)); Vx = _lib.result_array((n, N, B))
    Vxx = _lib.result_array((n, n, N, B)); co = _lib.result_array((CL, B)); dV = np.zeros((2, B), order="F")
    st = np.zeros((_lib.ILQGKL_NSTATS, B), order="F")
    _lib.check(_lib.lib().ddp_ilqgkl_f64(h.raw, _C.byref(dp.struct), _C.byref(o), _lib.ptr(x), _lib.ptr(c0), _lib.ptr(Kp), _lib.ptr(u),
                                         _lib.ptr(Sp), _lib.ptr(Sip), _lib.ptr(mfx), int(mfx.ndim == 4), _lib.ptr(R1), _lib.ptr(Lh), _lib.ptr(eb),
                                         *map(_lib.ptr, (xo, uo, K, S, Si, Vx, Vxx, co, dV, st)), None))
    trace = dict(status=st[0].astype(int), iter=st[1].astype(int), η=eb, divergence=st[7].copy(), satisfied=st[3] != 0,
                 n_backpass=st[2].astype(int), dV=dV, cost=st[8].copy(), improvement=st[9].copy(), expected_reduction=st[10].copy(),
                 grad_norm=st[11].copy())
    if not batched:
        traj_new = GaussianPolicy(N, n, m, K[..., 0], uo[..., 0].copy(), S[..., 0], Si[..., 0])                  # traj_new.k = copy(u) (:239)
        trace = {k_: (v[..., 0] if isinstance(v, np.ndarray) else v) for k_, v in trace.items()}
        return xo[..., 0], uo[..., 0], traj_new, Vx[..., 0], Vxx[..., 0], co[..., 0], trace
    return xo, uo, GaussianPolicy(N, n, m, K, uo.copy(), S, Si), Vx, Vxx, co, trace


def _tv(a, N, batched=False):
    """give a [r,c] (or, batched, [r,c,B]) array the time axis back_pass_gps wants: [r,c,N] / [r,c,N,B]"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 4 or (a.ndim == 3 and not batched):
        return a
    if batched:
        return np.repeat(a[:, :, None, :], N, 2)
    return np.repeat(a[:, :, None], N, 2)


class _SubProblem:
    """a registered problem restricted to a subset of the batch (per-trajectory dynamics are sliced)"""

    def __new__(cls, problem, idx, B):
        import copy
        p = copy.copy(problem)
        for name in ("A", "B"):
            a = getattr(p, name, None)
            if isinstance(a, np.ndarray) and getattr(p, "dyn_batched", False) and a.shape[-1] == B:
                setattr(p, name, a[..., idx])
        return p


def demo_linear_kl(*, kl_step=1.0, rng=None, T=1000, n=10, m=2, h=0.01, outer=5, R1=None, handle=None, **kwargs):
    """``demo_linear_kl(;kwargs...)`` (src/demo_linear.jl:63-136): the random LTI problem of ``demo_linear``, a rollout of the random
    initial controls, the exact model ``SimpleLTVModel(repeat(A), repeat(B))`` and five outer calls of ``iLQGkl`` starting from the
    identity policy ``GaussianPolicy(Float64,T,n,m)`` (k = 0, quirk Q20).  ``R1`` is what ``covariance(model,x,u)`` of the un-vendored
    LinearTimeVaryingModelsBase would return for that model — unknown here, default 1e-3·I (it only enters Σ of the state)."""
    import scipy.linalg as _sla
    from . import LQProblem, forward_pass
    rng = rng if rng is not None else np.random.default_rng()
    A0 = rng.standard_normal((n, n))
    A = _sla.expm(h * (A0 - A0.T))
    Bm = h * rng.standard_normal((n, m))
    Q, R = h * np.eye(n), 0.1 * h * np.eye(m)
    prob = LQProblem(A, Bm, Q, R)
    u = 0.1 * rng.standard_normal((m, T))
    x, _, _ = forward_pass(None, np.ones(n), u, None, 1.0, prob, None, handle=handle)            # rollout(u) (:107-114)
    x = x.reshape(n, T)
    model = Model(np.repeat(A[:, :, None], T, 2), np.repeat(Bm[:, :, None], T, 2), 1e-3 * np.eye(n) if R1 is None else _lib.f64(R1))
    eye = np.repeat(np.eye(m)[:, :, None], T, 2)
    traj = GaussianPolicy(T, n, m, np.zeros((m, n, T)), np.zeros((m, T)), eye, eye.copy())        # identity ctor: k = 0
    out = None
    outercosts = np.zeros(outer)
    for it in range(outer):
        cost0 = 0.5 * np.sum(x * (Q @ x)) + 0.5 * np.sum(u * (R @ u))                              # :122
        out = iLQGkl(prob, x, traj, model, kl_step=kl_step, cost=cost0, handle=handle, **kwargs)
        x, u, traj = out[0], out[1], out[2]
        outercosts[it] = float(np.sum(out[5]))
    out[6]["outercosts"] = outercosts
    return out
