"""ctypes binding of oracle/libddp_oracle.so (CPU restatement, TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class QPOpts(C.Structure):
    _fields_ = [("maxIter", C.c_int), ("minGrad", C.c_double), ("minRelImprove", C.c_double),
                ("stepDec", C.c_double), ("minStep", C.c_double), ("Armijo", C.c_double)]


class Problem(C.Structure):
    _fields_ = [("kind", C.c_int), ("n", C.c_int), ("m", C.c_int), ("N", C.c_int),
                ("A", dp), ("Bm", dp), ("dyn_tv", C.c_int), ("Q", dp), ("R", dp),
                ("g", C.c_double), ("l", C.c_double), ("h", C.c_double), ("d", C.c_double),
                ("goal", C.c_double * 4), ("diff_wrap", C.c_uint)]


class ILQGOpts(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("dlambda", C.c_double), ("lambda_factor", C.c_double),
                ("lambda_max", C.c_double), ("lambda_min", C.c_double), ("tol_fun", C.c_double),
                ("tol_grad", C.c_double), ("max_iter", C.c_int), ("regType", C.c_int),
                ("reduce_ratio_min", C.c_double), ("n_alpha", C.c_int), ("alpha", dp)]


class ILQGResult(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("accepted_iter", C.c_int),
                ("n_backpass", C.c_int), ("n_forward", C.c_int), ("lambda_", C.c_double),
                ("dlambda", C.c_double), ("g_norm", C.c_double), ("dV", C.c_double * 2),
                ("trace_len", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "libddp_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ddp_oracle.c", "ddp_oracle_kl.c", "ddp_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libddp_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ddp_oracle_boxqp.restype = C.c_int
        _LIB.ddp_oracle_back_pass.restype = C.c_int
        _LIB.ddp_oracle_ilqg.restype = C.c_int
        _LIB.ddp_oracle_pass_batch_lq.restype = C.c_int
        _LIB.ddp_oracle_cost_len.restype = C.c_int
    return _LIB


def _f(a):
    """Fortran-ordered contiguous float64 copy (Julia memory layout)."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return None if a is None else a.ctypes.data_as(dp)


def boxqp(H, g, lower, upper, x0, opts=None):
    m = len(g)
    H, g, lower, upper, x0 = _f(H), _f(g), _f(lower), _f(upper), _f(x0)
    x = np.zeros(m); Hfree = np.zeros((m, m), order="F"); free = np.zeros(m, dtype=np.int32)
    nfree = C.c_int(0); iters = C.c_int(0)
    o = None
    if opts is not None:
        o = QPOpts(**opts)
    res = lib().ddp_oracle_boxqp(m, _p(H), _p(g), _p(lower), _p(upper), _p(x0),
                                 C.byref(o) if o is not None else None, _p(x), _p(Hfree),
                                 free.ctypes.data_as(ip), C.byref(nfree), C.byref(iters))
    nf = nfree.value
    return x, res, Hfree[:nf, :nf].copy(), free.astype(bool), iters.value


def back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, lims, x, u):
    """Same argument order as the reference's back_pass (backward_pass.jl:217)."""
    cx, cu, cxx, cxu, cuu, fx, fu, u = map(_f, (cx, cu, cxx, cxu, cuu, fx, fu, u))
    m, N = u.shape
    n = fx.shape[0]
    fx_tv = int(fx.ndim == 3)
    c_tv = int(cxx.ndim == 3)
    K = np.zeros((m, n, N), order="F"); k = np.zeros((m, N), order="F"); Quu = np.zeros((m, m, N), order="F")
    Vx = np.zeros((n, N), order="F"); Vxx = np.zeros((n, n, N), order="F"); dV = np.zeros(2)
    L = None if lims is None or np.size(lims) == 0 else _f(lims)
    d = lib().ddp_oracle_back_pass(n, m, N, _p(cx), _p(cu), _p(cxx), _p(cxu), _p(cuu), _p(fx), _p(fu),
                                   fx_tv, c_tv, C.c_double(lam), int(regType), _p(L), _p(u),
                                   _p(K), _p(k), _p(Quu), _p(Vx), _p(Vxx), _p(dV))
    return d, (K, k, Quu), Vx, Vxx, dV


class _Keep:
    """keeps numpy buffers alive next to the ctypes struct that points at them"""


def make_problem(kind, n, m, N, A=None, B=None, Q=None, R=None, pend=None, diff_wrap=0):
    p = Problem()
    p.diff_wrap = int(diff_wrap)
    keep = _Keep()
    p.kind = 0 if kind == "lq" else 1
    p.n, p.m, p.N = n, m, N
    keep.Q, keep.R = _f(Q), _f(R)
    p.Q, p.R = _p(keep.Q), _p(keep.R)
    if kind == "lq":
        keep.A, keep.B = _f(A), _f(B)
        p.A, p.Bm = _p(keep.A), _p(keep.B)
        p.dyn_tv = int(keep.A.ndim == 3)
    else:
        p.g, p.l, p.h, p.d = pend["g"], pend["l"], pend["h"], pend["d"]
        for i in range(4):
            p.goal[i] = float(pend["goal"][i])
    p._keep = keep
    return p


def forward_pass(p, policy, x0, u, x, alpha, lims):
    n, m, N = p.n, p.m, p.N
    CL = lib().ddp_oracle_cost_len(C.byref(p))
    x0, u = _f(x0), _f(u)
    K = k = xx = None
    if policy is not None:
        K, k = _f(policy[0]), _f(policy[1])
        xx = _f(x)
    L = None if lims is None or np.size(lims) == 0 else _f(lims)
    xnew = np.zeros((n, N), order="F"); unew = np.zeros((m, N), order="F"); cnew = np.zeros(CL)
    lib().ddp_oracle_forward_pass(C.byref(p), _p(K), _p(k), _p(x0), _p(u), _p(xx), C.c_double(alpha),
                                  _p(L), _p(xnew), _p(unew), _p(cnew))
    return xnew, unew, cnew


def df(p, x, u):
    n, m, N = p.n, p.m, p.N
    x, u = _f(x), _f(u).copy(order="F")
    cx = np.zeros((n, N), order="F"); cu = np.zeros((m, N), order="F")
    fx = np.zeros((n, n, N), order="F"); fu = np.zeros((n, m, N), order="F")
    lib().ddp_oracle_df(C.byref(p), _p(x), _p(u), _p(cx), _p(cu), _p(fx), _p(fu))
    return fx, fu, cx, cu


def expm(A):
    A = _f(A)
    E = np.zeros_like(A, order="F")
    lib().ddp_oracle_expm(A.shape[0], _p(A), _p(E))
    return E


def ilqg(p, x0, u0, lims=None, trace_cap=2048, **kw):
    """kw: alpha, tol_fun, tol_grad, max_iter, lam, dlam, lam_factor, lam_max, lam_min, regType,
    reduce_ratio_min (reference defaults, iLQG.jl:143-163)."""
    n, m, N = p.n, p.m, p.N
    CL = lib().ddp_oracle_cost_len(C.byref(p))
    o = ILQGOpts()
    lib().ddp_oracle_ilqg_default_opts(C.byref(o))
    alpha = _f(kw.pop("alpha", 10.0 ** np.linspace(0, -3, 11)))
    o.n_alpha, o.alpha = len(alpha), _p(alpha)
    names = dict(lam="lambda_", dlam="dlambda", lam_factor="lambda_factor", lam_max="lambda_max",
                 lam_min="lambda_min")
    for key, val in kw.items():
        setattr(o, names.get(key, key), val)
    x0, u0 = _f(x0), _f(u0)
    L = None if lims is None or np.size(lims) == 0 else _f(lims)
    x = np.zeros((n, N), order="F"); u = np.zeros((m, N), order="F")
    K = np.zeros((m, n, N), order="F"); k = np.zeros((m, N), order="F"); Quu = np.zeros((m, m, N), order="F")
    Vx = np.zeros((n, N), order="F"); Vxx = np.zeros((n, n, N), order="F"); cost = np.zeros(CL)
    res = ILQGResult()
    trc, trl, tra, trg = (np.zeros(trace_cap) for _ in range(4))
    lib().ddp_oracle_ilqg(C.byref(p), C.byref(o), _p(x0), _p(u0), _p(L), _p(x), _p(u), _p(K), _p(k), _p(Quu),
                          _p(Vx), _p(Vxx), _p(cost), C.byref(res), trace_cap, _p(trc), _p(trl), _p(tra), _p(trg))
    tl = res.trace_len
    info = dict(status=res.status, iter=res.iter, accepted_iter=res.accepted_iter, n_backpass=res.n_backpass,
                n_forward=res.n_forward, lam=res.lambda_, dlam=res.dlambda, g_norm=res.g_norm,
                dV=np.array(res.dV[:]), trace=dict(cost=trc[:tl], lam=trl[:tl], alpha=tra[:tl]))
    return x, u, (K, k, Quu), Vx, Vxx, cost, info


# ---------------------------------------------------------------- KL-constrained path (ddp_oracle_kl.c)
class ILQGKLResult(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("n_backpass", C.c_int), ("satisfied", C.c_int),
                ("eta", C.c_double * 3), ("divergence", C.c_double), ("g_norm", C.c_double), ("dV", C.c_double * 2),
                ("cost0", C.c_double)]


def kl_terms(K, k, Sigmai):
    K, k, Sigmai = _f(K), _f(k), _f(Sigmai)
    m, n, T = K.shape
    cx = np.zeros((n, T), order="F"); cu = np.zeros((m, T), order="F"); cxx = np.zeros((n, n, T), order="F")
    cxu = np.zeros((m, n, T), order="F"); cuu = np.zeros((m, m, T), order="F")
    lib().ddp_oracle_kl_terms(n, m, T, _p(K), _p(k), _p(Sigmai), _p(cx), _p(cu), _p(cxx), _p(cxu), _p(cuu))
    return cx, cu, cxx, cxu, cuu


def back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, kl_cost_terms):
    """Same argument order as the reference (backward_pass.jl:259).  kl_cost_terms = ((cxkl,cukl,cxxkl,cxukl,cuukl), ηbracket)
    with ηbracket a 3-vector or a [3,N] matrix.  Returns diverge, (K, k, Quui, Quu), Vx, Vxx, dV."""
    cx, cu, cxx, cxu, cuu, fx, fu, u = map(_f, (cx, cu, cxx, cxu, cuu, fx, fu, u))
    kl, etab = kl_cost_terms
    kl = [_f(a) for a in kl]
    etab = np.asarray(etab, dtype=np.float64)
    eta = _f(etab[1] if etab.ndim == 2 else [etab[1]])
    m, N = u.shape
    n = fx.shape[0]
    K = np.zeros((m, n, N), order="F"); k = np.zeros((m, N), order="F"); Quu = np.zeros((m, m, N), order="F")
    Quui = np.zeros((m, m, N), order="F"); Vx = np.zeros((n, N), order="F"); Vxx = np.zeros((n, n, N), order="F"); dV = np.zeros(2)
    L = None if lims is None or np.size(lims) == 0 else _f(lims)
    lib().ddp_oracle_back_pass_gps.restype = C.c_int
    d = lib().ddp_oracle_back_pass_gps(n, m, N, _p(cx), _p(cu), _p(cxx), _p(cxu), _p(cuu), _p(fx), _p(fu), _p(L), _p(u),
                                       _p(kl[0]), _p(kl[1]), _p(kl[2]), _p(kl[3]), _p(kl[4]), _p(eta), int(etab.ndim == 2),
                                       _p(K), _p(k), _p(Quu), _p(Quui), _p(Vx), _p(Vxx), _p(dV))
    return d, (K, k, Quui, Quu), Vx, Vxx, dV


def forward_covariance(model_fx, R1, K, Sigma):
    model_fx, R1, K, Sigma = _f(model_fx), _f(R1), _f(K), _f(Sigma)
    n, _, N = model_fx.shape
    m = K.shape[0]
    S = np.zeros((n + m, n + m, N), order="F")
    lib().ddp_oracle_forward_covariance(n, m, N, _p(model_fx), _p(R1), _p(K), _p(Sigma), _p(S))
    return S


def model_covariance(fx, fu, x, u):
    fx, fu, x, u = _f(fx), _f(fu), _f(x), _f(u)
    n, N = x.shape
    R1 = np.zeros((n, n), order="F")
    lib().ddp_oracle_model_covariance(n, u.shape[0], N, _p(fx), _p(fu), _p(x), _p(u), _p(R1))
    return R1


def kl_div_wiki(xnew, xold, S_new, new, prev):
    """new/prev: dicts K,k,S,Si.  Returns the clipped vector, or inf when a logdet threw."""
    xnew, xold, S_new = _f(xnew), _f(xold), _f(S_new)
    a = {key: _f(new[key]) for key in ("K", "k", "S")}
    b = {key: _f(prev[key]) for key in ("K", "k", "S", "Si")}
    m, n, T = a["K"].shape
    out = np.zeros(T)
    lib().ddp_oracle_kl_div_wiki.restype = C.c_int
    threw = lib().ddp_oracle_kl_div_wiki(n, m, T, _p(xnew), _p(xold), _p(S_new), _p(a["K"]), _p(a["k"]), _p(a["S"]),
                                         _p(b["K"]), _p(b["k"]), _p(b["S"]), _p(b["Si"]), _p(out))
    return np.inf if threw else out


def calc_eta(etab, divergence_mean, kl_step):
    e = _f(etab).copy()
    lib().ddp_oracle_calc_eta.restype = C.c_int
    s = lib().ddp_oracle_calc_eta(_p(e), C.c_double(divergence_mean), C.c_double(kl_step))
    return e, bool(s)


def ilqgkl(p, x0, cost0, prev, model, kl_step=1.0, lims=None, max_iter=50, etab=(1e-8, 1.0, 1e16), del0=1e-4):
    n, m, N = p.n, p.m, p.N
    CL = lib().ddp_oracle_cost_len(C.byref(p))
    x0 = _f(x0)
    b = {key: _f(prev[key]) for key in ("K", "k", "S", "Si")}
    mfx, R1, e = _f(model["fx"]), _f(model["R1"]), _f(etab)
    L = None if lims is None or np.size(lims) == 0 else _f(lims)
    x = np.zeros((n, N), order="F"); u = np.zeros((m, N), order="F"); K = np.zeros((m, n, N), order="F"); k = np.zeros((m, N), order="F")
    Quu = np.zeros((m, m, N), order="F"); Quui = np.zeros((m, m, N), order="F"); Vx = np.zeros((n, N), order="F")
    Vxx = np.zeros((n, n, N), order="F"); cost = np.zeros(CL)
    res = ILQGKLResult()
    lib().ddp_oracle_ilqgkl.restype = C.c_int
    lib().ddp_oracle_ilqgkl(C.byref(p), _p(x0), C.c_double(cost0), _p(b["K"]), _p(b["k"]), _p(b["S"]), _p(b["Si"]), _p(mfx), _p(R1),
                            _p(L), C.c_double(kl_step), int(max_iter), _p(e), C.c_double(del0), _p(x), _p(u), _p(K), _p(k),
                            _p(Quu), _p(Quui), _p(Vx), _p(Vxx), _p(cost), C.byref(res))
    info = dict(status=res.status, iter=res.iter, n_backpass=res.n_backpass, satisfied=bool(res.satisfied), eta=np.array(res.eta[:]),
                divergence=res.divergence, g_norm=res.g_norm, dV=np.array(res.dV[:]))
    return x, u, dict(K=K, k=k, S=Quui, Si=Quu), Vx, Vxx, cost, info


def ilqg_prerolled(p, x0, u0, cost0=None, lims=None, **kw):
    """pre-rolled initial trajectory x0[n,N] (iLQG.jl:193-197); kw as in ilqg()"""
    n, m, N = p.n, p.m, p.N
    CL = lib().ddp_oracle_cost_len(C.byref(p))
    o = ILQGOpts()
    lib().ddp_oracle_ilqg_default_opts(C.byref(o))
    alpha = _f(kw.pop("alpha", 10.0 ** np.linspace(0, -3, 11)))
    o.n_alpha, o.alpha = len(alpha), _p(alpha)
    names = dict(lam="lambda_", dlam="dlambda", lam_factor="lambda_factor", lam_max="lambda_max", lam_min="lambda_min")
    for key, val in kw.items():
        setattr(o, names.get(key, key), val)
    x0, u0 = _f(x0), _f(u0)
    c0 = None if cost0 is None else _f(cost0)
    L = None if lims is None or np.size(lims) == 0 else _f(lims)
    x = np.zeros((n, N), order="F"); u = np.zeros((m, N), order="F")
    K = np.zeros((m, n, N), order="F"); k = np.zeros((m, N), order="F"); Quu = np.zeros((m, m, N), order="F")
    Vx = np.zeros((n, N), order="F"); Vxx = np.zeros((n, n, N), order="F"); cost = np.zeros(CL)
    res = ILQGResult()
    lib().ddp_oracle_ilqg_prerolled.restype = C.c_int
    lib().ddp_oracle_ilqg_prerolled(C.byref(p), C.byref(o), _p(x0), _p(u0), _p(c0), _p(L), _p(x), _p(u), _p(K), _p(k), _p(Quu),
                                    _p(Vx), _p(Vxx), _p(cost), C.byref(res))
    info = dict(status=res.status, iter=res.iter, accepted_iter=res.accepted_iter, n_backpass=res.n_backpass,
                n_forward=res.n_forward, lam=res.lambda_, dlam=res.dlambda, g_norm=res.g_norm, dV=np.array(res.dV[:]))
    return x, u, (K, k, Quu), Vx, Vxx, cost, info


def ilqg_trace7(p, x0, u0, lims=None, trace_cap=2048, **kw):
    """ilqg() returning all seven per-iteration trace keys (iLQG.jl:257,325-330) as info["history"]"""
    n, m, N = p.n, p.m, p.N
    CL = lib().ddp_oracle_cost_len(C.byref(p))
    o = ILQGOpts()
    lib().ddp_oracle_ilqg_default_opts(C.byref(o))
    alpha = _f(kw.pop("alpha", 10.0 ** np.linspace(0, -3, 11)))
    o.n_alpha, o.alpha = len(alpha), _p(alpha)
    names = dict(lam="lambda_", dlam="dlambda", lam_factor="lambda_factor", lam_max="lambda_max", lam_min="lambda_min")
    for key, val in kw.items():
        setattr(o, names.get(key, key), val)
    x0, u0 = _f(x0), _f(u0)
    L = None if lims is None or np.size(lims) == 0 else _f(lims)
    x = np.zeros((n, N), order="F"); u = np.zeros((m, N), order="F")
    K = np.zeros((m, n, N), order="F"); k = np.zeros((m, N), order="F"); Quu = np.zeros((m, m, N), order="F")
    Vx = np.zeros((n, N), order="F"); Vxx = np.zeros((n, n, N), order="F"); cost = np.zeros(CL)
    res = ILQGResult()
    tr7 = np.zeros((7, trace_cap), order="F")
    lib().ddp_oracle_ilqg_trace7.restype = C.c_int
    lib().ddp_oracle_ilqg_trace7(C.byref(p), C.byref(o), _p(x0), _p(u0), _p(L), _p(x), _p(u), _p(K), _p(k), _p(Quu), _p(Vx), _p(Vxx),
                                 _p(cost), C.byref(res), trace_cap, _p(tr7))
    tl = res.trace_len
    hist = {key: tr7[c, :tl] for c, key in enumerate(("λ", "dλ", "α", "improvement", "cost", "reduce_ratio", "grad_norm"))}
    info = dict(status=res.status, iter=res.iter, history=hist, trace_len=tl)
    return x, u, (K, k, Quu), Vx, Vxx, cost, info
