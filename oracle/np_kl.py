"""Independent array-level NumPy/SciPy restatement of the KL-constrained path (TEST INFRASTRUCTURE ONLY).

Follows, line by line, baggepinnen/DifferentialDynamicProgramming.jl v0.5.0:
  back_pass_gps       src/backward_pass.jl:259-350
  ∇kl, kl_div_wiki, calc_η (scalar kl_step), geom   src/klutils.jl:8-23,70-133,154-155
  forward_covariance  src/forward_pass.jl:37-56
  iLQGkl (single KL constraint)                      src/iLQGkl.jl:25-178,234-252
It exists to cross-check oracle/ddp_oracle_kl.c (LAPACK inv/slogdet/cholesky here, hand-written LU there).
PARITY UNPINNED: no numeric fixture upstream; `df(model,·)`/`covariance(model,·)` come from an un-vendored dependency —
the model here is the triple (fx, fu, R1) and `model_covariance` is this build's documented choice.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

from .np_restatement import PosDef, _chol_upper, boxQP, forward_pass


def grad_kl(Kp, kp, Sip):
    """∇kl(traj_prev) -> cx[n,T], cu[m,T], cxx[n,n,T], cxu[m,n,T] (sic), cuu[m,m,T]   (klutils.jl:8-23)"""
    m, n, T = Kp.shape
    cx, cu, cxx, cuu, cxu = np.zeros((n, T)), np.zeros((m, T)), np.zeros((n, n, T)), np.zeros((m, m, T)), np.zeros((m, n, T))
    for t in range(T):
        K, k, Si = Kp[:, :, t], kp[:, t], Sip[:, :, t]
        cx[:, t] = K.T @ Si @ k
        cu[:, t] = -Si @ k
        cxx[:, :, t] = K.T @ Si @ K
        cuu[:, :, t] = Si
        cxu[:, :, t] = -Si @ K
    return cx, cu, cxx, cxu, cuu


def back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, kl_cost_terms):
    """returns diverge, (K, k, Quui, Quu), Vx, Vxx, dV   (backward_pass.jl:259-350)"""
    (cxkl, cukl, cxxkl, cxukl, cuukl), etab = kl_cost_terms
    etab = np.asarray(etab, dtype=float)
    m = u.shape[0]
    n, _, N = fx.shape
    eta_of = (lambda i: etab[1, i]) if etab.ndim == 2 else (lambda i: etab[1])
    k, K = np.zeros((m, N)), np.zeros((m, n, N))
    Vx, Vxx = np.zeros((n, N)), np.zeros((n, n, N))
    Quu, Quui = np.zeros((m, m, N)), np.zeros((m, m, N))          # `undef` upstream
    dV = np.zeros(2)
    Vx[:, N - 1] = cx[:, N - 1]
    Vxx[:, :, N - 1] = cxx[:, :, N - 1]
    Quu[:, :, N - 1] = cuu[:, :, N - 1] / eta_of(N - 1) + cuukl[:, :, N - 1]
    Quui[:, :, N - 1] = np.linalg.inv(Quu[:, :, N - 1])
    no_lims = lims is None or np.size(lims) == 0 or lims[0, 0] > lims[0, 1]
    for i in range(N - 2, -1, -1):
        F, G, V = fx[:, :, i], fu[:, :, i], Vxx[:, :, i + 1]
        Qu = cu[:, i] + G.T @ Vx[:, i + 1]
        Qx = cx[:, i] + F.T @ Vx[:, i + 1]
        Qux = cxu[:, :, i].T + G.T @ V @ F
        Q = cuu[:, :, i] + G.T @ V @ G
        Qxx = cxx[:, :, i] + F.T @ V @ F
        eta = eta_of(i)
        Qu = Qu / eta + cukl[:, i]
        Qux = Qux / eta + cxukl[:, :, i]
        Q = Q / eta + cuukl[:, :, i]
        Qx = Qx / eta + cxkl[:, i]
        Qxx = Qxx / eta + cxxkl[:, :, i]
        Q = 0.5 * (Q + Q.T)
        Quu[:, :, i] = Q
        if no_lims:
            try:
                R = _chol_upper(Q)
            except PosDef:
                return i + 1, (K, k, Quui, Quu), Vx, Vxx, dV
            k_i = -sla.cho_solve((R, False), Qu)
            K_i = -sla.cho_solve((R, False), Qux)
        else:
            lower, upper = lims[:, 0] - u[:, i], lims[:, 1] - u[:, i]
            try:
                k_i, result, R, free = boxQP(Q, Qu, lower, upper, k[:, min(i + 1, N - 2)].copy())
            except PosDef:
                result = 0
            if result < 1:
                return i + 1, (K, k, Quui, Quu), Vx, Vxx, dV
            K_i = np.zeros((m, n))
            if free.any():
                y = sla.solve_triangular(R, Qux[free, :], trans='T', lower=False)
                K_i[free, :] = -sla.solve_triangular(R, y, lower=False)
        dV = dV + np.array([k_i @ Qu, 0.5 * k_i @ Q @ k_i])
        Vx[:, i] = Qx + K_i.T @ Q @ k_i + K_i.T @ Qu + Qux.T @ k_i
        M = Qxx + K_i.T @ Q @ K_i + K_i.T @ Qux + Qux.T @ K_i
        Vxx[:, :, i] = 0.5 * (M + M.T)
        k[:, i] = k_i
        K[:, :, i] = K_i
        Quui[:, :, i] = np.linalg.inv(Q)
    return 0, (K, k, Quui, Quu), Vx, Vxx, dV


def forward_covariance(model_fx, R1, K, Sigma):
    """sigmanew[(n+m),(n+m),N]   (forward_pass.jl:37-56); `undef` entries are zero"""
    n, _, N = model_fx.shape
    m = K.shape[0]
    S = np.zeros((n + m, n + m, N))
    ix, iu = slice(0, n), slice(n, n + m)
    S[ix, ix, 0] = R1
    for i in range(N - 1):
        Ki, Sg, F = K[:, :, i], Sigma[:, :, i], model_fx[:, :, i]
        S[ix, ix, i + 1] = F @ S[ix, ix, i] @ F.T + R1
        S[iu, ix, i] = Ki @ S[ix, ix, i]
        S[ix, iu, i] = S[ix, ix, i] @ Ki.T
        S[iu, iu, i] = Ki @ S[ix, ix, i] @ Ki.T + Sg
    return S


def model_covariance(fx, fu, x, u):
    """this build's `covariance(model,x,u)`: Julia `cov` of the one-step prediction residuals"""
    N = x.shape[1]
    E = np.stack([x[:, t + 1] - fx[:, :, t] @ x[:, t] - fu[:, :, t] @ u[:, t] for t in range(N - 1)], axis=1)
    return np.atleast_2d(np.cov(E))


def _logdet(A):
    s, ld = np.linalg.slogdet(A)
    if s < 0:
        raise ValueError("DomainError")
    return ld if s > 0 else -np.inf


def kl_div_wiki(xnew, xold, S_new, new, prev):
    """new/prev: dicts with K,k,Σ,Σi   (klutils.jl:70-103).  Returns the clipped vector, or np.inf when a logdet threw."""
    mu_new = xnew - xold
    m, n, T = new["K"].shape
    kl = np.zeros(T)
    for t in range(T):
        mu, St = mu_new[:, t], S_new[:n, :n, t]
        k_diff = prev["k"][:, t] - new["k"][:, t]
        K_diff = prev["K"][:, :, t] - new["K"][:, :, t]
        Sip, Sp, Sn = prev["Si"][:, :, t], prev["S"][:, :, t], new["S"][:, :, t]
        try:
            v = 0.5 * (np.trace(Sip @ Sn) + k_diff @ Sip @ k_diff - m + _logdet(Sp) - _logdet(Sn))
            v += 0.5 * (mu @ K_diff.T @ Sip @ K_diff @ mu + np.trace(K_diff.T @ Sip @ K_diff @ St))
            v += k_diff @ Sip @ K_diff @ mu
        except ValueError:
            return np.inf
        kl[t] = v
    return np.maximum(0, kl)


def calc_eta(etab, divergence_mean, kl_step):
    """scalar kl_step branch (klutils.jl:112-133); mutates and returns etab, satisfied"""
    if not kl_step > 0:
        return etab, True
    viol = divergence_mean - kl_step
    satisfied = abs(viol) < 0.1 * kl_step
    if not satisfied:
        if viol < 0:
            etab[2] = etab[1]
            etab[1] = max(np.sqrt(etab[0] * etab[2]), 0.1 * etab[2])
        else:
            etab[0] = etab[1]
            etab[1] = min(np.sqrt(etab[0] * etab[2]), 10.0 * etab[0])
    return etab, satisfied


def iLQGkl(f, costfun, derivs, x0, prev, model, kl_step=1.0, lims=None, max_iter=50, etab=(1e-8, 1.0, 1e16), del0=1e-4):
    """single-constraint branch of iLQGkl.jl:25-178,234-252.  prev: dict K,k,S,Si (prev["k"] is the control sequence);
    model: dict fx, R1.  derivs(x,u) must return 3-D cost/dynamics arrays: fx,fu,cx,cu,cxx,cxu,cuu."""
    u = prev["k"].copy()
    prev0 = dict(prev, k=np.zeros_like(prev["k"]))
    etab = np.array(etab, dtype=float)
    x = x0
    fx, fu, cx, cu, cxx, cxu, cuu = derivs(x, u)
    kl_terms = grad_kl(prev0["K"], prev0["k"], prev0["Si"])
    info = dict(n_backpass=0)
    satisfied, divergence, status, it = False, 0.0, 3, 0
    for it in range(1, max_iter + 1):
        diverge = 1
        while diverge > 0:
            diverge, (K, k, Quui, Quu), Vx, Vxx, dV = back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, (kl_terms, etab))
            info["n_backpass"] += 1
            if diverge > 0:
                etab[1] += del0
                del0 *= 2
        g_norm = np.mean(np.max(np.abs(k) / (np.abs(u) + 1), axis=0))
        xnew, unew, costnew = forward_pass((K, k), x0[:, 0], u, x, 1.0, f, costfun, lims)
        sig = forward_covariance(model["fx"], model["R1"], K, Quui)
        new = dict(K=K, k=k, S=Quui, Si=Quu)
        kld = kl_div_wiki(xnew, x, sig, new, prev0)
        divergence = np.mean(kld)
        etab, satisfied = calc_eta(etab, divergence, kl_step)
        if satisfied:
            status = 1
            break
        if etab[1] > 0.999 * etab[2]:
            status = 2
            break
    info.update(status=status, iter=it, eta=etab, divergence=divergence, g_norm=g_norm, dV=dV, satisfied=satisfied)
    return xnew, unew, dict(K=K, k=unew.copy(), S=Quui, Si=Quu), Vx, Vxx, costnew, info
