"""NumPy/SciPy restatement of the iLQG hot path of DifferentialDynamicProgramming.jl v0.5.0.

TEST INFRASTRUCTURE ONLY (see oracle/ddp_oracle.h).  This is the *second*, independent
restatement: it is written at array level with numpy/scipy linear algebra (LAPACK potrf /
trtrs / expm from SciPy) whereas oracle/ddp_oracle.c spells every loop out.  The two are
checked against each other in tests/test_oracle.py and this one generates the committed
fixtures under tests/golden/ (tests/golden/make_golden.py).

PARITY PINNING: the reference has no numeric golden vectors and Julia is not installed in
this image, so neither restatement has been compared with outputs of the Julia code:
"parity unpinned" at bit level; pinned by mutual agreement + analytic KATs + the
reference's statistical thresholds (test/test_readme.jl:68-70).

All time indices are 0-based here; `diverge` is returned 1-based like the reference.
Arrays use the reference's shapes (x[n,N], K[m,n,N], ...) as ordinary numpy arrays.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


class PosDef(Exception):
    pass


def _chol_upper(A):
    """cholesky(Hermitian(A)).U — upper triangle is read (backward_pass.jl:35, boxQP.jl:111)."""
    Au = np.triu(A)
    As = Au + np.triu(A, 1).T
    try:
        return np.linalg.cholesky(As).T
    except np.linalg.LinAlgError as e:  # PosDefException
        raise PosDef() from e


def jl_clamp(x, lo, hi):
    x = np.asarray(x, dtype=float)
    return np.where(x > hi, hi, np.where(x < lo, lo, x))


# --------------------------------------------------------------------------- boxQP.jl:29-188
def boxQP(H, g, lower, upper, x0, maxIter=100, minGrad=1e-8, minRelImprove=1e-8,
          stepDec=0.6, minStep=1e-22, Armijo=0.1):
    n = H.shape[0]
    clamped = np.zeros(n, bool)
    free = np.ones(n, bool)
    oldvalue = 0.0
    result = 0
    Hfree = np.zeros((n, n))
    x = jl_clamp(x0, lower, upper)                                   # :58
    value = x @ g + (0.5 * x) @ H @ x                                # :63
    it = 1
    while it <= maxIter:                                             # :71
        if result != 0:
            break
        if it > 1 and (oldvalue - value) < minRelImprove * abs(oldvalue):   # :78
            result = 4
            break
        oldvalue = value
        grad = g + H @ x                                             # :85
        old_clamped = clamped
        clamped = ((x == lower) & (grad > 0)) | ((x == upper) & (grad < 0))  # :93
        free = ~clamped
        if clamped.all():                                            # :98
            result = 6
            break
        factorize = True if it == 1 else bool(np.any(old_clamped != clamped))
        if factorize:
            Hfree = _chol_upper(H[np.ix_(free, free)])               # :111 (may raise -> caller)
        gnorm = np.linalg.norm(grad[free])                           # :120
        if gnorm < minGrad:
            result = 5
            break
        grad_clamped = g + H @ (x * clamped)                         # :127
        search = np.zeros(n)
        rhs = grad_clamped[free]
        y = sla.solve_triangular(Hfree, rhs, trans='T', lower=False)
        search[free] = -sla.solve_triangular(Hfree, y, lower=False) - x[free]   # :129
        sdotg = float(np.sum(search * grad))                         # :132
        if sdotg >= 0:
            break
        step = 1.0
        xc = jl_clamp(x + step * search, lower, upper)
        vc = xc @ g + (0.5 * xc) @ H @ xc
        while (vc - oldvalue) / (step * sdotg) < Armijo:             # :142
            step = step * stepDec
            xc = jl_clamp(x + step * search, lower, upper)
            vc = xc @ g + (0.5 * xc) @ H @ xc
            if step < minStep:
                result = 2
                break
        x = xc
        value = vc
        it += 1
    if it == maxIter:                                                # :167
        result = 1
    return x, result, Hfree, free


# ------------------------------------------------------------- backward_pass.jl:162-252,28-79
def back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, lims, x, u):
    """Dispatch on array rank like the reference: fx 2-D -> LTI (:217), fx 3-D & cxx 2-D ->
    :162, cxx 3-D -> :179.  Returns diverge, (K, k, Quu), Vx, Vxx, dV."""
    m, N = u.shape
    n = fx.shape[0]
    fx_tv = fx.ndim == 3
    c_tv = cxx.ndim == 3
    k = np.zeros((m, N))
    K = np.zeros((m, n, N))
    Vx = np.zeros((n, N))
    Vxx = np.zeros((n, n, N))
    Quu = np.zeros((m, m, N))            # `undef` in the reference
    dV = np.zeros(2)
    Vx[:, N - 1] = cx[:, N - 1]
    Vxx[:, :, N - 1] = cxx[:, :, N - 1] if c_tv else cxx
    Quu[:, :, N - 1] = cuu[:, :, N - 1] if c_tv else cuu
    no_lims = (lims is None) or (np.size(lims) == 0) or (lims[0, 0] > lims[0, 1])   # :31
    diverge = 0
    In, Im = np.eye(n), np.eye(m)
    for i in range(N - 2, -1, -1):
        fxi = fx[:, :, i] if fx_tv else fx
        fui = fu[:, :, i] if fx_tv else fu
        cxxi = cxx[:, :, i] if c_tv else cxx
        cxui = cxu[:, :, i] if c_tv else cxu
        cuui = cuu[:, :, i] if c_tv else cuu
        V = Vxx[:, :, i + 1]
        Qu = cu[:, i] + fui.T @ Vx[:, i + 1]
        Qx = cx[:, i] + fxi.T @ Vx[:, i + 1]
        Qux = cxui.T + (fui.T @ V) @ fxi
        Quu[:, :, i] = cuui + (fui.T @ V) @ fui
        Qxx = cxxi + (fxi.T @ V) @ fxi
        Vreg = V + (lam * In if regType == 2 else 0)
        Qux_reg = cxui.T + (fui.T @ Vreg) @ fxi
        QuuF = cuui + (fui.T @ Vreg) @ fui + (lam * Im if regType == 1 else 0)
        # ---- @end_backward_pass
        if no_lims:
            try:
                R = _chol_upper(QuuF)
            except PosDef:
                diverge = i + 1
                return diverge, (K, k, Quu), Vx, Vxx, dV
            k_i = -sla.cho_solve((R, False), Qu)
            K_i = -sla.cho_solve((R, False), Qux_reg)
        else:
            lower = lims[:, 0] - u[:, i]
            upper = lims[:, 1] - u[:, i]
            ws = min(i + 1, N - 2)                                   # k[:,min(i+1,N-1)] 1-based
            try:
                k_i, result, R, free = boxQP(QuuF, Qu, lower, upper, k[:, ws].copy())
            except PosDef:
                result = 0
            if result < 1:
                diverge = i + 1
                return diverge, (K, k, Quu), Vx, Vxx, dV
            K_i = np.zeros((m, n))
            if free.any():
                y = sla.solve_triangular(R, Qux_reg[free, :], trans='T', lower=False)
                K_i[free, :] = -sla.solve_triangular(R, y, lower=False)
        Quuk = Quu[:, :, i] @ k_i
        kQuuk = k_i @ Quuk
        KQuuk = K_i.T @ Quuk
        KQuuK = (K_i.T @ Quu[:, :, i]) @ K_i
        dV = dV + np.array([k_i @ Qu, 0.5 * kQuuk])
        Vx[:, i] = Qx + KQuuk + K_i.T @ Qu + Qux.T @ k_i
        M = Qxx + KQuuK + K_i.T @ Qux + Qux.T @ K_i
        Vxx[:, :, i] = (M + M.T) / 2
        k[:, i] = k_i
        K[:, :, i] = K_i
    return diverge, (K, k, Quu), Vx, Vxx, dV


# ----------------------------------------------------------------- forward_pass.jl:9-33
def wrapped_diff(mask):
    """the `diff` a caller with angle states passes: subtraction, coordinates in the bit mask wrapped to [-pi, pi]"""
    sel = np.array([(mask >> j) & 1 for j in range(32)], bool)

    def diff(a, b):
        d = a - b
        w = sel[: d.shape[0]]
        d[w] = np.remainder(d[w] + np.pi, 2 * np.pi) - np.pi
        return d
    return diff


def forward_pass(policy, x0, u, x, alpha, f, costfun, lims, diff=np.subtract):
    """policy: None (empty GaussianPolicy) or (K, k); diff: forward_pass.jl:9 (iLQG.jl:160 passes `-`)."""
    n = x0.shape[0]
    m, N = u.shape
    xnew = np.empty((n, N))
    xnew[:, 0] = x0
    unew = u.copy()
    for i in range(N):
        if policy is not None:
            K, k = policy
            unew[:, i] += k[:, i] * alpha
            dx = diff(xnew[:, i], x[:, i])
            unew[:, i] += K[:, :, i] @ dx
        if lims is not None and np.size(lims) > 0:
            unew[:, i] = jl_clamp(unew[:, i], lims[:, 0], lims[:, 1])
        xn = f(xnew[:, i], unew[:, i], i)
        if i < N - 1:
            xnew[:, i + 1] = xn
    cnew = costfun(xnew, unew)
    return xnew, unew, cnew


# --------------------------------------------------------------------- iLQG.jl:143-341
DEFAULT_ALPHA = 10.0 ** np.linspace(0, -3, 11)


def iLQG(f, costfun, df, x0, u0, lims=None, alpha=DEFAULT_ALPHA, tol_fun=1e-7, tol_grad=1e-4,
         max_iter=500, lam=1.0, dlam=1.0, lam_factor=1.6, lam_max=1e10, lam_min=1e-6, regType=1,
         reduce_ratio_min=0.0, diff_fun=np.subtract):
    n = x0.shape[0]
    m, N = u0.shape
    u = u0
    trace = dict(cost=[], lam=[], alpha=[], g_norm=[])
    diverge = True
    for ai in alpha:                                                 # :181-192
        x, un, cost = forward_pass(None, x0, ai * u, None, 1, f, costfun, lims)
        if np.all(np.abs(x) < 1e8):
            u = un
            diverge = False
            break
    if diverge:
        return None
    flg_change = True
    status = 0
    it = acc = 1
    n_bp = n_fp = 0
    Vx = Vxx = dV = None
    K = np.zeros((m, n, N)); k = np.zeros((m, N)); Quu = np.zeros((m, m, N))
    g_norm = 0.0
    while acc <= max_iter:
        if flg_change:
            fx, fu, cx, cu, cxx, cxu, cuu = df(x, u)
            flg_change = False
        back_pass_done = False
        while not back_pass_done:
            dvg, (K, k, Quu), Vx, Vxx, dV = back_pass(cx, cu, cxx, cxu, cuu, fx, fu, lam, regType, lims, x, u)
            n_bp += 1
            if dvg > 0:
                dlam, lam = max(dlam * lam_factor, lam_factor), max(lam * dlam, lam_min)   # Q1
                if lam > lam_max:
                    break
                continue
            back_pass_done = True
        g_norm = float(np.mean(np.max(np.abs(k) / (np.abs(u) + 1), axis=0)))
        trace['g_norm'].append(g_norm)
        if g_norm < tol_grad and lam < 1e-5:
            status = 1
            break
        fwd_pass_done = False
        dcost = 0.0
        a_used = np.nan
        if back_pass_done:
            for ai in alpha:
                xnew, unew, costnew = forward_pass((K, k), x0, u, x, ai, f, costfun, lims, diff_fun)
                n_fp += 1
                a_used = ai
                dcost = np.sum(cost) - np.sum(costnew)
                expected = -ai * (dV[0] + ai * dV[1])
                z = dcost / expected if expected > 0 else np.sign(dcost)
                if z > reduce_ratio_min:
                    fwd_pass_done = True
                    break
        if fwd_pass_done:
            dlam = min(dlam / lam_factor, 1 / lam_factor)
            lam = max(lam * dlam, lam_min)
            x, u, cost = xnew.copy(), unew.copy(), np.copy(costnew)
            k = u.copy()                                              # Q3
            flg_change = True
            if dcost < tol_fun:
                status = 2
                break
            acc += 1
        else:
            a_used = np.nan
            dlam, lam = max(dlam * lam_factor, lam_factor), max(lam * dlam, lam_min)
            if lam > lam_max:
                status = 3
                break
        trace['cost'].append(float(np.sum(cost))); trace['lam'].append(lam); trace['alpha'].append(a_used)
        it += 1
    if status == 0:
        status = 4
    info = dict(status=status, iter=it, accepted_iter=acc, n_backpass=n_bp, n_forward=n_fp, lam=lam,
                dlam=dlam, g_norm=g_norm, dV=dV, trace=trace)
    return x, u, (K, k, Quu), Vx, Vxx, cost, info


# ------------------------------------------------------------- problem families (the demos)
def make_lq_problem(rng, n=10, m=2, T=1000, h=0.01):
    """demo_linear.jl:8-26 / test_readme.jl:6-23 with a NumPy generator."""
    A0 = rng.standard_normal((n, n))
    A = sla.expm(h * (A0 - A0.T))
    B = h * rng.standard_normal((n, m))
    Q = h * np.eye(n)
    R = 0.1 * h * np.eye(m)
    x0 = np.ones(n)
    u0 = 0.1 * rng.standard_normal((m, T))
    return dict(A=A, B=B, Q=Q, R=R, x0=x0, u0=u0, n=n, m=m, N=T)


def lq_closures(A, B, Q, R):
    def f(x, u, i):
        u[np.isnan(u)] = 0
        Ai = A[:, :, i] if A.ndim == 3 else A
        Bi = B[:, :, i] if B.ndim == 3 else B
        return Ai @ x + Bi @ u

    def costfun(x, u):   # per-step split of demo_linear.jl:49 (sum is the reference's scalar)
        return 0.5 * np.sum(x * (Q @ x), axis=0) + 0.5 * np.sum(u * (R @ u), axis=0)

    def df(x, u):
        u[np.isnan(u)] = 0
        return A, B, Q @ x, R @ u, Q, np.zeros((A.shape[0], B.shape[1])), R

    return f, costfun, df


PENDCART = dict(g=9.82, l=0.35, h=0.01, d=0.99, Q=np.diag([10.0, 1, 2, 1]), R=np.array([[1.0]]),
                goal=np.array([np.pi, 0, 0, 0]), x0=np.array([np.pi - 0.6, 0, 0, 0]),
                lims=5.0 * np.array([[-1.0, 1.0]]), T=600)


def pendcart_closures(P=PENDCART):
    g, l, h, d, Q, R, goal = P['g'], P['l'], P['h'], P['d'], P['Q'], P['R'], P['goal']

    def f(x, u, i):                                                  # system_pendcart.jl:83-89
        u[np.isnan(u)] = 0
        return np.array([x[0] + h * x[1],
                         x[1] + h * (-g / l * np.sin(x[0]) + u[0] / l * np.cos(x[0]) - d * x[1]),
                         x[2] + h * x[3],
                         x[3] + h * u[0]])

    def costfun(x, u):                                               # :97-106
        dx = x - goal[:, None]
        T = u.shape[1]
        c = np.empty(T + 1)
        c[:T] = 0.5 * (np.sum(dx * (Q @ dx), axis=0) + R[0, 0] * u[0] ** 2)
        c[T] = 0.5 * (dx[:, -1] @ Q @ dx[:, -1])
        return c

    def df(x, u):                                                    # :137-154
        u[np.isnan(u)] = 0
        D, I = x.shape[0], u.shape[1]
        cx = Q @ (x - goal[:, None])
        cu = R * u
        fxd = np.empty((D, D, I)); fud = np.empty((D, 1, I))
        for ii in range(I):
            fxc = np.array([[0, 1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 0]], float)
            fuc = np.array([0, 0, 0, 1.0])
            fxc[1, 0] = -g / l * np.cos(x[0, ii]) - u[0, ii] / l * np.sin(x[0, ii])
            fxc[1, 1] = -d
            fuc[1] = np.cos(x[0, ii]) / l
            M = np.zeros((5, 5)); M[:4, :4] = fxc * h; M[:4, 4] = fuc * h
            E = sla.expm(M)
            fxd[:, :, ii] = E[:4, :4]; fud[:, 0, ii] = E[:4, 4]
        return fxd, fud, cx, cu, Q, np.zeros((D, 1)), R

    return f, costfun, df
