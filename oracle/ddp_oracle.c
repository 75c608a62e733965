/*
 * ddp_oracle.c — CPU restatement (plain C99, fp64) of the iLQG hot path of
 * baggepinnen/DifferentialDynamicProgramming.jl v0.5.0.  See ddp_oracle.h for the
 * "test infrastructure only" and "parity unpinned" statements.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Operation ORDER follows the reference where it is visible in the source (e.g.
 * `fu'Vxx*fx` parses as `(fu'Vxx)*fx`); inside each product a plain left-to-right dot
 * product stands in for OpenBLAS, so results agree with Julia to rounding, not bitwise.
 */
#include "ddp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX2(a, b, rows) ((size_t)(a) + (size_t)(rows) * (size_t)(b))

static double jl_clamp(double x, double lo, double hi)
{ /* Base.clamp: ifelse(x > hi, hi, ifelse(x < lo, lo, x)) — NaN passes through */
    return x > hi ? hi : (x < lo ? lo : x);
}

/* Upper Cholesky of the leading nf x nf block of A (ld), reading only the upper triangle
 * (LAPACK dpotrf 'U', what `cholesky(Hermitian(A))` / `cholesky(A).U` run).  R gets the
 * factor (ld ldr, strictly-lower part zeroed).  Returns 0 ok, j+1 on a non-positive pivot. */
static int chol_upper(int nf, const double *A, int lda, double *R, int ldr)
{
    for (int j = 0; j < nf; ++j)
        for (int i = 0; i < nf; ++i) R[IDX2(i, j, ldr)] = 0.0;
    for (int j = 0; j < nf; ++j) {
        double ajj = A[IDX2(j, j, lda)];
        for (int k = 0; k < j; ++k) ajj -= R[IDX2(k, j, ldr)] * R[IDX2(k, j, ldr)];
        if (!(ajj > 0.0)) return j + 1;           /* dpotrf: ajj <= 0 or NaN -> info = j */
        ajj = sqrt(ajj);
        R[IDX2(j, j, ldr)] = ajj;
        for (int i = j + 1; i < nf; ++i) {
            double s = A[IDX2(j, i, lda)];
            for (int k = 0; k < j; ++k) s -= R[IDX2(k, j, ldr)] * R[IDX2(k, i, ldr)];
            R[IDX2(j, i, ldr)] = s / ajj;
        }
    }
    return 0;
}

/* solve (R'R) x = b in place, R upper nf x nf (potrs) */
static void chol_solve(int nf, const double *R, int ldr, double *b)
{
    for (int i = 0; i < nf; ++i) {                /* R' y = b */
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= R[IDX2(k, i, ldr)] * b[k];
        b[i] = s / R[IDX2(i, i, ldr)];
    }
    for (int i = nf - 1; i >= 0; --i) {           /* R x = y */
        double s = b[i];
        for (int k = i + 1; k < nf; ++k) s -= R[IDX2(i, k, ldr)] * b[k];
        b[i] = s / R[IDX2(i, i, ldr)];
    }
}

/* ===================================================================================
 * boxQP — src/boxQP.jl:29-188
 * =================================================================================== */
void ddp_oracle_qp_default_opts(ddp_oracle_qp_opts *o)
{ /* boxQP.jl:30-35 */
    o->maxIter = 100; o->minGrad = 1e-8; o->minRelImprove = 1e-8;
    o->stepDec = 0.6; o->minStep = 1e-22; o->Armijo = 0.1;
}

static double qp_value(int m, const double *H, const double *g, const double *x)
{ /* (x'g + 0.5x'H*x)[1] — boxQP.jl:63,141,146: x'g, then (0.5x')*H, then *x */
    double xg = 0.0, q = 0.0;
    for (int i = 0; i < m; ++i) xg += x[i] * g[i];
    for (int j = 0; j < m; ++j) {
        double t = 0.0;
        for (int i = 0; i < m; ++i) t += (0.5 * x[i]) * H[IDX2(i, j, m)];
        q += t * x[j];
    }
    return xg + q;
}

int ddp_oracle_boxqp(int m, const double *H, const double *g, const double *lower,
                     const double *upper, const double *x0, const ddp_oracle_qp_opts *opts,
                     double *x, double *Hfree, int *free_out, int *nfree_out, int *iters_out)
{
    ddp_oracle_qp_opts dflt;
    if (!opts) { ddp_oracle_qp_default_opts(&dflt); opts = &dflt; }
    int    *clamped = (int *)calloc((size_t)m * 3 + 1, sizeof(int));
    int    *old_clamped = clamped + m, *idx = clamped + 2 * m;
    double *w = (double *)calloc((size_t)m * 6 + (size_t)m * m + 1, sizeof(double));
    double *grad = w, *gc = w + m, *search = w + 2 * m, *xc = w + 3 * m, *rhs = w + 4 * m,
           *xcl = w + 5 * m, *Hsub = w + 6 * m;
    int nfree = m, result = 0, iter = 1;
    double oldvalue = 0.0, value;

    for (int i = 0; i < m; ++i) free_out[i] = 1;                    /* :47-48 */
    for (int i = 0; i < m * m; ++i) Hfree[i] = 0.0;                 /* :54    */
    for (int i = 0; i < m; ++i) x[i] = jl_clamp(x0[i], lower[i], upper[i]);   /* :58 */
    value = qp_value(m, H, g, x);                                   /* :63    */

    while (iter <= opts->maxIter) {                                 /* :71    */
        if (result != 0) break;                                     /* :73-75 */
        if (iter > 1 && (oldvalue - value) < opts->minRelImprove * fabs(oldvalue)) {
            result = 4; break;                                      /* :78-81 */
        }
        oldvalue = value;                                           /* :82    */
        for (int i = 0; i < m; ++i) {                               /* :85 grad = g + H*x */
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += H[IDX2(i, j, m)] * x[j];
            grad[i] = g[i] + s;
        }
        int all_clamped = 1, changed = 0;
        for (int i = 0; i < m; ++i) {                               /* :88-95 (exact ==, Q13) */
            old_clamped[i] = clamped[i];
            clamped[i] = ((x[i] == lower[i]) && (grad[i] > 0)) || ((x[i] == upper[i]) && (grad[i] < 0));
            free_out[i] = !clamped[i];
            if (!clamped[i]) all_clamped = 0;
            if (clamped[i] != old_clamped[i]) changed = 1;
        }
        nfree = 0;
        for (int i = 0; i < m; ++i) if (free_out[i]) idx[nfree++] = i;
        if (all_clamped) { result = 6; break; }                     /* :98-101 */
        if (iter == 1 || changed) {                                 /* :104-117 */
            for (int b = 0; b < nfree; ++b)
                for (int a = 0; a < nfree; ++a) Hsub[IDX2(a, b, m)] = H[IDX2(idx[a], idx[b], m)];
            /* NOTE: Julia's cholesky(::Matrix) throws on a not-exactly-Hermitian
             * argument; here the upper triangle is read as for Hermitian(H[free,free]).  */
            if (chol_upper(nfree, Hsub, m, Hfree, m) != 0) { result = 0; goto done_throw; }
        }
        double gn = 0.0;                                            /* :120-124 */
        for (int a = 0; a < nfree; ++a) gn += grad[idx[a]] * grad[idx[a]];
        gn = sqrt(gn);
        if (gn < opts->minGrad) { result = 5; break; }
        for (int i = 0; i < m; ++i) xcl[i] = x[i] * (double)clamped[i];    /* :127 */
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += H[IDX2(i, j, m)] * xcl[j];
            gc[i] = g[i] + s;
        }
        for (int a = 0; a < nfree; ++a) rhs[a] = gc[idx[a]];        /* :128-129 */
        chol_solve(nfree, Hfree, m, rhs);
        for (int i = 0; i < m; ++i) search[i] = 0.0;
        for (int a = 0; a < nfree; ++a) search[idx[a]] = -rhs[a] - x[idx[a]];
        double sdotg = 0.0;                                         /* :132-135 */
        for (int i = 0; i < m; ++i) sdotg += search[i] * grad[i];
        if (sdotg >= 0) break;                                      /* result stays 0 (Q10) */
        double step = 1.0, vc;                                      /* :138-151 */
        for (int i = 0; i < m; ++i) xc[i] = jl_clamp(x[i] + step * search[i], lower[i], upper[i]);
        vc = qp_value(m, H, g, xc);
        while ((vc - oldvalue) / (step * sdotg) < opts->Armijo) {
            step = step * opts->stepDec;
            for (int i = 0; i < m; ++i) xc[i] = jl_clamp(x[i] + step * search[i], lower[i], upper[i]);
            vc = qp_value(m, H, g, xc);
            if (step < opts->minStep) { result = 2; break; }        /* Q11b */
        }
        for (int i = 0; i < m; ++i) x[i] = xc[i];                   /* :161-163 */
        value = vc;
        iter += 1;
    }
    if (iter == opts->maxIter) result = 1;                          /* :167-169 (Q11) */
done_throw:
    if (nfree_out) *nfree_out = nfree;
    if (iters_out) *iters_out = iter;
    free(clamped); free(w);
    return result;
}

/* ===================================================================================
 * back_pass — src/backward_pass.jl:162-252, shared tail :28-79
 * =================================================================================== */
int ddp_oracle_back_pass(int n, int m, int N,
                         const double *cx, const double *cu,
                         const double *cxx, const double *cxu, const double *cuu,
                         const double *fx, const double *fu,
                         int fx_tv, int cost_tv,
                         double lambda, int regType,
                         const double *lims, const double *u,
                         double *K, double *k, double *Quu,
                         double *Vx, double *Vxx, double *dV)
{
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    size_t wlen = 4 * nn + 8 * nm + 8 * mm + 8 * (size_t)n + 16 * (size_t)m + 16;
    double *w = (double *)calloc(wlen, sizeof(double));
    double *fxV = w, *Vreg = fxV + nn, *fuV = Vreg + nn, *fuVr = fuV + nm, *Qux = fuVr + nm,
           *Quxr = Qux + nm, *QuuF = Quxr + nm, *Rf = QuuF + mm, *Qx = Rf + mm, *Qu = Qx + n,
           *ki = Qu + m, *lower = ki + m, *upper = lower + m, *x0 = upper + m, *Quuk = x0 + m,
           *col = Quuk + m, *Hfree = col + m, *Ki = Hfree + mm, *Qxx = Ki + nm;
    int *freev = (int *)calloc((size_t)m + 1, sizeof(int));
    int diverge = 0;
    /* no-limits test, backward_pass.jl:31 (Q8) */
    const int no_lims = (lims == NULL) || (lims[IDX2(0, 0, m)] > lims[IDX2(0, 1, m)]);

    memset(k, 0, sizeof(double) * (size_t)m * N);                   /* :226-229 */
    memset(K, 0, sizeof(double) * nm * N);
    memset(Vx, 0, sizeof(double) * (size_t)n * N);
    memset(Vxx, 0, sizeof(double) * nn * N);
    memset(Quu, 0, sizeof(double) * mm * N);                        /* `undef` in the reference */
    dV[0] = dV[1] = 0.0;

    /* terminal step, :234-236 / :197-199 (Q6) */
    memcpy(Vx + (size_t)n * (N - 1), cx + (size_t)n * (N - 1), sizeof(double) * n);
    memcpy(Vxx + nn * (N - 1), cxx + (cost_tv ? nn * (N - 1) : 0), sizeof(double) * nn);
    memcpy(Quu + mm * (N - 1), cuu + (cost_tv ? mm * (N - 1) : 0), sizeof(double) * mm);

    for (int i = N - 2; i >= 0; --i) {                              /* for i = N-1:-1:1 */
        const double *fxi = fx + (fx_tv ? nn * i : 0), *fui = fu + (fx_tv ? nm * i : 0);
        const double *cxxi = cxx + (cost_tv ? nn * i : 0), *cxui = cxu + (cost_tv ? nm * i : 0),
                     *cuui = cuu + (cost_tv ? mm * i : 0);
        const double *V = Vxx + nn * (i + 1), *v = Vx + (size_t)n * (i + 1);
        double *Quui = Quu + mm * i;

        for (int a = 0; a < m; ++a) {                               /* Qu = cu[:,i] + fu'Vx[:,i+1] */
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += fui[IDX2(l, a, n)] * v[l];
            Qu[a] = cu[IDX2(a, i, m)] + s;
        }
        for (int j = 0; j < n; ++j) {                               /* Qx = cx[:,i] + fx'Vx[:,i+1] */
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += fxi[IDX2(l, j, n)] * v[l];
            Qx[j] = cx[IDX2(j, i, n)] + s;
        }
        for (int c = 0; c < n; ++c) {                               /* fu'Vxx (m x n), fx'Vxx (n x n) */
            for (int a = 0; a < m; ++a) {
                double s = 0.0, sr = 0.0;
                for (int l = 0; l < n; ++l) {
                    double vlc = V[IDX2(l, c, n)];
                    double vr = vlc + ((regType == 2 && l == c) ? lambda : 0.0);   /* Vxx_reg, :245 */
                    s += fui[IDX2(l, a, n)] * vlc;
                    sr += fui[IDX2(l, a, n)] * vr;
                }
                fuV[IDX2(a, c, m)] = s; fuVr[IDX2(a, c, m)] = sr;
            }
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += fxi[IDX2(l, r, n)] * V[IDX2(l, c, n)];
                fxV[IDX2(r, c, n)] = s;
            }
        }
        for (int j = 0; j < n; ++j)                                 /* Qux, Qux_reg (m x n) :242,246 */
            for (int a = 0; a < m; ++a) {
                double s = 0.0, sr = 0.0;
                for (int l = 0; l < n; ++l) {
                    s += fuV[IDX2(a, l, m)] * fxi[IDX2(l, j, n)];
                    sr += fuVr[IDX2(a, l, m)] * fxi[IDX2(l, j, n)];
                }
                Qux[IDX2(a, j, m)] = cxui[IDX2(j, a, n)] + s;
                Quxr[IDX2(a, j, m)] = cxui[IDX2(j, a, n)] + sr;
            }
        for (int b = 0; b < m; ++b)                                 /* Quu[:,:,i], QuuF :243,247 */
            for (int a = 0; a < m; ++a) {
                double s = 0.0, sr = 0.0;
                for (int l = 0; l < n; ++l) {
                    s += fuV[IDX2(a, l, m)] * fui[IDX2(l, b, n)];
                    sr += fuVr[IDX2(a, l, m)] * fui[IDX2(l, b, n)];
                }
                Quui[IDX2(a, b, m)] = cuui[IDX2(a, b, m)] + s;
                QuuF[IDX2(a, b, m)] = cuui[IDX2(a, b, m)] + sr + ((regType == 1 && a == b) ? lambda : 0.0);
            }
        for (int c = 0; c < n; ++c)                                 /* Qxx :244 */
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += fxV[IDX2(r, l, n)] * fxi[IDX2(l, c, n)];
                Qxx[IDX2(r, c, n)] = cxxi[IDX2(r, c, n)] + s;
            }

        /* ---------------- @end_backward_pass, backward_pass.jl:28-79 ---------------- */
        for (size_t t = 0; t < nm; ++t) Ki[t] = 0.0;
        if (no_lims) {                                              /* :31-42 */
            if (chol_upper(m, QuuF, m, Rf, m) != 0) { diverge = i + 1; goto out; }
            for (int a = 0; a < m; ++a) col[a] = Qu[a];
            chol_solve(m, Rf, m, col);
            for (int a = 0; a < m; ++a) ki[a] = -col[a];
            for (int j = 0; j < n; ++j) {
                for (int a = 0; a < m; ++a) col[a] = Quxr[IDX2(a, j, m)];
                chol_solve(m, Rf, m, col);
                for (int a = 0; a < m; ++a) Ki[IDX2(a, j, m)] = -col[a];
            }
        } else {                                                    /* :44-61 */
            const int ws = (i + 1 < N - 2) ? i + 1 : N - 2;         /* k[:,min(i+1,N-1)] (Q9) */
            int nfree = 0, result;
            for (int a = 0; a < m; ++a) {
                lower[a] = lims[IDX2(a, 0, m)] - u[IDX2(a, i, m)];
                upper[a] = lims[IDX2(a, 1, m)] - u[IDX2(a, i, m)];
                x0[a] = k[IDX2(a, ws, m)];
            }
            result = ddp_oracle_boxqp(m, QuuF, Qu, lower, upper, x0, NULL, ki, Hfree, freev, &nfree, NULL);
            if (result < 1) { diverge = i + 1; goto out; }          /* :53-56 */
            if (nfree > 0) {                                        /* :58-61 */
                int idx[64], nf = 0;
                for (int a = 0; a < m && nf < 64; ++a) if (freev[a]) idx[nf++] = a;
                for (int j = 0; j < n; ++j) {
                    for (int a = 0; a < nf; ++a) col[a] = Quxr[IDX2(idx[a], j, m)];
                    chol_solve(nf, Hfree, m, col);
                    for (int a = 0; a < nf; ++a) Ki[IDX2(idx[a], j, m)] = -col[a];
                }
            }
        }
        /* value update, :64-72 (always the UNregularised Quu, Qux — Q7) */
        double kQuuk = 0.0, kQu = 0.0;
        for (int a = 0; a < m; ++a) {
            double s = 0.0;
            for (int b = 0; b < m; ++b) s += Quui[IDX2(a, b, m)] * ki[b];
            Quuk[a] = s;
        }
        for (int a = 0; a < m; ++a) { kQuuk += ki[a] * Quuk[a]; kQu += ki[a] * Qu[a]; }
        dV[0] += kQu; dV[1] += 0.5 * kQuuk;                         /* :68 */
        double *Vxi = Vx + (size_t)n * i, *Vxxi = Vxx + nn * i;
        for (int j = 0; j < n; ++j) {                               /* :69 */
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int a = 0; a < m; ++a) {
                s1 += Ki[IDX2(a, j, m)] * Quuk[a];
                s2 += Ki[IDX2(a, j, m)] * Qu[a];
                s3 += Qux[IDX2(a, j, m)] * ki[a];
            }
            Vxi[j] = ((Qx[j] + s1) + s2) + s3;
        }
        /* :67,70  KiQuuKi = (K_i'Quu)*K_i */
        for (int c = 0; c < n; ++c)
            for (int r = 0; r < n; ++r) {
                double s1 = 0.0, s2 = 0.0, s3 = 0.0;
                for (int b = 0; b < m; ++b) {
                    double kq = 0.0;                                /* (K'Quu)[r,b] */
                    for (int a = 0; a < m; ++a) kq += Ki[IDX2(a, r, m)] * Quui[IDX2(a, b, m)];
                    s1 += kq * Ki[IDX2(b, c, m)];
                }
                for (int a = 0; a < m; ++a) {
                    s2 += Ki[IDX2(a, r, m)] * Qux[IDX2(a, c, m)];
                    s3 += Qux[IDX2(a, r, m)] * Ki[IDX2(a, c, m)];
                }
                fxV[IDX2(r, c, n)] = ((Qxx[IDX2(r, c, n)] + s1) + s2) + s3;
            }
        for (int c = 0; c < n; ++c)                                 /* :71-72 (Q14) */
            for (int r = 0; r < n; ++r)
                Vxxi[IDX2(r, c, n)] = (fxV[IDX2(r, c, n)] + fxV[IDX2(c, r, n)]) / 2;
        for (int a = 0; a < m; ++a) k[IDX2(a, i, m)] = ki[a];       /* :75-76 */
        for (size_t t = 0; t < nm; ++t) K[nm * i + t] = Ki[t];
    }
out:
    free(w); free(freev);
    return diverge;
}

/* ===================================================================================
 * matrix exponential — what `exp(::Matrix{Float64})` runs (Higham 2005 Padé scaling &
 * squaring), without the gebal balancing step (affects rounding only).
 * Used by the pendcart `df` — src/system_pendcart.jl:148.
 * =================================================================================== */
static void mm_(int n, const double *A, const double *B, double *C)
{
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[IDX2(i, k, n)] * B[IDX2(k, j, n)];
            C[IDX2(i, j, n)] = s;
        }
}
static void gesv_(int n, double *A, double *X)
{ /* solve A X = X (n rhs) by LU with partial pivoting, in place */
    for (int c = 0; c < n; ++c) {
        int p = c; double best = fabs(A[IDX2(c, c, n)]);
        for (int r = c + 1; r < n; ++r) if (fabs(A[IDX2(r, c, n)]) > best) { best = fabs(A[IDX2(r, c, n)]); p = r; }
        if (p != c)
            for (int j = 0; j < n; ++j) {
                double t = A[IDX2(c, j, n)]; A[IDX2(c, j, n)] = A[IDX2(p, j, n)]; A[IDX2(p, j, n)] = t;
                t = X[IDX2(c, j, n)]; X[IDX2(c, j, n)] = X[IDX2(p, j, n)]; X[IDX2(p, j, n)] = t;
            }
        for (int r = c + 1; r < n; ++r) {
            double f = A[IDX2(r, c, n)] / A[IDX2(c, c, n)];
            A[IDX2(r, c, n)] = 0.0;
            for (int j = c + 1; j < n; ++j) A[IDX2(r, j, n)] -= f * A[IDX2(c, j, n)];
            for (int j = 0; j < n; ++j) X[IDX2(r, j, n)] -= f * X[IDX2(c, j, n)];
        }
    }
    for (int j = 0; j < n; ++j)
        for (int r = n - 1; r >= 0; --r) {
            double s = X[IDX2(r, j, n)];
            for (int c = r + 1; c < n; ++c) s -= A[IDX2(r, c, n)] * X[IDX2(c, j, n)];
            X[IDX2(r, j, n)] = s / A[IDX2(r, r, n)];
        }
}

void ddp_oracle_expm(int n, const double *Ain, double *E)
{
    const size_t nn = (size_t)n * n;
    double *w = (double *)calloc(8 * nn, sizeof(double));
    double *A = w, *A2 = w + nn, *P = w + 2 * nn, *U = w + 3 * nn, *V = w + 4 * nn, *T = w + 5 * nn,
           *A4 = w + 6 * nn, *A6 = w + 7 * nn;
    memcpy(A, Ain, nn * sizeof(double));
    double nA = 0.0;
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int i = 0; i < n; ++i) s += fabs(A[IDX2(i, j, n)]); if (s > nA) nA = s; }
    if (nA <= 2.1) {
        static const double C9[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.};
        static const double C7[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
        static const double C5[] = {30240., 15120., 3360., 420., 30., 1.};
        static const double C3[] = {120., 60., 12., 1.};
        const double *C; int nc;
        if (nA > 0.95) { C = C9; nc = 10; } else if (nA > 0.25) { C = C7; nc = 8; }
        else if (nA > 0.015) { C = C5; nc = 6; } else { C = C3; nc = 4; }
        mm_(n, A, A, A2);
        for (size_t t = 0; t < nn; ++t) { P[t] = 0.0; U[t] = 0.0; V[t] = 0.0; }
        for (int i = 0; i < n; ++i) { P[IDX2(i, i, n)] = 1.0; U[IDX2(i, i, n)] = C[1]; V[IDX2(i, i, n)] = C[0]; }
        for (int kk = 1; kk <= nc / 2 - 1; ++kk) {
            mm_(n, P, A2, T); memcpy(P, T, nn * sizeof(double));
            for (size_t t = 0; t < nn; ++t) { U[t] += C[2 * kk + 1] * P[t]; V[t] += C[2 * kk] * P[t]; }
        }
        mm_(n, A, U, T); memcpy(U, T, nn * sizeof(double));
        for (size_t t = 0; t < nn; ++t) { E[t] = V[t] + U[t]; T[t] = V[t] - U[t]; }
        gesv_(n, T, E);
    } else {
        static const double CC[] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                                    129060195264000., 10559470521600., 670442572800., 33522128640., 1323241920.,
                                    40840800., 960960., 16380., 182., 1.};
        double s = log2(nA / 5.4); int si = 0;
        if (s > 0) { si = (int)ceil(s); double sc = ldexp(1.0, si); for (size_t t = 0; t < nn; ++t) A[t] /= sc; }
        mm_(n, A, A, A2); mm_(n, A2, A2, A4); mm_(n, A2, A4, A6);
        for (size_t t = 0; t < nn; ++t) P[t] = CC[13] * A6[t] + CC[11] * A4[t] + CC[9] * A2[t];
        mm_(n, A6, P, T);
        for (size_t t = 0; t < nn; ++t) T[t] += CC[7] * A6[t] + CC[5] * A4[t] + CC[3] * A2[t];
        for (int i = 0; i < n; ++i) T[IDX2(i, i, n)] += CC[1];
        mm_(n, A, T, U);
        for (size_t t = 0; t < nn; ++t) P[t] = CC[12] * A6[t] + CC[10] * A4[t] + CC[8] * A2[t];
        mm_(n, A6, P, V);
        for (size_t t = 0; t < nn; ++t) V[t] += CC[6] * A6[t] + CC[4] * A4[t] + CC[2] * A2[t];
        for (int i = 0; i < n; ++i) V[IDX2(i, i, n)] += CC[0];
        for (size_t t = 0; t < nn; ++t) { E[t] = V[t] + U[t]; T[t] = V[t] - U[t]; }
        gesv_(n, T, E);
        for (int q = 0; q < si; ++q) { mm_(n, E, E, T); memcpy(E, T, nn * sizeof(double)); }
    }
    free(w);
}

/* ===================================================================================
 * problem families: the closures of src/demo_linear.jl:35-50 and
 * src/system_pendcart.jl:83-154
 * =================================================================================== */
int ddp_oracle_cost_len(const ddp_oracle_problem *p)
{
    return p->kind == DDP_ORACLE_PENDCART ? p->N + 1 : p->N;
}

void ddp_oracle_f(const ddp_oracle_problem *p, const double *x, double *u, int i, double *xn)
{
    const int n = p->n, m = p->m;
    for (int a = 0; a < m; ++a) if (isnan(u[a])) u[a] = 0.0;       /* u[isnan.(u)] .= 0 */
    if (p->kind == DDP_ORACLE_LQ) {                                 /* demo_linear.jl:42-46 */
        const double *A = p->A + (p->dyn_tv ? (size_t)n * n * i : 0);
        const double *Bm = p->Bm + (p->dyn_tv ? (size_t)n * m * i : 0);
        for (int r = 0; r < n; ++r) {
            double s = 0.0, t = 0.0;
            for (int c = 0; c < n; ++c) s += A[IDX2(r, c, n)] * x[c];
            for (int a = 0; a < m; ++a) t += Bm[IDX2(r, a, n)] * u[a];
            xn[r] = s + t;
        }
    } else {                                                        /* system_pendcart.jl:83-89 */
        const double g = p->g, l = p->l, h = p->h, d = p->d;
        xn[0] = x[0] + h * x[1];
        xn[1] = x[1] + h * (-g / l * sin(x[0]) + u[0] / l * cos(x[0]) - d * x[1]);
        xn[2] = x[2] + h * x[3];
        xn[3] = x[3] + h * u[0];
    }
}

static double quad_(int n, const double *M, const double *v)
{ /* sum(v .* (M*v)) */
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = 0; j < n; ++j) t += M[IDX2(i, j, n)] * v[j];
        s += v[i] * t;
    }
    return s;
}

void ddp_oracle_costfun(const ddp_oracle_problem *p, const double *X, const double *U, double *c)
{
    const int n = p->n, m = p->m, N = p->N;
    if (p->kind == DDP_ORACLE_LQ) {
        /* demo_linear.jl:49: 0.5*sum(x.*(Q*x)) + 0.5*sum(u.*(R*u)) — returned per time step;
         * sum(c) equals the reference's scalar up to summation order. */
        for (int t = 0; t < N; ++t)
            c[t] = 0.5 * quad_(n, p->Q, X + (size_t)n * t) + 0.5 * quad_(m, p->R, U + (size_t)m * t);
    } else {                                                        /* system_pendcart.jl:97-106 */
        double dx[4];
        for (int t = 0; t < N; ++t) {
            for (int i = 0; i < 4; ++i) dx[i] = X[IDX2(i, t, 4)] - p->goal[i];
            c[t] = 0.5 * (quad_(4, p->Q, dx) + U[t] * p->R[0] * U[t]);
        }
        for (int i = 0; i < 4; ++i) dx[i] = X[IDX2(i, N - 1, 4)] - p->goal[i];
        c[N] = 0.5 * (quad_(4, p->Q, dx) + 0.0 * p->R[0] * 0.0);    /* cost_quadratic(x[:,end],[0.0]) */
    }
}

void ddp_oracle_df(const ddp_oracle_problem *p, const double *X, double *U,
                   double *cx, double *cu, double *fx, double *fu)
{
    const int n = p->n, m = p->m, N = p->N;
    for (size_t t = 0; t < (size_t)m * N; ++t) if (isnan(U[t])) U[t] = 0.0;
    if (p->kind == DDP_ORACLE_LQ) {                                 /* demo_linear.jl:35-41 */
        for (int t = 0; t < N; ++t) {
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += p->Q[IDX2(i, j, n)] * X[IDX2(j, t, n)];
                cx[IDX2(i, t, n)] = s;
            }
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int b = 0; b < m; ++b) s += p->R[IDX2(a, b, m)] * U[IDX2(b, t, m)];
                cu[IDX2(a, t, m)] = s;
            }
        }
    } else {                                                        /* system_pendcart.jl:112-116,137-154 */
        const double g = p->g, l = p->l, h = p->h, d = p->d;
        for (int t = 0; t < N; ++t) {
            double dx[4];
            for (int i = 0; i < 4; ++i) dx[i] = X[IDX2(i, t, 4)] - p->goal[i];
            for (int i = 0; i < 4; ++i) {
                double s = 0.0;
                for (int j = 0; j < 4; ++j) s += p->Q[IDX2(i, j, 4)] * dx[j];
                cx[IDX2(i, t, 4)] = s;
            }
            cu[t] = p->R[0] * U[t];
            double M[25], E[25];
            memset(M, 0, sizeof M);
            const double th = X[IDX2(0, t, 4)];
            /* fxc = [0 1 0 0; a21 -d 0 0; 0 0 0 1; 0 0 0 0], fuc = [0, cos/l, 0, 1] (:130-147) */
            M[IDX2(0, 1, 5)] = 1.0 * h;
            M[IDX2(1, 0, 5)] = (-g / l * cos(th) - U[t] / l * sin(th)) * h;
            M[IDX2(1, 1, 5)] = (-d) * h;
            M[IDX2(2, 3, 5)] = 1.0 * h;
            M[IDX2(1, 4, 5)] = (cos(th) / l) * h;
            M[IDX2(3, 4, 5)] = 1.0 * h;
            ddp_oracle_expm(5, M, E);                               /* :148 */
            for (int c = 0; c < 4; ++c)
                for (int r = 0; r < 4; ++r) fx[(size_t)16 * t + IDX2(r, c, 4)] = E[IDX2(r, c, 5)];
            for (int r = 0; r < 4; ++r) fu[(size_t)4 * t + r] = E[IDX2(r, 4, 5)];
        }
    }
}

/* ===================================================================================
 * forward_pass — src/forward_pass.jl:9-33  (diff == `-`, or `-` with the coordinates of p->diff_wrap wrapped to [-pi, pi])
 * =================================================================================== */
void ddp_oracle_forward_pass(const ddp_oracle_problem *p,
                             const double *K, const double *k,
                             const double *x0, const double *u, const double *x,
                             double alpha, const double *lims,
                             double *xnew, double *unew, double *cnew)
{
    const int n = p->n, m = p->m, N = p->N;
    double *xn = (double *)calloc((size_t)n + 1, sizeof(double));
    memcpy(xnew, x0, sizeof(double) * n);                           /* :12 */
    memcpy(unew, u, sizeof(double) * (size_t)m * N);                /* :13 */
    for (int i = 0; i < N; ++i) {                                   /* :16 */
        double *ui = unew + (size_t)m * i;
        const double *xi = xnew + (size_t)n * i;
        if (K && k) {                                               /* :17-21 (Q16) */
            for (int a = 0; a < m; ++a) ui[a] += k[IDX2(a, i, m)] * alpha;
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int j = 0; j < n; ++j) {
                    double dxj = xi[j] - x[IDX2(j, i, n)];                  /* :19 diff(xnew[:,i], x[:,i]) */
                    if (j < 32 && ((p->diff_wrap >> j) & 1u)) dxj = remainder(dxj, 6.283185307179586);
                    s += K[(size_t)m * n * i + IDX2(a, j, m)] * dxj;
                }
                ui[a] += s;
            }
        }
        if (lims)                                                   /* :22-24 */
            for (int a = 0; a < m; ++a) ui[a] = jl_clamp(ui[a], lims[IDX2(a, 0, m)], lims[IDX2(a, 1, m)]);
        ddp_oracle_f(p, xi, ui, i, xn);                             /* :25 (also at i == N, discarded) */
        if (i < N - 1) memcpy(xnew + (size_t)n * (i + 1), xn, sizeof(double) * n);
    }
    ddp_oracle_costfun(p, xnew, unew, cnew);                        /* :30 */
    free(xn);
}

/* ===================================================================================
 * iLQG outer loop — src/iLQG.jl:143-341
 * =================================================================================== */
void ddp_oracle_ilqg_default_opts(ddp_oracle_ilqg_opts *o)
{ /* iLQG.jl:143-163 */
    static double alpha_default[11];
    for (int i = 0; i < 11; ++i) alpha_default[i] = pow(10.0, 0.0 + (-3.0 - 0.0) * i / 10.0);
    o->lambda = 1.0; o->dlambda = 1.0; o->lambda_factor = 1.6; o->lambda_max = 1e10; o->lambda_min = 1e-6;
    o->tol_fun = 1e-7; o->tol_grad = 1e-4; o->max_iter = 500; o->regType = 1; o->reduce_ratio_min = 0.0;
    o->n_alpha = 11; o->alpha = alpha_default;
}

static double sum_(const double *v, int len) { double s = 0.0; for (int i = 0; i < len; ++i) s += v[i]; return s; }

static int ilqg_impl(const ddp_oracle_problem *p, const ddp_oracle_ilqg_opts *o,
                     const double *x0, const double *u0, const double *lims,
                     double *x, double *u, double *K, double *k, double *Quu,
                     double *Vx, double *Vxx, double *cost,
                     ddp_oracle_ilqg_result *res,
                     int trace_cap, double *tr_cost, double *tr_lambda, double *tr_alpha,
                     double *tr_gnorm, int prerolled, const double *cost0,
                     double *tr_dlambda, double *tr_improvement, double *tr_ratio)
{
    const int n = p->n, m = p->m, N = p->N, CL = ddp_oracle_cost_len(p);
    const size_t nN = (size_t)n * N, mN = (size_t)m * N;
    double *xnew = (double *)calloc(nN, sizeof(double)), *unew = (double *)calloc(mN, sizeof(double)),
           *costnew = (double *)calloc((size_t)CL, sizeof(double)), *cx = (double *)calloc(nN, sizeof(double)),
           *cu = (double *)calloc(mN, sizeof(double)), *us = (double *)calloc(mN, sizeof(double));
    double *fxb = NULL, *fub = NULL, *cxu = (double *)calloc((size_t)n * m, sizeof(double));
    const double *fx, *fu; int fx_tv;
    if (p->kind == DDP_ORACLE_PENDCART) {
        fxb = (double *)calloc((size_t)16 * N, sizeof(double)); fub = (double *)calloc((size_t)4 * N, sizeof(double));
        fx = fxb; fu = fub; fx_tv = 1;
    } else { fx = p->A; fu = p->Bm; fx_tv = p->dyn_tv; }
    double lambda = o->lambda, dlambda = o->dlambda, dV[2] = {0, 0}, g_norm = 0.0;
    int status = DDP_EXIT_RUNNING, n_bp = 0, n_fp = 0, tl = 0;

    /* --- initial trajectory, iLQG.jl:181-192 (x0 is a single column) */
    int diverge0 = 1;
    if (prerolled) {                                                /* x0 is [n,N]: iLQG.jl:193-197 */
        memcpy(x, x0, sizeof(double) * nN);
        memcpy(u, u0, sizeof(double) * mN);
        if (cost0) memcpy(cost, cost0, sizeof(double) * (size_t)CL);
        else ddp_oracle_costfun(p, x, u, cost);                     /* isempty(cost) && (cost = costfun(x, u)) */
        diverge0 = 0;
    }
    for (int ai = 0; !prerolled && ai < o->n_alpha; ++ai) {
        for (size_t t = 0; t < mN; ++t) us[t] = o->alpha[ai] * u0[t];
        ddp_oracle_forward_pass(p, NULL, NULL, x0, us, NULL, 1.0, lims, x, unew, cost);
        int ok = 1;
        for (size_t t = 0; t < nN; ++t) if (!(fabs(x[t]) < 1e8)) { ok = 0; break; }
        if (ok) { memcpy(u, unew, sizeof(double) * mN); diverge0 = 0; break; }
    }
    memset(K, 0, sizeof(double) * (size_t)m * n * N); memset(k, 0, sizeof(double) * mN);
    memset(Vx, 0, sizeof(double) * nN); memset(Vxx, 0, sizeof(double) * (size_t)n * n * N);
    memset(Quu, 0, sizeof(double) * (size_t)m * m * N);
    int iter = 1, accepted_iter = 1;
    if (diverge0) { status = DDP_EXIT_INIT_DIVERGED; goto finish; }  /* :205-210 */

    int flg_change = 1;
    while (accepted_iter <= o->max_iter) {                          /* :222 */
        double reduce_ratio = 0.0, dcost = 0.0, expected = 0.0, alpha_used = NAN;
        (void)reduce_ratio;
        if (flg_change) {                                           /* STEP 1, :225-229 */
            ddp_oracle_df(p, x, u, cx, cu, fxb, fub);
            flg_change = 0;
        }
        int back_pass_done = 0, diverge;                            /* STEP 2, :234-251 */
        while (!back_pass_done) {
            diverge = ddp_oracle_back_pass(n, m, N, cx, cu, p->Q, cxu, p->R, fx, fu, fx_tv, 0, lambda,
                                           o->regType, lims, u, K, k, Quu, Vx, Vxx, dV);
            ++n_bp;
            if (diverge > 0) {
                double dl_old = dlambda;                            /* tuple assignment (Q1) */
                dlambda = fmax(dl_old * o->lambda_factor, o->lambda_factor);
                lambda = fmax(lambda * dl_old, o->lambda_min);
                if (lambda > o->lambda_max) break;
                continue;
            }
            back_pass_done = 1;
        }
        {                                                           /* :254-261 */
            double s = 0.0;
            for (int t = 0; t < N; ++t) {
                double mx = 0.0;
                for (int a = 0; a < m; ++a) {
                    double r = fabs(k[IDX2(a, t, m)]) / (fabs(u[IDX2(a, t, m)]) + 1.0);
                    if (a == 0 || r > mx || isnan(r)) mx = r;
                }
                s += mx;
            }
            g_norm = s / N;
        }
        if (tl < trace_cap && tr_gnorm) tr_gnorm[tl] = g_norm;
        if (g_norm < o->tol_grad && lambda < 1e-5) { status = DDP_EXIT_GRAD; break; }

        int fwd_pass_done = 0;                                      /* STEP 3, :264-283 */
        if (back_pass_done) {
            const double c0 = sum_(cost, CL);
            for (int ai = 0; ai < o->n_alpha; ++ai) {
                const double a = o->alpha[ai];
                ddp_oracle_forward_pass(p, K, k, x0, u, x, a, lims, xnew, unew, costnew);
                ++n_fp;
                alpha_used = a;
                dcost = c0 - sum_(costnew, CL);
                expected = -a * (dV[0] + a * dV[1]);
                if (expected > 0) reduce_ratio = dcost / expected;
                else reduce_ratio = (dcost > 0) - (dcost < 0);      /* sign(Δcost) */
                if (reduce_ratio > o->reduce_ratio_min) { fwd_pass_done = 1; break; }
            }
        }
        if (fwd_pass_done) {                                        /* STEP 4, :293-310 */
            dlambda = fmin(dlambda / o->lambda_factor, 1.0 / o->lambda_factor);
            lambda = fmax(lambda * dlambda, o->lambda_min);         /* sequential: NEW dlambda (Q1,Q2) */
            memcpy(x, xnew, sizeof(double) * nN); memcpy(u, unew, sizeof(double) * mN);
            memcpy(cost, costnew, sizeof(double) * CL);
            memcpy(k, u, sizeof(double) * mN);                      /* traj_new.k = copy(u) (Q3) */
            flg_change = 1;
            if (dcost < o->tol_fun) { status = DDP_EXIT_COST; break; }
            accepted_iter += 1;
        } else {                                                    /* :311-323 */
            alpha_used = NAN;
            double dl_old = dlambda;
            dlambda = fmax(dl_old * o->lambda_factor, o->lambda_factor);
            lambda = fmax(lambda * dl_old, o->lambda_min);
            if (lambda > o->lambda_max) { status = DDP_EXIT_LAMBDA; break; }
        }
        if (tl < trace_cap) {                                       /* :325-330 */
            if (tr_cost) tr_cost[tl] = sum_(cost, CL);
            if (tr_lambda) tr_lambda[tl] = lambda;
            if (tr_alpha) tr_alpha[tl] = alpha_used;
            if (tr_dlambda) tr_dlambda[tl] = dlambda;               /* :326 */
            if (tr_improvement) tr_improvement[tl] = dcost;         /* :328 */
            if (tr_ratio) tr_ratio[tl] = reduce_ratio;              /* :330 */
            ++tl;
        }
        iter += 1;
    }
    if (status == DDP_EXIT_RUNNING) status = DDP_EXIT_MAXITER;
finish:
    if (res) {
        res->status = status; res->iter = iter; res->accepted_iter = accepted_iter;
        res->n_backpass = n_bp; res->n_forward = n_fp; res->lambda = lambda; res->dlambda = dlambda;
        res->g_norm = g_norm; res->dV[0] = dV[0]; res->dV[1] = dV[1]; res->trace_len = tl;
    }
    free(xnew); free(unew); free(costnew); free(cx); free(cu); free(us); free(cxu);
    if (fxb) free(fxb);
    if (fub) free(fub);
    return status;
}

int ddp_oracle_ilqg(const ddp_oracle_problem *p, const ddp_oracle_ilqg_opts *o,
                    const double *x0, const double *u0, const double *lims,
                    double *x, double *u, double *K, double *k, double *Quu,
                    double *Vx, double *Vxx, double *cost,
                    ddp_oracle_ilqg_result *res,
                    int trace_cap, double *tr_cost, double *tr_lambda, double *tr_alpha,
                    double *tr_gnorm)
{
    return ilqg_impl(p, o, x0, u0, lims, x, u, K, k, Quu, Vx, Vxx, cost, res, trace_cap, tr_cost, tr_lambda, tr_alpha, tr_gnorm, 0, NULL,
                     NULL, NULL, NULL);
}

/* all seven per-iteration trace keys of iLQG.jl:257,325-330: tr7[7,trace_cap] rows λ, dλ, α (NaN: no step), improvement, cost,
 * reduce_ratio, grad_norm */
int ddp_oracle_ilqg_trace7(const ddp_oracle_problem *p, const ddp_oracle_ilqg_opts *o,
                           const double *x0, const double *u0, const double *lims,
                           double *x, double *u, double *K, double *k, double *Quu,
                           double *Vx, double *Vxx, double *cost, ddp_oracle_ilqg_result *res, int trace_cap, double *tr7)
{
    double *t[7];
    for (int c = 0; c < 7; ++c) t[c] = (double *)calloc((size_t)(trace_cap > 0 ? trace_cap : 1), sizeof(double));
    const int st = ilqg_impl(p, o, x0, u0, lims, x, u, K, k, Quu, Vx, Vxx, cost, res, trace_cap, t[4], t[0], t[2], t[6], 0, NULL, t[1], t[3], t[5]);
    for (int i = 0; i < trace_cap; ++i)
        for (int c = 0; c < 7; ++c) tr7[c + 7 * i] = t[c][i];
    for (int c = 0; c < 7; ++c) free(t[c]);
    return st;
}

/* pre-rolled initial trajectory x0[n,N] (iLQG.jl:193-197): no initial rollout; cost0[cost_len] or NULL (= costfun(x0,u0)) */
int ddp_oracle_ilqg_prerolled(const ddp_oracle_problem *p, const ddp_oracle_ilqg_opts *o,
                              const double *x0, const double *u0, const double *cost0, const double *lims,
                              double *x, double *u, double *K, double *k, double *Quu,
                              double *Vx, double *Vxx, double *cost, ddp_oracle_ilqg_result *res)
{
    return ilqg_impl(p, o, x0, u0, lims, x, u, K, k, Quu, Vx, Vxx, cost, res, 0, NULL, NULL, NULL, NULL, 1, cost0, NULL, NULL, NULL);
}

int ddp_oracle_pass_batch_lq(const ddp_oracle_problem *p, int B,
                             const double *cx, const double *cu, const double *cxx,
                             const double *cxu, const double *cuu, double lambda, int regType,
                             const double *x0, const double *u, const double *x, double alpha,
                             double *K, double *k, double *Quu, double *Vx, double *Vxx,
                             double *dV, double *xnew, double *unew, double *cnew)
{
    const size_t n = p->n, m = p->m, N = p->N;
    int ndiv = 0;
    for (int b = 0; b < B; ++b) {
        int d = ddp_oracle_back_pass((int)n, (int)m, (int)N, cx + n * N * b, cu + m * N * b, cxx, cxu, cuu,
                                     p->A, p->Bm, p->dyn_tv, 0, lambda, regType, NULL, u + m * N * b,
                                     K + m * n * N * b, k + m * N * b, Quu + m * m * N * b,
                                     Vx + n * N * b, Vxx + n * n * N * b, dV + 2 * b);
        ndiv += d > 0;
        ddp_oracle_forward_pass(p, K + m * n * N * b, k + m * N * b, x0 + n * b, u + m * N * b,
                                x + n * N * b, alpha, NULL, xnew + n * N * b, unew + m * N * b,
                                cnew + N * b);
    }
    return ndiv;
}

/* the same, nrep times over (one long call per host thread for the all-cores figure of bench.py) */
int ddp_oracle_pass_batch_lq_rep(const ddp_oracle_problem *p, int B, int nrep,
                                 const double *cx, const double *cu, const double *cxx,
                                 const double *cxu, const double *cuu, double lambda, int regType,
                                 const double *x0, const double *u, const double *x, double alpha,
                                 double *K, double *k, double *Quu, double *Vx, double *Vxx,
                                 double *dV, double *xnew, double *unew, double *cnew)
{
    int nd = 0;
    for (int r = 0; r < nrep; ++r)
        nd += ddp_oracle_pass_batch_lq(p, B, cx, cu, cxx, cxu, cuu, lambda, regType, x0, u, x, alpha, K, k, Quu, Vx, Vxx, dV, xnew, unew, cnew);
    return nd;
}
