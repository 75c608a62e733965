/*
 * ddp_oracle_kl.c — CPU restatement (plain C99, fp64) of the KL-constrained path of
 * baggepinnen/DifferentialDynamicProgramming.jl v0.5.0 (BASELINE config 5):
 *   back_pass_gps        src/backward_pass.jl:259-350
 *   ∇kl, kl_div_wiki, calc_η (scalar kl_step), geom      src/klutils.jl:8-23,70-130,154-155
 *   forward_covariance   src/forward_pass.jl:37-56
 *   iLQGkl, single-constraint branch                      src/iLQGkl.jl:25-178,234-252
 *
 * TEST INFRASTRUCTURE ONLY (see ddp_oracle.h).  PARITY UNPINNED: the reference has no numeric fixture for this
 * path (test/runtests.jl:9 is a smoke run) and `df(model,·)`, `covariance(model,·)` live in the un-vendored
 * dependency LinearTimeVaryingModelsBase (Project.toml compat 0.2.1, no Manifest).  Here the model is the triple
 * (fx[n,n,N], fu[n,m,N], R1[n,n]) handed in by the caller; ddp_oracle_model_covariance() is this build's documented
 * choice for `covariance`: the empirical covariance of the one-step prediction residuals (the inline comment at
 * forward_pass.jl:42).  The per-time-step branch (constrain_per_step, iLQGkl.jl:180-232) is not restated.
 */
#include "ddp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX2(i, j, ld) ((size_t)(i) + (size_t)(ld) * (size_t)(j))

/* ---- small dense helpers ------------------------------------------------------------------------------- */
static int chol_upper_(int m, const double *A, double *R)
{   /* R'R = Hermitian(A) (upper triangle read); 0 ok, else failing 1-based pivot */
    for (int j = 0; j < m; ++j) {
        double d = A[IDX2(j, j, m)];
        for (int k = 0; k < j; ++k) d -= R[IDX2(k, j, m)] * R[IDX2(k, j, m)];
        if (!(d > 0.0)) return j + 1;
        R[IDX2(j, j, m)] = sqrt(d);
        for (int c = j + 1; c < m; ++c) {
            double s = A[IDX2(j, c, m)];
            for (int k = 0; k < j; ++k) s -= R[IDX2(k, j, m)] * R[IDX2(k, c, m)];
            R[IDX2(j, c, m)] = s / R[IDX2(j, j, m)];
        }
        for (int r = j + 1; r < m; ++r) R[IDX2(r, j, m)] = 0.0;
    }
    return 0;
}
static void chol_solve_ld(int nf, const double *R, int ld, double *b)
{   /* b <- (R'R)\b with the leading nf x nf block of R (leading dimension ld) */
    for (int i = 0; i < nf; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= R[IDX2(k, i, ld)] * b[k];
        b[i] = s / R[IDX2(i, i, ld)];
    }
    for (int i = nf - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < nf; ++k) s -= R[IDX2(i, k, ld)] * b[k];
        b[i] = s / R[IDX2(i, i, ld)];
    }
}
/* LU with partial pivoting (what inv / logdet of a general Matrix run): A is overwritten, returns the sign of
 * the permutation (0 if singular); piv[c] = row swapped with c */
static int lu_(int m, double *A, int *piv)
{
    int sgn = 1;
    for (int c = 0; c < m; ++c) {
        int p = c; double best = fabs(A[IDX2(c, c, m)]);
        for (int r = c + 1; r < m; ++r) if (fabs(A[IDX2(r, c, m)]) > best) { best = fabs(A[IDX2(r, c, m)]); p = r; }
        piv[c] = p;
        if (best == 0.0) return 0;
        if (p != c) {
            sgn = -sgn;
            for (int j = 0; j < m; ++j) { double t = A[IDX2(c, j, m)]; A[IDX2(c, j, m)] = A[IDX2(p, j, m)]; A[IDX2(p, j, m)] = t; }
        }
        for (int r = c + 1; r < m; ++r) {
            const double f = A[IDX2(r, c, m)] / A[IDX2(c, c, m)];
            A[IDX2(r, c, m)] = f;
            for (int j = c + 1; j < m; ++j) A[IDX2(r, j, m)] -= f * A[IDX2(c, j, m)];
        }
    }
    return sgn;
}
static void inv_(int m, const double *Ain, double *X)
{   /* X = inv(Ain) via LU (getrf/getri semantics up to rounding order) */
    double *A = (double *)malloc(sizeof(double) * (size_t)m * m);
    int *piv = (int *)malloc(sizeof(int) * (size_t)m);
    memcpy(A, Ain, sizeof(double) * (size_t)m * m);
    const int sgn = lu_(m, A, piv);
    for (int j = 0; j < m; ++j) {
        double *x = X + (size_t)m * j;
        for (int r = 0; r < m; ++r) x[r] = (r == j) ? 1.0 : 0.0;
        if (sgn == 0) { for (int r = 0; r < m; ++r) x[r] = NAN; continue; }
        for (int c = 0; c < m; ++c) { const int p = piv[c]; if (p != c) { double t = x[c]; x[c] = x[p]; x[p] = t; } }
        for (int r = 0; r < m; ++r) for (int c = 0; c < r; ++c) x[r] -= A[IDX2(r, c, m)] * x[c];
        for (int r = m - 1; r >= 0; --r) {
            double s = x[r];
            for (int c = r + 1; c < m; ++c) s -= A[IDX2(r, c, m)] * x[c];
            x[r] = s / A[IDX2(r, r, m)];
        }
    }
    free(A); free(piv);
}
static double logdet_(int m, const double *Ain)
{   /* logdet of a general Matrix: LU; negative determinant is a DomainError upstream -> NaN here */
    double *A = (double *)malloc(sizeof(double) * (size_t)m * m);
    int *piv = (int *)malloc(sizeof(int) * (size_t)m);
    memcpy(A, Ain, sizeof(double) * (size_t)m * m);
    int sgn = lu_(m, A, piv);
    double s = 0.0;
    if (sgn == 0) s = -INFINITY;
    else for (int c = 0; c < m; ++c) { const double d = A[IDX2(c, c, m)]; if (d < 0.0) sgn = -sgn; s += log(fabs(d)); }
    free(A); free(piv);
    return (sgn < 0) ? NAN : s;
}

/* ===================================================================================
 * ∇kl — src/klutils.jl:8-23.  Outputs cx[n,T], cu[m,T], cxx[n,n,T], cxu[m,n,T] (sic: m x n), cuu[m,m,T].
 * =================================================================================== */
void ddp_oracle_kl_terms(int n, int m, int T, const double *K, const double *k, const double *Sigmai,
                         double *cx, double *cu, double *cxx, double *cxu, double *cuu)
{
    const size_t nm = (size_t)n * m, mm = (size_t)m * m, nn = (size_t)n * n;
    double *SiK = (double *)malloc(sizeof(double) * nm), *Sik = (double *)malloc(sizeof(double) * (size_t)m);
    for (int t = 0; t < T; ++t) {
        const double *Kt = K + nm * t, *kt = k + (size_t)m * t, *Si = Sigmai + mm * t;
        for (int j = 0; j < n; ++j)                                        /* Σi*K */
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int b = 0; b < m; ++b) s += Si[IDX2(a, b, m)] * Kt[IDX2(b, j, m)];
                SiK[IDX2(a, j, m)] = s;
            }
        for (int a = 0; a < m; ++a) {                                      /* Σi*k */
            double s = 0.0;
            for (int b = 0; b < m; ++b) s += Si[IDX2(a, b, m)] * kt[b];
            Sik[a] = s;
        }
        for (int j = 0; j < n; ++j) {                                      /* cx = K'Σi k  (:16) */
            double s = 0.0;
            for (int a = 0; a < m; ++a) s += Kt[IDX2(a, j, m)] * Sik[a];
            cx[IDX2(j, t, n)] = s;
        }
        for (int a = 0; a < m; ++a) cu[IDX2(a, t, m)] = -Sik[a];           /* cu = -Σi k  (:17) */
        for (int c = 0; c < n; ++c)                                        /* cxx = K'Σi K (:18) */
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
                for (int a = 0; a < m; ++a) s += Kt[IDX2(a, r, m)] * SiK[IDX2(a, c, m)];
                cxx[nn * t + IDX2(r, c, n)] = s;
            }
        for (size_t e = 0; e < mm; ++e) cuu[mm * t + e] = Si[e];           /* cuu = Σi     (:19) */
        for (size_t e = 0; e < nm; ++e) cxu[nm * t + e] = -SiK[e];         /* cxu = -Σi K  (:20), m x n */
    }
    free(SiK); free(Sik);
}

/* ===================================================================================
 * back_pass_gps — src/backward_pass.jl:259-350.  All cost/dynamics arrays are 3-D ([..,N]).
 * eta: eta_tv == 0 -> one value; 1 -> eta[N] (ηbracket[2,i]).  lims NULL or [m,2].
 * Outputs zero-filled where the reference leaves zeros() or undef.  Returns diverge (1-based) or 0.
 * =================================================================================== */
int ddp_oracle_back_pass_gps(int n, int m, int N,
                             const double *cx, const double *cu, const double *cxx, const double *cxu, const double *cuu,
                             const double *fx, const double *fu, const double *lims, const double *u,
                             const double *cxkl, const double *cukl, const double *cxxkl, const double *cxukl,
                             const double *cuukl, const double *eta, int eta_tv,
                             double *K, double *k, double *Quu, double *Quui, double *Vx, double *Vxx, double *dV)
{
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    double *w = (double *)calloc(3 * nn + 4 * nm + 4 * mm + 2 * (size_t)n + 8 * (size_t)m + 8, sizeof(double));
    double *fxV = w, *Qxx = fxV + nn, *M = Qxx + nn, *fuV = M + nn, *Qux = fuV + nm, *Ki = Qux + nm, *tmpnm = Ki + nm,
           *Rf = tmpnm + nm, *Hfree = Rf + mm, *Qs = Hfree + mm, *Qx = Qs + mm, *Qu = Qx + n, *ki = Qu + m, *lower = ki + m,
           *upper = lower + m, *x0 = upper + m, *Quuk = x0 + m, *col = Quuk + m;
    int *freev = (int *)calloc((size_t)m + 1, sizeof(int));
    int diverge = 0;
    const int no_lims = (lims == NULL) || (lims[IDX2(0, 0, m)] > lims[IDX2(0, 1, m)]);     /* :303 */
    (void)Qs;
    memset(k, 0, sizeof(double) * (size_t)m * N);
    memset(K, 0, sizeof(double) * nm * N);
    memset(Vx, 0, sizeof(double) * (size_t)n * N);
    memset(Vxx, 0, sizeof(double) * nn * N);
    memset(Quu, 0, sizeof(double) * mm * N);
    memset(Quui, 0, sizeof(double) * mm * N);
    dV[0] = dV[1] = 0.0;
    {   /* terminal step :280-283 */
        const double etaN = eta_tv ? eta[N - 1] : eta[0];
        memcpy(Vx + (size_t)n * (N - 1), cx + (size_t)n * (N - 1), sizeof(double) * n);
        memcpy(Vxx + nn * (N - 1), cxx + nn * (N - 1), sizeof(double) * nn);
        for (size_t e = 0; e < mm; ++e) Quu[mm * (N - 1) + e] = cuu[mm * (N - 1) + e] / etaN + cuukl[mm * (N - 1) + e];
        inv_(m, Quu + mm * (N - 1), Quui + mm * (N - 1));
    }
    for (int i = N - 2; i >= 0; --i) {
        const double *fxi = fx + nn * i, *fui = fu + nm * i, *V = Vxx + nn * (i + 1), *v = Vx + (size_t)n * (i + 1);
        const double et = eta_tv ? eta[i] : eta[0];
        double *Q = Quu + mm * i;
        for (int a = 0; a < m; ++a) {                                        /* Qu :286 */
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += fui[IDX2(l, a, n)] * v[l];
            Qu[a] = cu[IDX2(a, i, m)] + s;
        }
        for (int j = 0; j < n; ++j) {                                        /* Qx :287 */
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += fxi[IDX2(l, j, n)] * v[l];
            Qx[j] = cx[IDX2(j, i, n)] + s;
        }
        for (int c = 0; c < n; ++c) {                                        /* fu'Vxx, fx'Vxx */
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += fui[IDX2(l, a, n)] * V[IDX2(l, c, n)];
                fuV[IDX2(a, c, m)] = s;
            }
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += fxi[IDX2(l, r, n)] * V[IDX2(l, c, n)];
                fxV[IDX2(r, c, n)] = s;
            }
        }
        for (int j = 0; j < n; ++j)                                          /* Qux :288 */
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += fuV[IDX2(a, l, m)] * fxi[IDX2(l, j, n)];
                Qux[IDX2(a, j, m)] = cxu[nm * i + IDX2(j, a, n)] + s;
            }
        for (int b = 0; b < m; ++b)                                          /* Quu :289 */
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += fuV[IDX2(a, l, m)] * fui[IDX2(l, b, n)];
                Q[IDX2(a, b, m)] = cuu[mm * i + IDX2(a, b, m)] + s;
            }
        for (int c = 0; c < n; ++c)                                          /* Qxx :290 */
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += fxV[IDX2(r, l, n)] * fxi[IDX2(l, c, n)];
                Qxx[IDX2(r, c, n)] = cxx[nn * i + IDX2(r, c, n)] + s;
            }
        /* :294-299  Q• <- Q•/η + c•kl */
        for (int a = 0; a < m; ++a) Qu[a] = Qu[a] / et + cukl[IDX2(a, i, m)];
        for (size_t e = 0; e < nm; ++e) Qux[e] = Qux[e] / et + cxukl[nm * i + e];
        for (size_t e = 0; e < mm; ++e) Q[e] = Q[e] / et + cuukl[mm * i + e];
        for (int j = 0; j < n; ++j) Qx[j] = Qx[j] / et + cxkl[IDX2(j, i, n)];
        for (size_t e = 0; e < nn; ++e) Qxx[e] = Qxx[e] / et + cxxkl[nn * i + e];
        for (int b = 0; b < m; ++b)                                          /* :301 */
            for (int a = b + 1; a < m; ++a) {
                const double s = 0.5 * (Q[IDX2(a, b, m)] + Q[IDX2(b, a, m)]);
                Q[IDX2(a, b, m)] = s; Q[IDX2(b, a, m)] = s;
            }
        for (size_t t = 0; t < nm; ++t) Ki[t] = 0.0;
        if (no_lims) {                                                       /* :303-316 */
            if (chol_upper_(m, Q, Rf) != 0) { diverge = i + 1; goto out; }
            for (int a = 0; a < m; ++a) col[a] = Qu[a];
            chol_solve_ld(m, Rf, m, col);
            for (int a = 0; a < m; ++a) ki[a] = -col[a];
            for (int j = 0; j < n; ++j) {
                for (int a = 0; a < m; ++a) col[a] = Qux[IDX2(a, j, m)];
                chol_solve_ld(m, Rf, m, col);
                for (int a = 0; a < m; ++a) Ki[IDX2(a, j, m)] = -col[a];
            }
        } else {                                                             /* :317-335 */
            const int ws = (i + 1 < N - 2) ? i + 1 : N - 2;
            int nfree = 0, result;
            for (int a = 0; a < m; ++a) {
                lower[a] = lims[IDX2(a, 0, m)] - u[IDX2(a, i, m)];
                upper[a] = lims[IDX2(a, 1, m)] - u[IDX2(a, i, m)];
                x0[a] = k[IDX2(a, ws, m)];
            }
            result = ddp_oracle_boxqp(m, Q, Qu, lower, upper, x0, NULL, ki, Hfree, freev, &nfree, NULL);
            if (result < 1) { diverge = i + 1; goto out; }
            if (nfree > 0) {
                int idx[64], nf = 0;
                for (int a = 0; a < m && nf < 64; ++a) if (freev[a]) idx[nf++] = a;
                for (int j = 0; j < n; ++j) {
                    for (int a = 0; a < nf; ++a) col[a] = Qux[IDX2(idx[a], j, m)];
                    chol_solve_ld(nf, Hfree, m, col);
                    for (int a = 0; a < nf; ++a) Ki[IDX2(idx[a], j, m)] = -col[a];
                }
            }
        }
        /* :338-341 value update with the KL-augmented Quu, Qux */
        double kQuuk = 0.0, kQu = 0.0;
        for (int a = 0; a < m; ++a) {
            double s = 0.0;
            for (int b = 0; b < m; ++b) s += Q[IDX2(a, b, m)] * ki[b];
            Quuk[a] = s;
        }
        for (int a = 0; a < m; ++a) { kQuuk += ki[a] * Quuk[a]; kQu += ki[a] * Qu[a]; }
        dV[0] += kQu; dV[1] += 0.5 * kQuuk;
        double *Vxi = Vx + (size_t)n * i, *Vxxi = Vxx + nn * i;
        for (int j = 0; j < n; ++j) {
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int a = 0; a < m; ++a) {
                s1 += Ki[IDX2(a, j, m)] * Quuk[a];
                s2 += Ki[IDX2(a, j, m)] * Qu[a];
                s3 += Qux[IDX2(a, j, m)] * ki[a];
            }
            Vxi[j] = ((Qx[j] + s1) + s2) + s3;
        }
        for (int c = 0; c < n; ++c)
            for (int r = 0; r < n; ++r) {
                double s1 = 0.0, s2 = 0.0, s3 = 0.0;
                for (int b = 0; b < m; ++b) {
                    double kq = 0.0;
                    for (int a = 0; a < m; ++a) kq += Ki[IDX2(a, r, m)] * Q[IDX2(a, b, m)];
                    s1 += kq * Ki[IDX2(b, c, m)];
                }
                for (int a = 0; a < m; ++a) {
                    s2 += Ki[IDX2(a, r, m)] * Qux[IDX2(a, c, m)];
                    s3 += Qux[IDX2(a, r, m)] * Ki[IDX2(a, c, m)];
                }
                M[IDX2(r, c, n)] = ((Qxx[IDX2(r, c, n)] + s1) + s2) + s3;
            }
        for (int c = 0; c < n; ++c)
            for (int r = 0; r < n; ++r) Vxxi[IDX2(r, c, n)] = .5 * (M[IDX2(r, c, n)] + M[IDX2(c, r, n)]);
        for (int a = 0; a < m; ++a) k[IDX2(a, i, m)] = ki[a];                 /* :344-346 */
        for (size_t t = 0; t < nm; ++t) K[nm * i + t] = Ki[t];
        inv_(m, Q, Quui + mm * i);
    }
out:
    free(w); free(freev);
    return diverge;
}

/* ===================================================================================
 * forward_covariance — src/forward_pass.jl:37-56.  sigmanew[(n+m),(n+m),N]; entries the reference leaves
 * `undef` (the u-blocks of the last time step) are zero here.  R1 is both the process noise and Σ0 (:42-43).
 * =================================================================================== */
void ddp_oracle_forward_covariance(int n, int m, int N, const double *fx, const double *R1,
                                   const double *K, const double *Sigma, double *sigmanew)
{
    const int p = n + m;
    const size_t pp = (size_t)p * p, nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    double *T1 = (double *)malloc(sizeof(double) * nn), *KS = (double *)malloc(sizeof(double) * nm);
    memset(sigmanew, 0, sizeof(double) * pp * N);
    for (int c = 0; c < n; ++c) for (int r = 0; r < n; ++r) sigmanew[IDX2(r, c, p)] = R1[IDX2(r, c, n)];
    for (int i = 0; i < N - 1; ++i) {
        double *S = sigmanew + pp * i, *Sn = S + pp;
        const double *F = fx + nn * i, *Ki = K + nm * i, *Sg = Sigma + mm * i;
        for (int c = 0; c < n; ++c)                                        /* (fx*Sxx) */
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += F[IDX2(r, l, n)] * S[IDX2(l, c, p)];
                T1[IDX2(r, c, n)] = s;
            }
        for (int c = 0; c < n; ++c)                                        /* *fx' + R1  (:49) */
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += T1[IDX2(r, l, n)] * F[IDX2(c, l, n)];
                Sn[IDX2(r, c, p)] = s + R1[IDX2(r, c, n)];
            }
        for (int c = 0; c < n; ++c)                                        /* K*Sxx (:50) */
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += Ki[IDX2(a, l, m)] * S[IDX2(l, c, p)];
                KS[IDX2(a, c, m)] = s;
                S[IDX2(n + a, c, p)] = s;
            }
        for (int a = 0; a < m; ++a)                                        /* Sxx*K' (:51) */
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += S[IDX2(r, l, p)] * Ki[IDX2(a, l, m)];
                S[IDX2(r, n + a, p)] = s;
            }
        for (int b = 0; b < m; ++b)                                        /* K*Sxx*K' + Σ (:52) */
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += KS[IDX2(a, l, m)] * Ki[IDX2(b, l, m)];
                S[IDX2(n + a, n + b, p)] = s + Sg[IDX2(a, b, m)];
            }
    }
    free(T1); free(KS);
}

/* this build's `covariance(model,x,u)`: empirical covariance (1/(T-1) normalisation, mean removed — Julia's
 * `cov`) of the residuals x[:,t+1] - fx_t x[:,t] - fu_t u[:,t], t = 1..N-1 */
void ddp_oracle_model_covariance(int n, int m, int N, const double *fx, const double *fu, const double *x,
                                 const double *u, double *R1)
{
    const size_t nn = (size_t)n * n, nm = (size_t)n * m;
    const int T = N - 1;
    double *E = (double *)calloc((size_t)n * (size_t)(T > 0 ? T : 1), sizeof(double)), *mu = (double *)calloc((size_t)n, sizeof(double));
    for (int t = 0; t < T; ++t)
        for (int r = 0; r < n; ++r) {
            double s = x[IDX2(r, t + 1, n)];
            for (int l = 0; l < n; ++l) s -= fx[nn * t + IDX2(r, l, n)] * x[IDX2(l, t, n)];
            for (int a = 0; a < m; ++a) s -= fu[nm * t + IDX2(r, a, n)] * u[IDX2(a, t, m)];
            E[IDX2(r, t, n)] = s; mu[r] += s;
        }
    for (int r = 0; r < n; ++r) mu[r] /= (T > 0 ? T : 1);
    for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) {
            double s = 0.0;
            for (int t = 0; t < T; ++t) s += (E[IDX2(r, t, n)] - mu[r]) * (E[IDX2(c, t, n)] - mu[c]);
            R1[IDX2(r, c, n)] = (T > 1) ? s / (T - 1) : 0.0;
        }
    free(E); free(mu);
}

/* ===================================================================================
 * kl_div_wiki — src/klutils.jl:70-103.  kldiv[T] (clipped at 0, :101); returns 0, or 1 if a logdet threw
 * (the reference then returns the scalar Inf, :95-99).
 * =================================================================================== */
int ddp_oracle_kl_div_wiki(int n, int m, int T, const double *xnew, const double *xold, const double *sigmanew,
                           const double *Kn, const double *kn, const double *Sn,
                           const double *Kp, const double *kp, const double *Sp, const double *Sip, double *kldiv)
{
    const int p = n + m;
    const size_t pp = (size_t)p * p, nm = (size_t)n * m, mm = (size_t)m * m;
    double *kd = (double *)malloc(sizeof(double) * (size_t)m), *Kd = (double *)malloc(sizeof(double) * nm),
           *SK = (double *)malloc(sizeof(double) * nm), *mu = (double *)malloc(sizeof(double) * (size_t)n),
           *Kmu = (double *)malloc(sizeof(double) * (size_t)m);
    int threw = 0;
    for (int t = 0; t < T; ++t) {
        const double *St = sigmanew + pp * t, *Sipt = Sip + mm * t, *Snt = Sn + mm * t, *Spt = Sp + mm * t;
        for (int j = 0; j < n; ++j) mu[j] = xnew[IDX2(j, t, n)] - xold[IDX2(j, t, n)];
        for (int a = 0; a < m; ++a) kd[a] = kp[IDX2(a, t, m)] - kn[IDX2(a, t, m)];
        for (size_t e = 0; e < nm; ++e) Kd[e] = Kp[nm * t + e] - Kn[nm * t + e];
        double tr1 = 0.0, q1 = 0.0;
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) {
                tr1 += Sipt[IDX2(a, b, m)] * Snt[IDX2(b, a, m)];             /* tr(Σip*Σn) */
                q1 += kd[a] * Sipt[IDX2(a, b, m)] * kd[b];                   /* k_diff'Σip k_diff */
            }
        const double ldp = logdet_(m, Spt), ldn = logdet_(m, Snt);
        if (isnan(ldp) || isnan(ldn)) { threw = 1; break; }
        double v = 0.5 * (tr1 + q1 - m + ldp - ldn);                          /* :92 */
        for (int j = 0; j < n; ++j)                                           /* Σip*K_diff */
            for (int a = 0; a < m; ++a) {
                double s = 0.0;
                for (int b = 0; b < m; ++b) s += Sipt[IDX2(a, b, m)] * Kd[IDX2(b, j, m)];
                SK[IDX2(a, j, m)] = s;
            }
        for (int a = 0; a < m; ++a) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += Kd[IDX2(a, j, m)] * mu[j];
            Kmu[a] = s;
        }
        double q2 = 0.0, tr2 = 0.0, q3 = 0.0;
        for (int a = 0; a < m; ++a) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += SK[IDX2(a, j, m)] * mu[j];
            q2 += Kmu[a] * s;                                                  /* μ'K_diff'Σip K_diff μ */
            q3 += kd[a] * s;                                                   /* k_diff'Σip K_diff μ   (:94) */
        }
        for (int r = 0; r < n; ++r)                                            /* tr(K_diff'Σip K_diff Σt), Σt = Σ_new[1:n,1:n,t] */
            for (int c = 0; c < n; ++c) {
                double s = 0.0;
                for (int a = 0; a < m; ++a) s += Kd[IDX2(a, r, m)] * SK[IDX2(a, c, m)];
                tr2 += s * St[IDX2(c, r, p)];
            }
        v += 0.5 * (q2 + tr2);                                                 /* :93 */
        v += q3;
        kldiv[t] = v > 0.0 ? v : 0.0;                                          /* :101 */
    }
    free(kd); free(Kd); free(SK); free(mu); free(Kmu);
    return threw;
}

/* calc_η, scalar kl_step — src/klutils.jl:112-133.  etab[3] is mutated; returns `satisfied`; *divergence out. */
int ddp_oracle_calc_eta(double *etab, double divergence_mean, double kl_step)
{
    if (!(kl_step > 0)) return 1;
    const double viol = divergence_mean - kl_step;
    const int satisfied = fabs(viol) < 0.1 * kl_step;
    if (!satisfied) {
        if (viol < 0) {                                                        /* η was too big */
            etab[2] = etab[1];
            const double g = sqrt(etab[0] * etab[2]);
            etab[1] = g > 0.1 * etab[2] ? g : 0.1 * etab[2];
        } else {                                                               /* η was too small */
            etab[0] = etab[1];
            const double g = sqrt(etab[0] * etab[2]);
            etab[1] = g < 10.0 * etab[0] ? g : 10.0 * etab[0];
        }
    }
    return satisfied;
}

/* ===================================================================================
 * iLQGkl, single KL constraint — src/iLQGkl.jl:25-178,234-252, for a registered problem family.
 * x0[n,N] pre-rolled trajectory with cost0 = sum(cost); traj_prev = (Kp,kp,Sp,Sip) [kp is the previous control
 * sequence u]; model = (mfx[n,n,N], R1[n,n]).  Outputs x,u,K,k(:=u, :239),Quu(Σi field),Quui(Σ field),Vx,Vxx,cost.
 * =================================================================================== */
int ddp_oracle_ilqgkl(const ddp_oracle_problem *p, const double *x0, double cost0,
                      const double *Kp, const double *kp_in, const double *Sp, const double *Sip,
                      const double *mfx, const double *R1, const double *lims,
                      double kl_step, int max_iter, const double *etab_in, double del0,
                      double *x, double *u, double *K, double *k, double *Quu, double *Quui,
                      double *Vx, double *Vxx, double *cost, ddp_oracle_ilqgkl_result *res)
{
    const int n = p->n, m = p->m, N = p->N, pdim = n + m;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m, pp = (size_t)pdim * pdim;
    const int clen = ddp_oracle_cost_len(p);
    double etab[3] = {etab_in[0], etab_in[1], etab_in[2]};
    double *uu = (double *)malloc(sizeof(double) * (size_t)m * N);             /* u = copy(traj_prev.k) :45 */
    double *kzero = (double *)calloc((size_t)m * N, sizeof(double));           /* traj_prev.k *= 0      :51 */
    double *xx = (double *)malloc(sizeof(double) * (size_t)n * N);
    double *cxk = (double *)malloc(sizeof(double) * (size_t)n * N), *cuk = (double *)malloc(sizeof(double) * (size_t)m * N),
           *cxxk = (double *)malloc(sizeof(double) * nn * N), *cxuk = (double *)malloc(sizeof(double) * nm * N),
           *cuuk = (double *)malloc(sizeof(double) * mm * N);
    double *cx = (double *)malloc(sizeof(double) * (size_t)n * N), *cu = (double *)malloc(sizeof(double) * (size_t)m * N);
    double *fx = (double *)malloc(sizeof(double) * nn * N), *fu = (double *)malloc(sizeof(double) * nm * N);
    double *cxx = (double *)malloc(sizeof(double) * nn * N), *cxu = (double *)calloc(nm * N, sizeof(double)),
           *cuu = (double *)malloc(sizeof(double) * mm * N);
    double *xnew = (double *)malloc(sizeof(double) * (size_t)n * N), *unew = (double *)malloc(sizeof(double) * (size_t)m * N),
           *cnew = (double *)malloc(sizeof(double) * (size_t)clen), *sig = (double *)malloc(sizeof(double) * pp * N),
           *kld = (double *)malloc(sizeof(double) * (size_t)N);
    double dV[2] = {0, 0}, divergence = 0.0, g_norm = 0.0;
    int satisfied = 0, iter = 0, nback = 0, status = 0;
    memcpy(uu, kp_in, sizeof(double) * (size_t)m * N);
    memcpy(xx, x0, sizeof(double) * (size_t)n * N);
    /* STEP 1 :86 — derivs(x,u); the KL demos hand out 3-D arrays (demo_linear.jl:91-101) */
    ddp_oracle_df(p, xx, uu, cx, cu, fx, fu);
    for (int t = 0; t < N; ++t) {
        if (p->kind == DDP_ORACLE_LQ) {
            memcpy(fx + nn * t, p->A + (p->dyn_tv ? nn * t : 0), sizeof(double) * nn);
            memcpy(fu + nm * t, p->Bm + (p->dyn_tv ? nm * t : 0), sizeof(double) * nm);
        }
        memcpy(cxx + nn * t, p->Q, sizeof(double) * nn);
        memcpy(cuu + mm * t, p->R, sizeof(double) * mm);
    }
    ddp_oracle_kl_terms(n, m, N, Kp, kzero, Sip, cxk, cuk, cxxk, cxuk, cuuk);   /* :90 (traj_prev.k is zero here) */
    for (iter = 1; iter <= max_iter; ++iter) {                                  /* :91 */
        int diverge = 1;
        while (diverge > 0) {                                                   /* :95-122 */
            diverge = ddp_oracle_back_pass_gps(n, m, N, cx, cu, cxx, cxu, cuu, fx, fu, lims, uu, cxk, cuk, cxxk, cxuk, cuuk,
                                               &etab[1], 0, K, k, Quu, Quui, Vx, Vxx, dV);
            ++nback;
            if (diverge > 0) { etab[1] += del0; del0 *= 2; }
            if (nback > 10000) { status = -2; goto done; }                      /* guard: the reference loops forever */
        }
        {   /* g_norm :125 */
            double s = 0.0;
            for (int t = 0; t < N; ++t) {
                double mx = 0.0;
                for (int a = 0; a < m; ++a) { const double r = fabs(k[IDX2(a, t, m)]) / (fabs(uu[IDX2(a, t, m)]) + 1); if (r > mx) mx = r; }
                s += mx;
            }
            g_norm = s / N;
        }
        ddp_oracle_forward_pass(p, K, k, x0, uu, xx, 1.0, lims, xnew, unew, cnew);   /* :132 */
        ddp_oracle_forward_covariance(n, m, N, mfx, R1, K, Quui, sig);          /* :133 (traj.Σ = Quui field) */
        /* :134 traj_new.k .+= traj_prev.k (zero) */
        if (ddp_oracle_kl_div_wiki(n, m, N, xnew, xx, sig, K, k, Quui, Kp, kzero, Sp, Sip, kld)) divergence = INFINITY;
        else { double s = 0.0; for (int t = 0; t < N; ++t) s += kld[t]; divergence = s / N; }
        satisfied = ddp_oracle_calc_eta(etab, divergence, kl_step);             /* :141 */
        if (satisfied) { status = 1; break; }                                   /* :169-173 */
        if (etab[1] > 0.999 * etab[2]) { status = 2; break; }                   /* :174-177 */
    }
    if (iter > max_iter) { iter = max_iter; status = 3; }                       /* :234 */
done:
    memcpy(x, xnew, sizeof(double) * (size_t)n * N);                            /* :236-237 */
    memcpy(u, unew, sizeof(double) * (size_t)m * N);
    memcpy(cost, cnew, sizeof(double) * (size_t)clen);
    memcpy(k, unew, sizeof(double) * (size_t)m * N);                            /* traj_new.k = copy(u) */
    if (res) {
        res->status = status; res->iter = iter; res->n_backpass = nback; res->eta[0] = etab[0]; res->eta[1] = etab[1];
        res->eta[2] = etab[2]; res->divergence = divergence; res->g_norm = g_norm; res->dV[0] = dV[0]; res->dV[1] = dV[1];
        res->satisfied = satisfied; res->cost0 = cost0;
    }
    free(uu); free(kzero); free(xx); free(cxk); free(cuk); free(cxxk); free(cxuk); free(cuuk); free(cx); free(cu); free(fx);
    free(fu); free(cxx); free(cxu); free(cuu); free(xnew); free(unew); free(cnew); free(sig); free(kld);
    return status;
}
