// boxqp_dev.h — device-side box-constrained QP (projected Newton), src/boxQP.jl:29-188.
//
// Every lane of the calling wavefront runs the SAME solve on the SAME data (operands come from
// LDS broadcasts), so all branches are wave-uniform and no cross-lane traffic is needed; for the
// m <= 8 of the hot path a lane-parallel factorisation would cost more in shuffles than it saves.
//
// H[free,free] is never gathered: the Cholesky runs on the full m x m matrix with clamped
// rows/columns replaced by identity ("masked" factorisation).  For the free indices this performs
// exactly the arithmetic of cholesky(H[free,free]) (the extra terms are exact zeros), keeps every
// loop bound static, and therefore keeps H, R and the vectors in registers.
#pragma once
#include <hip/hip_runtime.h>
#include "ddp_internal.h"   // ddp_rsqrt

struct QPOptsDev {
    int    maxIter;
    double minGrad, minRelImprove, stepDec, minStep, Armijo;
};

__device__ __forceinline__ double ddp_clamp(double x, double lo, double hi)
{   // Base.clamp: NaN passes through
    return x > hi ? hi : (x < lo ? lo : x);
}

// Upper Cholesky of H with rows/cols in `clamped` replaced by identity; reads the upper triangle
// (LAPACK potrf 'U' / cholesky(Hermitian(.))).  Returns 0 ok, j+1 on a non-positive pivot.
template <int MM>
__device__ __forceinline__ int chol_masked(int m, const double (&H)[MM * MM], unsigned clamped,
                                           double (&R)[MM * MM])
{
    int fail = 0;
#pragma unroll
    for (int j = 0; j < MM; ++j) {
        if (j < m) {
            const bool cj = (clamped >> j) & 1u;
            double ajj = cj ? 1.0 : H[j + MM * j];
#pragma unroll
            for (int k = 0; k < j; ++k) ajj -= R[k + MM * j] * R[k + MM * j];
            if (!(ajj > 0.0) && fail == 0) fail = j + 1;
            ajj = sqrt(ajj);
            R[j + MM * j] = ajj;
#pragma unroll
            for (int i = j + 1; i < MM; ++i) {
                if (i < m) {
                    const bool ci = (clamped >> i) & 1u;
                    double s = (cj || ci) ? 0.0 : H[j + MM * i];
#pragma unroll
                    for (int k = 0; k < j; ++k) s -= R[k + MM * j] * R[k + MM * i];
                    R[j + MM * i] = s / ajj;
                }
            }
        }
    }
    return fail;
}

// The same factorisation with RECIPROCAL pivots ri[j] = 1/R[j][j] (v_rsq_f64 + Newton steps): no square root and no division.
// Every lane of a wave repeats these solves, and an IEEE fp64 division is ~20 instructions: with limits at m = 8 the
// divisions of the generic routines were most of a backward step.  R keeps the true factor (diagonal = ajj·ri).
template <int MM>
__device__ __forceinline__ int chol_masked_ri(int m, const double (&H)[MM * MM], unsigned clamped,
                                              double (&R)[MM * MM], double (&ri)[MM])
{
    int fail = 0;
#pragma unroll
    for (int j = 0; j < MM; ++j) {
        if (j < m) {
            const bool cj = (clamped >> j) & 1u;
            double ajj = cj ? 1.0 : H[j + MM * j];
#pragma unroll
            for (int k = 0; k < j; ++k) ajj -= R[k + MM * j] * R[k + MM * j];
            if (!(ajj > 0.0) && fail == 0) fail = j + 1;
            const double r = ddp_rsqrt(ajj);
            ri[j] = r;
            R[j + MM * j] = ajj * r;
#pragma unroll
            for (int i = j + 1; i < MM; ++i) {
                if (i < m) {
                    const bool ci = (clamped >> i) & 1u;
                    double s = (cj || ci) ? 0.0 : H[j + MM * i];
#pragma unroll
                    for (int k = 0; k < j; ++k) s -= R[k + MM * j] * R[k + MM * i];
                    R[j + MM * i] = s * r;
                }
            }
        } else {
            ri[j] = 0.0;
        }
    }
    return fail;
}

template <int MM>
__device__ __forceinline__ void chol_solve_ri(int m, const double (&R)[MM * MM], const double (&ri)[MM], double (&b)[MM])
{
#pragma unroll
    for (int i = 0; i < MM; ++i) {
        if (i < m) {
            double s = b[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s -= R[k + MM * i] * b[k];
            b[i] = s * ri[i];
        }
    }
#pragma unroll
    for (int i = MM - 1; i >= 0; --i) {
        if (i < m) {
            double s = b[i];
#pragma unroll
            for (int k = i + 1; k < MM; ++k)
                if (k < m) s -= R[i + MM * k] * b[k];
            b[i] = s * ri[i];
        }
    }
}

// solve (R'R) b = b in place (potrs)
template <int MM>
__device__ __forceinline__ void chol_solve(int m, const double (&R)[MM * MM], double (&b)[MM])
{
#pragma unroll
    for (int i = 0; i < MM; ++i) {
        if (i < m) {
            double s = b[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s -= R[k + MM * i] * b[k];
            b[i] = s / R[i + MM * i];
        }
    }
#pragma unroll
    for (int i = MM - 1; i >= 0; --i) {
        if (i < m) {
            double s = b[i];
#pragma unroll
            for (int k = i + 1; k < MM; ++k)
                if (k < m) s -= R[i + MM * k] * b[k];
            b[i] = s / R[i + MM * i];
        }
    }
}

template <int MM>
__device__ __forceinline__ double qp_value(int m, const double (&H)[MM * MM], const double (&g)[MM],
                                           const double (&x)[MM])
{   // (x'g + 0.5x'H*x)[1]  — boxQP.jl:63,141,146
    double xg = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < MM; ++i)
        if (i < m) xg += x[i] * g[i];
#pragma unroll
    for (int j = 0; j < MM; ++j) {
        if (j < m) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < MM; ++i)
                if (i < m) t += (0.5 * x[i]) * H[i + MM * j];
            q += t * x[j];
        }
    }
    return xg + q;
}

// Returns `result` (boxQP.jl:172-179; 0 also for the swallowed PosDefException,
// backward_pass.jl:48-52).  On return x is the solution, `clamped` the bit mask of clamped
// coordinates belonging to the returned factor R (quirk Q12: on result 4 both are from the
// previous iteration), `iters` the final value of `iter`.
// `ri`: reciprocal pivots of the returned factor (chol_solve_ri); zeros if nothing was factorised.
template <int MM>
__device__ __forceinline__ int boxqp_dev_ri(int m, const double (&H)[MM * MM], const double (&g)[MM],
                                            const double (&lower)[MM], const double (&upper)[MM],
                                            const double (&x0)[MM], const QPOptsDev &o,
                                            double (&x)[MM], double (&R)[MM * MM], double (&ri)[MM], unsigned &clamped,
                                            int &iters)
{
    const unsigned all = (m >= 32) ? 0xffffffffu : ((1u << m) - 1u);
    double grad[MM], search[MM], xc[MM];
    int    result = 0, iter = 1;
    double oldvalue = 0.0, value;
    clamped = 0u;
#pragma unroll
    for (int i = 0; i < MM * MM; ++i) R[i] = 0.0;
#pragma unroll
    for (int i = 0; i < MM; ++i) ri[i] = 0.0;
#pragma unroll
    for (int i = 0; i < MM; ++i) x[i] = (i < m) ? ddp_clamp(x0[i], lower[i], upper[i]) : 0.0;   // :58
    value = qp_value<MM>(m, H, g, x);                                                          // :63

    while (iter <= o.maxIter) {                                                                // :71
        if (result != 0) break;
        if (iter > 1 && (oldvalue - value) < o.minRelImprove * fabs(oldvalue)) { result = 4; break; }
        oldvalue = value;
        unsigned newc = 0u;
#pragma unroll
        for (int i = 0; i < MM; ++i) {                                                         // :85-95
            if (i < m) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < MM; ++j)
                    if (j < m) s += H[i + MM * j] * x[j];
                grad[i] = g[i] + s;
                const bool c = ((x[i] == lower[i]) && (grad[i] > 0)) || ((x[i] == upper[i]) && (grad[i] < 0));
                newc |= (c ? 1u : 0u) << i;
            } else {
                grad[i] = 0.0;
            }
        }
        const unsigned oldc = clamped;
        clamped = newc;
        if (clamped == all) { result = 6; break; }                                             // :98-101
        if (iter == 1 || oldc != clamped) {                                                    // :104-117
            if (chol_masked_ri<MM>(m, H, clamped, R, ri) != 0) { result = 0; break; }          // throw -> 0
        }
        double gn = 0.0;                                                                       // :120-124
#pragma unroll
        for (int i = 0; i < MM; ++i)
            if (i < m && !((clamped >> i) & 1u)) gn += grad[i] * grad[i];
        if (gn < o.minGrad * o.minGrad) { result = 5; break; }                                 // norm(grad[free]) < minGrad
#pragma unroll
        for (int i = 0; i < MM; ++i) {                                                         // :127-129
            if (i < m) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < MM; ++j)
                    if (j < m && ((clamped >> j) & 1u)) s += H[i + MM * j] * x[j];
                search[i] = ((clamped >> i) & 1u) ? 0.0 : (g[i] + s);
            } else {
                search[i] = 0.0;
            }
        }
        chol_solve_ri<MM>(m, R, ri, search);
        double sdotg = 0.0;
#pragma unroll
        for (int i = 0; i < MM; ++i) {
            if (i < m) {
                search[i] = ((clamped >> i) & 1u) ? 0.0 : (-search[i] - x[i]);
                sdotg += search[i] * grad[i];                                                  // :132
            }
        }
        if (sdotg >= 0) break;                                                                 // :133-135
        double step = 1.0, vc;                                                                 // :138-151
#pragma unroll
        for (int i = 0; i < MM; ++i) xc[i] = (i < m) ? ddp_clamp(x[i] + step * search[i], lower[i], upper[i]) : 0.0;
        vc = qp_value<MM>(m, H, g, xc);
        while ((vc - oldvalue) > o.Armijo * (step * sdotg)) {                                  // ratio < Armijo, step·sdotg < 0
            step = step * o.stepDec;
#pragma unroll
            for (int i = 0; i < MM; ++i) xc[i] = (i < m) ? ddp_clamp(x[i] + step * search[i], lower[i], upper[i]) : 0.0;
            vc = qp_value<MM>(m, H, g, xc);
            if (step < o.minStep) { result = 2; break; }
        }
#pragma unroll
        for (int i = 0; i < MM; ++i) x[i] = xc[i];                                             // :161-163
        value = vc;
        iter += 1;
    }
    if (iter == o.maxIter) result = 1;                                                         // :167-169
    iters = iter;
    return result;
}

template <int MM>
__device__ __forceinline__ int boxqp_dev(int m, const double (&H)[MM * MM], const double (&g)[MM],
                                         const double (&lower)[MM], const double (&upper)[MM],
                                         const double (&x0)[MM], const QPOptsDev &o,
                                         double (&x)[MM], double (&R)[MM * MM], unsigned &clamped,
                                         int &iters)
{
    double ri[MM];
    return boxqp_dev_ri<MM>(m, H, g, lower, upper, x0, o, x, R, ri, clamped, iters);
}

// m = 1 (the reference's own limited case, pendcart): the same control flow as boxqp_dev<1> written on scalars and
// without divisions or square roots —  (g/R)/R with R = sqrt(H) becomes g·(1/H) (v_rcp_f64 + 2 Newton steps), the
// gradient norm sqrt(grad²) is |grad|, and the Armijo test (vc - old)/(step·sdotg) < Armijo is multiplied through by
// step·sdotg < 0.  Differences to the generic routine are of rounding order (decisions can differ only when a tested
// quantity sits within an ulp of its threshold).  `rH` = 1/H[free] of the returned factor replaces R (0 if none).
__device__ __forceinline__ double ddp_rcp_nr(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

__device__ __forceinline__ int boxqp_dev1(double H, double g, double lower, double upper, double x0, const QPOptsDev &o,
                                          double &x, double &rH, unsigned &clamped, int &iters)
{
    int result = 0, iter = 1;
    double oldvalue = 0.0, value;
    clamped = 0u;
    rH = 0.0;
    x = ddp_clamp(x0, lower, upper);                                                           // :58
    auto val = [&](double xx) { return xx * g + ((0.5 * xx) * H) * xx; };                      // :63 (same association as qp_value)
    value = val(x);
    while (iter <= o.maxIter) {                                                                // :71
        if (result != 0) break;
        if (iter > 1 && (oldvalue - value) < o.minRelImprove * fabs(oldvalue)) { result = 4; break; }
        oldvalue = value;
        const double grad = g + H * x;                                                         // :85
        const unsigned newc = (((x == lower) && (grad > 0)) || ((x == upper) && (grad < 0))) ? 1u : 0u;
        const unsigned oldc = clamped;
        clamped = newc;
        if (clamped == 1u) { result = 6; break; }                                              // :98-101
        if (iter == 1 || oldc != clamped) {                                                    // :104-117
            if (!(H > 0.0)) { result = 0; break; }                                             // PosDefException -> 0
            rH = ddp_rcp_nr(H);
        }
        if (fabs(grad) < o.minGrad) { result = 5; break; }                                     // :120-124
        const double search = -(g * rH) - x;                                                   // :127-129 (nothing clamped here)
        const double sdotg = search * grad;                                                    // :132
        if (sdotg >= 0) break;                                                                 // :133-135
        double step = 1.0;                                                                     // :138-151
        double xc = ddp_clamp(x + step * search, lower, upper), vc = val(xc);
        while ((vc - oldvalue) > o.Armijo * (step * sdotg)) {                                  // ratio < Armijo with step·sdotg < 0
            step = step * o.stepDec;
            xc = ddp_clamp(x + step * search, lower, upper);
            vc = val(xc);
            if (step < o.minStep) { result = 2; break; }
        }
        x = xc;                                                                                // :161-163
        value = vc;
        iter += 1;
    }
    if (iter == o.maxIter) result = 1;                                                         // :167-169
    iters = iter;
    return result;
}
