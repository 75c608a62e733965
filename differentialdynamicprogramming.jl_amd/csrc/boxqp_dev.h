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

// m = 2 in STRAIGHT-LINE code (src/boxQP.jl:58-169 for the reference's usual flow; everything else falls into boxqp_dev_ri<2>).
//
// The generic loop is a serial chain: per projected-Newton iteration ~55 dependent fp64 instructions (gradient, clamp tests, masked
// Cholesky with two reciprocal square roots, two triangular solves, line-search value, Armijo test) and eight wave-uniform branches —
// 1.1 us of a 1.6 us backward step with limits at one wave per trajectory, and in the 16-lane-row kernels the four trajectories of a
// wave serialise their different paths through it.  For m = 2 almost all of that work does not depend on the iterate:
//   * the masked factorisation depends on the CLAMPED SET only: three candidates (nothing clamped, coordinate 0, coordinate 1; both
//     clamped is exit 6 before any factorisation) — computed up front, side by side, from rsqrt(H00), rsqrt(H11), rsqrt(H11 - R01^2);
//   * a clamped coordinate sits exactly ON a bound (boxQP.jl:88-94 tests x == lower / upper), so the right-hand side g_f + H_fc x_c of the
//     Newton system takes one of five values (free set {0,1}; {1} with x0 at lower / upper; {0} with x1 at lower / upper): the five
//     solves are done up front as well.
// What is left per iteration is: gradient (2 deep), clamp tests, a select among the candidates, search = -sol - x, s'g, the projected
// step and its value, the Armijo test — ~19 dependent instructions.  Three iterations are unrolled, each left as soon as the loop would
// leave (the reference's flow: iteration 1 factorises and steps, iteration 2 or 3 finds the same clamped set and leaves through exit
// 4 / 5 / 6; the first form ran all three under a predicate and was SLOWER than the loop on the bench shapes, whose controls sit on their
// bounds at almost every step: exit 6 in the first iteration, tested here before anything else is computed);
// a back-tracking line search (Armijo fails at step 1) or a fourth iteration hands the WHOLE problem to the generic routine, which
// starts from x0 again — so every case it does not finish itself is the generic result by construction.  Same formulas, same order of
// sums as boxqp_dev_ri<2>; results, result codes, clamped sets, factors and iteration counts are those of the loop.
#ifdef DDP_QP2_STATS
__device__ unsigned long long ddp_qp2_stats[12];
#endif
__device__ __forceinline__ int boxqp_dev2(const double (&H)[4], const double (&g)[2], const double (&lower)[2], const double (&upper)[2],
                                          const double (&x0)[2], const QPOptsDev &o, double (&x)[2], double (&R)[4], double (&ri)[2],
                                          unsigned &clamped, int &iters)
{
#ifdef DDP_QP2_OFF       // A/B builds: the generic loop everywhere
    return boxqp_dev_ri<2>(2, H, g, lower, upper, x0, o, x, R, ri, clamped, iters);
#endif
    if (o.maxIter < 5) return boxqp_dev_ri<2>(2, H, g, lower, upper, x0, o, x, R, ri, clamped, iters);      // (the unrolled part assumes iterations 1..4 are allowed)
    const double H0 = H[0], H1 = H[1], H2 = H[2], H3 = H[3], g0 = g[0], g1 = g[1], lo0 = lower[0], lo1 = lower[1], up0 = upper[0], up1 = upper[1];
    double xa = ddp_clamp(x0[0], lo0, up0), xb = ddp_clamp(x0[1], lo1, up1);                             // :58
    // ---- the cheapest way out first: the warm start sits on the bounds and the gradient pushes outward — exit 6 in iteration 1
    // (:98-101), before anything is factorised (with tight limits this is most steps of a backward pass)
    // (no lambdas in this routine: a closure that captures the locals by reference keeps them in memory, and the selects among the
    // candidates became selects among POINTERS into scratch — 9.7 ms instead of 1.6 for the first build)
    double gr0, gr1;
    bool c0, c1;
#define DDP_QP2_GRAD_CLAMP()                                                                                   \
    do {                                                                                                        \
        double s_ = 0.0; s_ += H0 * xa; s_ += H2 * xb; gr0 = g0 + s_;                            /* :85 */     \
        s_ = 0.0; s_ += H1 * xa; s_ += H3 * xb; gr1 = g1 + s_;                                                  \
        c0 = ((xa == lo0) && (gr0 > 0)) || ((xa == up0) && (gr0 < 0));                        /* :88-94 */     \
        c1 = ((xb == lo1) && (gr1 > 0)) || ((xb == up1) && (gr1 < 0));                                          \
    } while (0)
#define DDP_QP2_VAL(res_, a0_, a1_)                                                          /* qp_value<2> */ \
    do {                                                                                                        \
        double xg_ = 0.0; xg_ += (a0_) * g0; xg_ += (a1_) * g1;                                                 \
        double q_ = 0.0, t_ = 0.0;                                                                              \
        t_ += (0.5 * (a0_)) * H0; t_ += (0.5 * (a1_)) * H1; q_ += t_ * (a0_);                                   \
        t_ = 0.0; t_ += (0.5 * (a0_)) * H2; t_ += (0.5 * (a1_)) * H3; q_ += t_ * (a1_);                         \
        res_ = xg_ + q_;                                                                                        \
    } while (0)
#define DDP_QP2_SOLVE(ria_, rib_, r01_, b0i_, b1i_, s0_, s1_)                          /* chol_solve_ri<2> */ \
    do {                                                                                                        \
        double b0_ = (b0i_) * (ria_), t2_ = (b1i_); t2_ -= (r01_) * b0_; double b1_ = t2_ * (rib_);             \
        b1_ = b1_ * (rib_); t2_ = b0_; t2_ -= (r01_) * b1_; b0_ = t2_ * (ria_);                                 \
        s0_ = b0_; s1_ = b1_;                                                                                   \
    } while (0)
    DDP_QP2_GRAD_CLAMP();
    if (c0 && c1) {
#ifdef DDP_QP2_STATS
        if (threadIdx.x == 0) atomicAdd(&ddp_qp2_stats[10], 1ull);
#endif
        x[0] = xa; x[1] = xb; ri[0] = ri[1] = 0.0; R[0] = R[1] = R[2] = R[3] = 0.0;
        clamped = 3u; iters = 1;
        return 6;
    }
    // ---- candidates that do not depend on the iterate (chol_masked_ri<2> for the masks 0, 1, 2; chol_solve_ri<2> on the five right-hand sides)
    const double rone = ddp_rsqrt(1.0);
    const double r0 = ddp_rsqrt(H0), r1b = ddp_rsqrt(H3);
    const double R00 = H0 * r0, R01 = H2 * r0;
    double a11 = H3;
    a11 -= R01 * R01;
    const double r1 = ddp_rsqrt(a11);
    const bool bad00 = !(H0 > 0.0), bad11a = !(a11 > 0.0), bad11b = !(H3 > 0.0);
    // factor of mask c: ri0, ri1, R00, R01, R11 and its failure (scalars, not arrays: a select between two array elements becomes a
    // select between two POINTERS, and the arrays — with the options struct captured by the lambdas — went to scratch memory: 9.7 ms
    // instead of 1.6 for the first build of this routine)
    const double F0a = r0, F0b = r1, F0c = R00, F0d = R01, F0e = a11 * r1;
    const double F1a = rone, F1b = r1b, F1c = 1.0 * rone, F1d = 0.0 * rone, F1e = H3 * r1b;
    const double F2a = r0, F2b = rone, F2c = R00, F2d = 0.0 * r0, F2e = 1.0 * rone;
    const bool fail0 = bad00 || bad11a, fail1 = bad11b, fail2 = bad00;
    double S0a, S0b, S1lo, S1up, S2lo, S2up, dmy;
    DDP_QP2_SOLVE(F0a, F0b, F0d, g0 + 0.0, g1 + 0.0, S0a, S0b);                             // nothing clamped: rhs = g + 0
    DDP_QP2_SOLVE(F1a, F1b, F1d, 0.0, g1 + fma(H1, lo0, 0.0), dmy, S1lo);                   // coordinate 0 clamped at its lower / upper bound
    DDP_QP2_SOLVE(F1a, F1b, F1d, 0.0, g1 + fma(H1, up0, 0.0), dmy, S1up);
    DDP_QP2_SOLVE(F2a, F2b, F2d, g0 + fma(H2, lo1, 0.0), 0.0, S2lo, dmy);                   // coordinate 1 clamped
    DDP_QP2_SOLVE(F2a, F2b, F2d, g0 + fma(H2, up1, 0.0), 0.0, S2up, dmy);
    // ---- the iterations; `done` is the same in every lane that works on this problem, so the early ways out are cheap
    double value, oldvalue = 0.0;                                                                     // :63
    DDP_QP2_VAL(value, xa, xb);
    int result = 0, iter = 1;
    unsigned cl = 0u;
    double Fca = 0.0, Fcb = 0.0, Fcc = 0.0, Fcd = 0.0, Fce = 0.0;      // the factor in use (zeros: nothing factorised yet, as the loop returns it)
    bool slow = false;
    const double mg2 = o.minGrad * o.minGrad, mri = o.minRelImprove, arm = o.Armijo;
#pragma unroll
    for (int it = 1; it <= 4; ++it) {
        if (it > 1) {
            if ((oldvalue - value) < mri * fabs(oldvalue)) { result = 4; break; }                     // :76-79
            if (it == 4) { slow = true; break; }                                                      // a fourth iteration: the generic loop
            DDP_QP2_GRAD_CLAMP();
        }
        oldvalue = value;
        const unsigned newc = (c0 ? 1u : 0u) | (c1 ? 2u : 0u), oldc = cl;
        cl = newc;
        if (newc == 3u) { result = 6; break; }                                                        // :98-101
        if (it == 1 || oldc != newc) {                                                                // :104-117
            Fca = newc == 0u ? F0a : (newc == 1u ? F1a : F2a); Fcb = newc == 0u ? F0b : (newc == 1u ? F1b : F2b);
            Fcc = newc == 0u ? F0c : (newc == 1u ? F1c : F2c); Fcd = newc == 0u ? F0d : (newc == 1u ? F1d : F2d);
            Fce = newc == 0u ? F0e : (newc == 1u ? F1e : F2e);
            if (newc == 0u ? fail0 : (newc == 1u ? fail1 : fail2)) break;                              // PosDefException -> result 0
        }
        double gn = 0.0; gn += c0 ? 0.0 : gr0 * gr0; gn += c1 ? 0.0 : gr1 * gr1;                       // :120-124
        if (gn < mg2) { result = 5; break; }
        // :127-129 with the solves done up front: the Newton point of the face
        const double sol0 = newc == 0u ? S0a : (xb == lo1 ? S2lo : S2up);
        const double sol1 = newc == 0u ? S0b : (xa == lo0 ? S1lo : S1up);
        const double se0 = c0 ? 0.0 : (-sol0 - xa), se1 = c1 ? 0.0 : (-sol1 - xb);
        double sdotg = 0.0; sdotg += se0 * gr0; sdotg += se1 * gr1;                                   // :132
        if (sdotg >= 0) break;                                                                        // :133-135 (result stays 0)
        const double xca = ddp_clamp(xa + 1.0 * se0, lo0, up0), xcb = ddp_clamp(xb + 1.0 * se1, lo1, up1);   // :138-151, step = 1
        double vc;
        DDP_QP2_VAL(vc, xca, xcb);
        if ((vc - oldvalue) > arm * (1.0 * sdotg)) {
            // The line search backs off (:142-151).  Usual cause: the Newton point lies beyond the bounds, the projected step is shorter than
            // the model promises.  While every moving coordinate of the ray x + step·search stays outside its bound the projected point — and
            // its value — do not change with the step, so the loop accepts exactly this point as soon as step <= s* = (old - vc) / (Armijo
            // |s'g|), after however many step sizes (tens, when the iterate starts near the bound: with tight limits that was the time of a
            // limited backward step; the m = 1 form in back_pass_q4.hip: 18 % of the steps of BASELINE config 3).  The accepted step is the
            // largest 0.6^k <= s*, which is > 0.6 s*: each moving coordinate must re-enter the box below that (0.59, no divisions).
            // Everything else — a coordinate that moves inside the box, no improvement at the projected point — is the generic loop's.
            const double dv = oldvalue - vc, ps = -(arm * sdotg);
            const bool pin0 = (se0 == 0.0) || ((xca == (se0 > 0 ? up0 : lo0)) && ((0.59 * dv) * fabs(se0) >= fabs(xca - xa) * ps));
            const bool pin1 = (se1 == 0.0) || ((xcb == (se1 > 0 ? up1 : lo1)) && ((0.59 * dv) * fabs(se1) >= fabs(xcb - xb) * ps));
            if (!(pin0 && pin1 && dv > 1e-21 * ps && o.stepDec == 0.6 && o.minStep <= 1e-22)) { slow = true; break; }
        }
        xa = xca; xb = xcb; value = vc;                                                               // :161-163
        iter += 1;
    }
#undef DDP_QP2_GRAD_CLAMP
#undef DDP_QP2_VAL
#undef DDP_QP2_SOLVE
#ifdef DDP_QP2_STATS     // profiling builds: [calls that fell into the generic loop, calls finished here, their iterations]
    if (threadIdx.x == 0) { atomicAdd(&ddp_qp2_stats[slow ? 0 : 1], 1ull); if (!slow) atomicAdd(&ddp_qp2_stats[2], (unsigned long long)iter); atomicAdd(&ddp_qp2_stats[3 + (result < 0 ? 0 : (result > 6 ? 6 : result))], slow ? 0ull : 1ull); }
#endif
    if (__builtin_expect(slow, 0)) return boxqp_dev_ri<2>(2, H, g, lower, upper, x0, o, x, R, ri, clamped, iters);
    x[0] = xa; x[1] = xb;
    ri[0] = Fca; ri[1] = Fcb; R[0] = Fcc; R[1] = 0.0; R[2] = Fcd; R[3] = Fce;
    clamped = cl;
    iters = iter;
    return result;
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
