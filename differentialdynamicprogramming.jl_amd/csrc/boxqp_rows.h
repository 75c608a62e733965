// boxqp_rows.h — the box-constrained QP of a control-limited backward step (src/boxQP.jl:29-188) with ONE COORDINATE PER LANE.
//
// boxqp_dev.h lets every lane repeat the whole m x m solve on registers (H, the factor and the iterates: ~200 registers for m = 8,
// ~1 500 instructions per projected-Newton iteration) — fine for m <= 3, but in the n = 64 / m = 8 matrix-core kernel that solve was
// half of the step and its register pressure slowed the rest of wave 0.  Here lane i < M of every 16-lane DPP row
// holds coordinate i: x_i, g_i, the bounds, row i and column i of H, column i and row i of the Cholesky factor (no LDS).  Matrix-vector
// products, the factorisation and the triangular solves are runs of  v_fmac_f64_dpp row_newbcast:j  (acc += v[lane j]·own), the
// clamped set is a ballot.  ~350 instructions per iteration with a re-factorisation, 8 + 8 + 8 + 8 matrix registers per lane.
// The four rows of the wave compute identical copies; lanes M..15 of a row carry neutral values.
// Same control flow and result codes as boxQP.jl; sums run in index order but products are rounded before they are added where
// the serial code fuses them — differences of rounding order only.
#pragma once
#include <hip/hip_runtime.h>
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace bqr {

template <int L>
__device__ __forceinline__ void fmac_bc(double &acc, double src0, double src1)
{   // acc += src0[lane L of this 16-lane row] * src1
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src0), "v"(src1), "n"(L));
}
template <int L>
__device__ __forceinline__ double bcast(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, false); }
// freshly written DPP sources: VALU write -> DPP read needs 2 wait states, not tracked into inline asm
__device__ __forceinline__ void fence(double &v) { asm volatile("s_nop 1" : "+v"(v)); }

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

// Σ_j v[lane j], j = 0..M-1 ascending, in every lane
template <int M>
__device__ __forceinline__ double rowsum(double v, double one)
{
    double s = 0.0;
    fence(v);
    sfor<0, M>([&](auto jc) { fmac_bc<decltype(jc)::value>(s, v, one); });
    return s;
}
// Σ_j v[lane j]·A[j]  (A: this lane's row of a matrix)
template <int M>
__device__ __forceinline__ double rowdot(double v, const double (&A)[M])
{
    double s = 0.0;
    fence(v);
    sfor<0, M>([&](auto jc) { constexpr int j = decltype(jc)::value; fmac_bc<j>(s, v, A[j]); });
    return s;
}

// State of one solve; every member is per lane (lane i = coordinate i).  No LDS: in the matrix-core kernel the other three waves of
// the work-group saturate it, and a round trip of this wave through it took microseconds.
template <int M>
struct Rows {
    double Hrow[M], Hcol[M];       // H[i][:], H[:][i]
    double Rcol[M], Rrow[M];       // R[:][i] (entries k <= i, diagonal included), R[i][:] (entries k > i only, diagonal excluded)
    double ri[M];                  // 1 / R[k][k], the same in every lane
};

// Upper Cholesky of H with clamped rows/columns replaced by identity (chol_masked_ri of boxqp_dev.h), column i in lane i; the row
// form is collected on the way: row j of R is what the lanes i > j have just computed, lane j keeps the broadcasts.  Returns 0 / j+1.
template <int M>
__device__ __forceinline__ int factor(Rows<M> &q, unsigned clamped, int i, bool in)
{
    int fail = 0;
    const bool ci = (clamped >> (i & 31)) & 1u;
#pragma unroll
    for (int k = 0; k < M; ++k) { q.Rcol[k] = 0.0; q.Rrow[k] = 0.0; }
    sfor<0, M>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const bool cj = (clamped >> j) & 1u;
        // s_i = Hm[j][i] - Σ_{k<j} R[k][j] R[k][i]   (lanes i >= j; lane j: the pivot)
        double acc = 0.0;
        sfor<0, j>([&](auto kc) { constexpr int k = decltype(kc)::value; fmac_bc<j>(acc, q.Rcol[k], q.Rcol[k]); });
        const double hm = (i == j) ? (cj ? 1.0 : q.Hcol[j]) : ((cj || ci) ? 0.0 : q.Hcol[j]);
        double s = hm - acc;
        fence(s);
        const double ajj = bcast<j>(s);
        if (!(ajj > 0.0) && fail == 0) fail = j + 1;
        const double r = ddp_rsqrt(ajj);
        q.ri[j] = r;
        q.Rcol[j] = (in && i >= j) ? ((i == j) ? ajj * r : s * r) : 0.0;
        fence(q.Rcol[j]);
        sfor<j + 1, M>([&](auto ic) {                           // R[j][i'] for i' > j into lane j
            constexpr int i2 = decltype(ic)::value;
            const double v = bcast<i2>(q.Rcol[j]);
            q.Rrow[i2] = (i == j) ? v : q.Rrow[i2];
        });
    });
    return fail;
}

// b <- (R'R)^{-1} b   (chol_solve_ri), b_i in lane i
template <int M>
__device__ __forceinline__ double solve(const Rows<M> &q, double b, int i)
{
    double acc = 0.0;                                           // forward: R' z = b
    sfor<0, M>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const double t = (b - acc) * q.ri[k];
        b = (i == k) ? t : b;
        fence(b);
        if constexpr (k + 1 < M) fmac_bc<k>(acc, b, (i > k) ? q.Rcol[k] : 0.0);
    });
    acc = 0.0;                                                  // backward: R y = z
    sfor<0, M>([&](auto kc) {
        constexpr int k = M - 1 - decltype(kc)::value;
        const double t = (b - acc) * q.ri[k];
        b = (i == k) ? t : b;
        fence(b);
        if constexpr (k > 0) fmac_bc<k>(acc, b, q.Rrow[k]);
    });
    return b;
}

// Returns `result` (boxQP.jl:172-179; 0 also for a failed factorisation).  x: this lane's coordinate of the solution; `clamped`: bit
// mask belonging to the factor left in q (quirk Q12 as in boxqp_dev_ri).  i = lane & 15; lanes i >= M pass neutral data.
template <int M>
__device__ __forceinline__ int boxqp_rows(Rows<M> &q, double g, double lower, double upper, double x0, const QPOptsDev &o, int i,
                                          double &x, unsigned &clamped, int &iters)
{
    const bool in = i < M;
    const unsigned all = (1u << M) - 1u;
    double one = 1.0;
    asm volatile("" : "+v"(one));
    int result = 0, iter = 1;
    clamped = 0u;
#pragma unroll
    for (int k = 0; k < M; ++k) { q.Rcol[k] = 0.0; q.Rrow[k] = 0.0; q.ri[k] = 0.0; }
    x = in ? ddp_clamp(x0, lower, upper) : 0.0;                                                    // :58
    double hx = rowdot<M>(x, q.Hrow);                                                              // (H x)_i
    double value = rowsum<M>(in ? x * g + (0.5 * x) * hx : 0.0, one);                              // :63
    double oldvalue = 0.0;
    while (iter <= o.maxIter) {                                                                    // :71
        if (result != 0) break;
        if (iter > 1 && (oldvalue - value) < o.minRelImprove * fabs(oldvalue)) { result = 4; break; }
        oldvalue = value;
        const double grad = g + hx;                                                                // :85
        const bool c = in && (((x == lower) && (grad > 0)) || ((x == upper) && (grad < 0)));       // :92-95
        const unsigned newc = (unsigned)(__ballot(c) & (unsigned long long)all);                   // row 0 of the wave: lanes 0..M-1
        const unsigned oldc = clamped;
        clamped = newc;
        if (clamped == all) { result = 6; break; }                                                 // :98-101
        if (iter == 1 || oldc != clamped) {                                                        // :104-117
            if (factor<M>(q, clamped, i, in) != 0) { result = 0; break; }
        }
        const double gn = rowsum<M>((in && !c) ? grad * grad : 0.0, one);                          // :120-124
        if (gn < o.minGrad * o.minGrad) { result = 5; break; }
        const double hxcl = rowdot<M>(c ? x : 0.0, q.Hrow);                                        // H (x .* clamped): every lane takes part in the DPP sums
        const double rhs = (in && !c) ? g + hxcl : 0.0;                                            // :127-129
        const double y = solve<M>(q, rhs, i);
        const double search = (in && !c) ? (-y - x) : 0.0;
        const double sdotg = rowsum<M>(search * grad, one);                                        // :132
        if (sdotg >= 0) break;                                                                     // :133-135
        double step = 1.0;                                                                         // :138-151
        double xc = in ? ddp_clamp(x + step * search, lower, upper) : 0.0;
        double hxc = rowdot<M>(xc, q.Hrow);
        double vc = rowsum<M>(in ? xc * g + (0.5 * xc) * hxc : 0.0, one);
        while ((vc - oldvalue) > o.Armijo * (step * sdotg)) {
            step = step * o.stepDec;
            xc = in ? ddp_clamp(x + step * search, lower, upper) : 0.0;
            hxc = rowdot<M>(xc, q.Hrow);
            vc = rowsum<M>(in ? xc * g + (0.5 * xc) * hxc : 0.0, one);
            if (step < o.minStep) { result = 2; break; }
        }
        x = xc; hx = hxc; value = vc;                                                              // :161-163
        iter += 1;
    }
    if (iter == o.maxIter) result = 1;                                                             // :167-169
    iters = iter;
    return result;
}

}   // namespace bqr
