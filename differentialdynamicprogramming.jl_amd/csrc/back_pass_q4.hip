// back_pass_q4.hip — backward pass for n = 4, m = 1 (the pendulum on a cart of BASELINE configs 3 and 5) on the fp64
// matrix cores: ONE TRAJECTORY PER BLOCK of v_mfma_f64_4x4x4_4b, four trajectories per wavefront, every 4x4 matrix of a
// Riccati step held as ONE element per lane.  Same arithmetic as src/backward_pass.jl:162-252 + :28-79 (all three rank
// dispatches: time-invariant operands are read with stride 0).
//
// Layout L: element [r][c] of a trajectory's matrix lives in lane 16 r + 4 b + c of a register (b = block = trajectory of the
// wave).  This is the D layout of the instruction; a register used as the B operand is read as it is, as the A operand it is
// read TRANSPOSED (profiles/microbench/mfma_4x4x4_layout_probe.hip, mfma_4x4x4_chain_probe.hip):
//        mm(X, Y, C) = X'·Y + C        per block, result in layout L again.
// Vectors are kept REPLICATED: "column form" v_c (lane [i][.] holds v[i]) and "row form" v_r (lane [.][j] holds v[j]); a
// replicated vector is a matrix with equal columns (rows), so matrix-vector products are the same instruction and deliver
// their result in every lane that needs it — no broadcast, no select, no data movement between lanes in the whole step:
//        W    = mm(V', fx)              = Vxx·fx                        W2 = mm(V, fu_c) = Vxx'·fu  (column form)
//        Qxx  = mm(fx, W, cxx)          = cxx + fx'Vxx fx               Qxx' = mm(W, fx, cxx') (bitwise the transpose)
//        Quu  = mm(fu_c, W2, cuu)       (every lane)                    Qu   = mm(fu_c, Vx_c, cu)      (every lane)
//        Qux_c = mm(fx, W2, cxu_c)      Qux_r = mm(W2, fx, cxu_r)       Qx_c = mm(fx, Vx_c, cx_c)
//        regType 2: fx'fu (both forms) and fu'fu by three products that do not depend on Vxx (off the chain)
//        gains: scalar (m = 1), every lane computes them; boxQP is a straight-line restatement of the first two
//               projected-Newton iterations of boxQP.jl:58-169 (where the m = 1 problem ends unless the Armijo search has to
//               back off or rounding leaves a gradient above minGrad) with the generic loop (boxqp_dev1) as the fall-back
//        value: P = Qxx + K_c∘T_r + T_c∘K_r  elementwise (T = Qux + ½Quu·K: K T' + T K' = K'QuuK + K'Qux + Qux'K, :70),
//               P' likewise from Qxx', Vxx_i = ½(P + P') exactly symmetric (:71-72).
// 9 (regType 2: 12) matrix instructions and ~90 vector instructions per step of FOUR trajectories; the 16-lane DPP-row
// kernel this replaces for the shape (back_pass_dpp.hip) issued ~280 vector + ~130 scalar instructions, most of them the
// divergent generic boxQP loop (profiles/r02_c3_baseline_pmc.txt).
//
// Memory: at one wave per SIMD the kernel is bound by the NUMBER of vector-memory instructions (the address unit of a CU takes
// ~16 cycles per wave instruction and is shared by its four SIMDs; measured: the first version with 9 loads + 2 stores per
// step ran 0.82 ms, 0.39 ms with the memory instructions removed).  back_pass_q4p_kernel therefore works on PAIRS of time
// steps: rows 0, 2 of a block load the 16-byte pair {M[r], M[r+1]} of step 2j+1 while rows 1, 3 load {M[r-1], M[r]} of step
// 2j, and one v_permlane16_swap per dword turns the two halves into the two matrices; results go back the same way
// (Vxx_i pairs; K_i, Vx_i, k_i, Quu_i merged into ONE 16-byte store): 3.5 memory instructions per step.  It needs an even
// horizon N (16-byte alignment of the [1,N] arrays) and time-invariant cost; back_pass_q4_kernel (one step at a time, any
// N, time-varying cost) covers the rest.
#include <stdlib.h>
#include "ddp_internal.h"
#include "boxqp_dev.h"

#ifndef Q4_DP
#define Q4_DP 4
#endif
namespace {

struct Q4Args {
    int N, B;
    int regType;
    // element strides (doubles): per time step (0 when time-invariant) and per trajectory (0 when shared)
    int fx_t, fu_t, cxx_t, cxu_t, cuu_t;
    long fx_b, fu_b, cxx_b, cxu_b, cuu_b;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
    const double *eta;          // back_pass_gps only: η per trajectory
    double *Quui;               // back_pass_gps only (chunked kernel): inv(Quu)
    double *sink;               // >= 64 x 16 B that lanes without an output may write (paired kernel: stores carry no exec-mask branch)
};

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double mm(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
#ifdef Q4_COUNT_SLOW
__device__ unsigned long long q4_slow_count = 0ull, q4_step_count = 0ull, q4_slow_why[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
#endif

// (a, b) -> (x, y): x = rows [a0, b0, a2, b2], y = rows [a1, b1, a3, b3] of the 16-lane rows (v_permlane16_swap per dword)
__device__ __forceinline__ void row_swap(double a, double b, double &x, double &y)
{
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    const u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    const u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    x = __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x));
    y = __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
}

// x = the lower 32 lanes' value in every lane, y = the upper 32 lanes' (v_permlane32_swap per dword)
__device__ __forceinline__ void half_swap(double a, double &x, double &y)
{
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a);
    const u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ua, false, false);
    const u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ua >> 32), false, false);
    x = __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x));
    y = __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
}

// First two iterations of boxQP(H,g,lower,upper,x0) for m = 1 (boxQP.jl:58-169) as straight-line code, arithmetic of
// boxqp_dev1 (boxqp_dev.h).  Decided here: result 6 in the first iteration (x0 clamped, gradient pointing outwards) and, after
// one accepted full Newton step, results 4, 6, 5 of the second iteration.  Everything else — a non-positive H, a gradient below
// minGrad at the start, no descent, an Armijo back-off, a third iteration — is `slow` and goes to the generic loop.
__device__ __forceinline__ void boxqp1_two_iterations(double H, double g, double lower, double upper, double x0, const QPOptsDev &o,
                                                       double &x, double &rH, bool &clamped, bool &slow)
{
    auto val = [&](double xx) { return xx * g + ((0.5 * xx) * H) * xx; };                      // :63
    // clamp by v_max/v_min: like Base.clamp for ordered operands; a NaN iterate (only possible after an earlier step of the
    // trajectory has already failed) comes out as a bound instead of NaN, and the generic loop still decides such a case
    auto clampf = [](double v, double lo_, double hi_) { return fmin(fmax(v, lo_), hi_); };
    const double x1 = clampf(x0, lower, upper);                                                 // :58
    const double v1 = val(x1);
    const double grad1 = g + H * x1;                                                            // :85
    // (the tests are combined with & and |, not && and ||: short-circuit evaluation costs an exec-mask branch per term here)
    const bool c1 = bool(((x1 == lower) & (grad1 > 0)) | ((x1 == upper) & (grad1 < 0)));        // :92-95 -> result 6 (:98-101)
    rH = ddp_rcp_nr(H);
    const double search = -(g * rH) - x1;                                                       // :127-129
    const double sdotg = search * grad1;                                                        // :132
    double xc = clampf(x1 + search, lower, upper);                                              // step = 1 (:138-141)
    double vc = val(xc);
    // iteration 1 runs to its end: H > 0 (:111), |grad| >= minGrad (:120), sdotg < 0 (:133), and the line search accepts a step (:142-151)
    const bool descent = bool((H > 0.0) & !(fabs(grad1) < o.minGrad) & (sdotg < 0));
    bool need = bool(descent & ((vc - v1) > o.Armijo * sdotg));                                 // Armijo fails at step 1
    // The back-tracking line search (:142-151) in closed form for the case that makes it long.  It is not rare: when the Newton point lies
    // beyond a bound the projected step is shorter than the model promises, Armijo fails, and at BASELINE config 3 (pendulum, limits +-5)
    // 18 % of all trajectory-steps — at four trajectories per wave about half of the wave-steps — left through here into the generic
    // loop, many of them for TENS of step sizes (a warm start near the bound: 0.6^k has to fall to ~(v1 - v(bound)) / (Armijo |s'g|)),
    // 285 ns on top of a 360 ns step (profiles/microbench/q4c_chain_floor.hip, profiles/q4_slow_count.py).  As long as the ray
    // x1 + step·search stays outside the box the projected point IS the bound and its value does not change with the step, so the loop
    // ends at x = bound as soon as step <= s* = (v1 - v(bound)) / (Armijo |s'g|), whatever the number of step sizes — provided it gets
    // there before the ray re-enters the box at s_in = |bound - x1| / |search|.  The accepted step is the largest 0.6^k <= s*, which is
    // > 0.6 s*: 0.6 s* >= s_in is sufficient (tested with 0.59 and without divisions; what is nearer goes to the generic loop, as do a
    // value at the bound that is no improvement and step sizes near minStep).
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
        const double bnd = search > 0 ? upper : lower;
        const double vbn = val(bnd);
        const double dv = v1 - vbn, ps = -(o.Armijo * sdotg), dist = fabs(bnd - x1), as = fabs(search);
        const bool closed = bool(need & (xc == bnd) & (dv > 0.0) & (((0.59 * dv) * as) >= (dist * ps)) & (dv > 1e-21 * ps));
        xc = closed ? bnd : xc; vc = closed ? vbn : vc;
        need = bool(need & !closed);
    }
    const bool plain = bool(descent & !need);
    // second iteration
    const bool relimp = (v1 - vc) < o.minRelImprove * fabs(v1);                                 // result 4 (:78-81), free set of iteration 1
    const double grad2 = g + H * xc;
    const bool c2 = bool(((xc == lower) & (grad2 > 0)) | ((xc == upper) & (grad2 < 0)));        // result 6
    const bool small2 = fabs(grad2) < o.minGrad;                                                // result 5
    slow = bool(!c1 & !(plain & (relimp | c2 | small2)));
#ifdef Q4_COUNT_SLOW      // why: [0] H <= 0, [1] |grad| < minGrad at the warm start, [2] no descent, [3] Armijo back-off, [4] a third iteration
    if (slow && (threadIdx.x & 15) == 0) {
        const int why = !(H > 0.0) ? 0 : (fabs(grad1) < o.minGrad ? 1 : (!(sdotg < 0) ? 2 : (((vc - v1) > o.Armijo * sdotg) ? 3 : 4)));
        atomicAdd(&q4_slow_why[why], 1ull);
    }
#endif
    x = c1 ? x1 : xc;
    clamped = bool(c1 | (!relimp & c2));
}

struct Q4In { double fx, fu, cx, cu, u, cxx, cxxT, cxuc, cxur, cuu; };    // operands of one step (layout L / column / row forms)
struct Q4State { double V, VT, vxc, kprev, dV0, dV1; int diverge; };        // Vxx_{i+1} (and transposed), Vx_{i+1} column form
struct Q4Out { double Vn, Kc, vx, kk, Quu; };
struct Q4Par { double lam, limlo, limhi; bool nolims; double ieta; };

struct Q4NoMid { __device__ __forceinline__ void operator()() const {} };


// `mid` runs once the matrix instructions of the step have been issued (the LDS-chunk kernel puts the LDS writes of the PREVIOUS
// step there: behind the products their latency costs nothing, in front of them it delays the first product)
// GPS (back_pass_gps, backward_pass.jl:290-307): Q• = (c• + f'V f)/η + c•kl.  The caller hands in c̃• = c•/η + c•kl (a prepass), so what is
// left is the factor 1/η on the products with V: on Vx and on W = Vxx fx, W2 = Vxx'fu — three multiplications per step.
template <bool LIMS, bool REG2, int EXP, class Mid = Q4NoMid, bool GPS = false>
__device__ __forceinline__ void q4_step(int i, const Q4In &o, Q4State &s, Q4Out &out, const Q4Par &p, Mid mid = Mid())
{
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};                                   // boxQP.jl:30-35
    // ---- products that do not depend on Vxx (regType 2: Vxx_reg = Vxx + λI adds λ·fu'fx, λ·fu'fu; :245-247)
    double Fc = 0.0, Fr = 0.0, ff = 0.0;
    if (REG2) {
        Fc = mm(o.fx, o.fu, 0.0);                               // (fx'fu)[i], column form
        Fr = mm(o.fu, o.fx, 0.0);                               // (fu'fx)[j], row form
        ff = mm(o.fu, o.fu, 0.0);                               // fu'fu
    }
    // ---- Q-function expansion (:165-169 / :203-210 / :240-244)
    const double vxs = GPS ? s.vxc * p.ieta : s.vxc;
    const double Qu = mm(o.fu, vxs, o.cu);
    const double Qxc = mm(o.fx, vxs, o.cx);
    double W2 = mm(s.V, o.fu, 0.0);                             // Vxx'·fu: (fu'Vxx)' in column form
    double W = mm(s.VT, o.fx, 0.0);                             // Vxx·fx
    if (GPS) { W2 *= p.ieta; W *= p.ieta; }
    const double Quu = mm(o.fu, W2, o.cuu);
    const double Quxc = mm(o.fx, W2, o.cxuc);
    const double Quxr = mm(W2, o.fx, o.cxur);
    const double Qxx = mm(o.fx, W, o.cxx);
    const double QxxT = mm(W, o.fx, o.cxxT);
    mid();
    const double QuuF = Quu + (REG2 ? p.lam * ff : p.lam);    // :247
    const double Qrc = Quxc + (REG2 ? p.lam * Fc : 0.0);      // Qux_reg, both forms (:246)
    const double Qrr = Quxr + (REG2 ? p.lam * Fr : 0.0);
    // ---- gains (:31-61), scalar system
    double kk, rH;
    bool clamped = false, fail;
    if (!LIMS || p.nolims || (EXP & 4)) {
        fail = !(QuuF > 0.0);                                   // cholesky(Hermitian(QuuF)) (:35)
        rH = ddp_rcp_nr(QuuF);
        kk = -(Qu * rH);                                        // k_i = -(R\Qu) (:41)
    } else {
        const double lo = p.limlo - o.u, up = p.limhi - o.u;    // :45-46
        bool slow;
        boxqp1_two_iterations(QuuF, Qu, lo, up, s.kprev, qpo, kk, rH, clamped, slow);     // :49 (warm start k[:, min(i+1,N-1)])
        fail = false;
#ifdef Q4_COUNT_SLOW      // profiling builds: how often the two-iteration form hands a step to the generic loop (ddp_q4_slow_count)
        if (slow && s.diverge == 0 && (threadIdx.x & 15) == 0) atomicAdd(&q4_slow_count, 1ull);
        if ((threadIdx.x & 15) == 0) atomicAdd(&q4_step_count, 1ull);
#endif
        if (__builtin_expect(slow && s.diverge == 0, 0)) {                           // rare: the generic loop decides
            unsigned cl; int iters;
            const int result = boxqp_dev1(QuuF, Qu, lo, up, s.kprev, qpo, kk, rH, cl, iters);
            clamped = (cl & 1u) != 0u;
            fail = result < 1;                                  // :53
        }
    }
    const bool alive = s.diverge == 0 && !fail;
    if (s.diverge == 0 && fail) s.diverge = i + 1;              // :37-38, :54-55
    const double nrH = clamped ? 0.0 : -rH;
    const double Kc = Qrc * nrH, Kr = Qrr * nrH;                // K_i = -(R\Qux_reg), clamped rows zero (:42, :57-61)
    // ---- value update (:64-72)
    const double Quuk = Quu * kk;
    s.dV0 += alive ? kk * Qu : 0.0;                             // :68
    s.dV1 += alive ? (0.5 * kk) * Quuk : 0.0;
    const double vx = Qxc + Kc * (Quuk + Qu) + Quxc * kk;       // :69
    const double hQ = 0.5 * Quu;
    const double Tc = Quxc + hQ * Kc, Tr = Quxr + hQ * Kr;      // K T' + T K' = K'QuuK + K'Qux + Qux'K
    const double P = Kc * Tr + (Tc * Kr + Qxx);                 // :70
    const double Pt = Kr * Tc + (Tr * Kc + QxxT);               // the same sums for the transposed element
    const double Vn = 0.5 * (P + Pt);                           // :71-72
    out.Vn = Vn; out.Kc = Kc; out.vx = vx; out.kk = kk; out.Quu = Quu;
    s.V = Vn; s.VT = Vn; s.vxc = vx; s.kprev = kk;
}

// outputs earlier in time than a failing step are zero (backward_pass.jl:37-38 with :226-229); Quu of the failing step itself
// stays (assigned before the failure), earlier Quu is `undef` upstream, zero here.  The 16 lanes of a block share the work.
__device__ __forceinline__ void q4_zero_fill(const Q4Args &a, int b, int q16, int diverge)
{
    const int N = a.N;
    const size_t ie = (size_t)diverge;                          // = failing 0-based step + 1
    double *Kg = a.K + (size_t)4 * N * b, *kg = a.k + (size_t)N * b, *Vxg = a.Vx + (size_t)4 * N * b,
           *Vxx0 = a.Vxx + (size_t)16 * N * b, *Quug = a.Quu + (size_t)N * b;
    for (size_t t = q16; t < 4 * ie; t += 16) { Kg[t] = 0.0; Vxg[t] = 0.0; }
    for (size_t t = q16; t < ie; t += 16) kg[t] = 0.0;
    for (size_t t = q16; t < 16 * ie; t += 16) Vxx0[t] = 0.0;
    for (size_t t = q16; t + 1 < ie; t += 16) Quug[t] = 0.0;
}

// ---- one time step at a time: any N, time-varying cost (CTV) or not
template <bool LIMS, bool CTV, bool REG2, int EXP = 0, bool GPS = false>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_q4_kernel(Q4Args a)
{
    constexpr int n = 4, D = 8;
    const int N = a.N;
    const int lane = threadIdx.x, r = lane >> 4, blk = (lane >> 2) & 3, c = lane & 3, q16 = 4 * r + c;
    long tb = (long)blockIdx.x * 4 + blk;
    const bool valid = tb < a.B;
    if (!valid) tb = a.B - 1;                                   // idle blocks repeat the last trajectory (no stores)
    const int b = (int)tb;
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;
    const int e = r + 4 * c, et = c + 4 * r;                    // column-major offsets of [r][c] and [c][r]

    const double *cx = a.cx + (size_t)n * N * b + r, *cu = a.cu + (size_t)N * b;
    const double *ug = LIMS ? a.u + (size_t)N * b : cu;
    const double *fx = a.fx + a.fx_b * b + e, *fu = a.fu + a.fu_b * b + r;
    const double *cxx = a.cxx + a.cxx_b * b, *cuu = a.cuu + a.cuu_b * b, *cxu = a.cxu + a.cxu_b * b;
    double *Vxxg = a.Vxx + (size_t)16 * N * b + e;
    // merged store of K_i[0, r] (lanes [r][0]), Vx_i[r] (lanes [r][1]), k_i (lane [0][2]), Quu_i (lane [1][2])
    const bool st_K = c == 0, st_Vx = c == 1, st_k = (c == 2 && r == 0), st_Q = (c == 2 && r == 1);
    const bool st_on = act && (st_K || st_Vx || st_k || st_Q);
    double *st_base = st_K ? a.K + (size_t)n * N * b + r : st_Vx ? a.Vx + (size_t)n * N * b + r : st_k ? a.k + (size_t)N * b : a.Quu + (size_t)N * b;
    const unsigned st_stride = (st_K || st_Vx) ? n * 8u : 8u;

    Q4Par par;
    par.lam = GPS ? 0.0 : a.lambda[b]; par.limlo = 0.0; par.limhi = 0.0; par.nolims = true;     // back_pass_gps: η is the only regularisation
    par.ieta = GPS ? 1.0 / a.eta[b] : 1.0;
    if (LIMS) { par.limlo = a.lims[0]; par.limhi = a.lims[1]; par.nolims = par.limlo > par.limhi; }     // backward_pass.jl:31

    // ---- terminal step (backward_pass.jl:21-23 / :197-199 / :234-236).  Vxx_i is exactly symmetric for i < N-1; the terminal cxx
    // is used as given (a non-symmetric one enters fx'Vxx fx and fu'Vxx fx like upstream), hence the transposed copy VT.
    const int tl = N - 1;
    Q4State s;
    s.V = cxx[(size_t)a.cxx_t * tl + e]; s.VT = cxx[(size_t)a.cxx_t * tl + et];
    s.vxc = cx[(size_t)n * tl];
    s.kprev = 0.0; s.dV0 = 0.0; s.dV1 = 0.0; s.diverge = 0;
    if (act) {
        Vxxg[(size_t)16 * tl] = s.V;
        if (st_on) {
            const double v0 = st_K ? 0.0 : st_Vx ? s.vxc : st_k ? 0.0 : cuu[(size_t)a.cuu_t * tl];
            *(double *)((char *)st_base + (size_t)tl * st_stride) = v0;
        }
    }
    Q4In cst;                                                   // time-invariant cost: read once
    cst.cxx = cxx[e]; cst.cxxT = cxx[et]; cst.cxuc = cxu[r]; cst.cxur = cxu[c]; cst.cuu = cuu[0];

    auto fetch = [&](int i, Q4In &o) {
        o.fx = fx[(size_t)(unsigned)(a.fx_t * i)]; o.fu = fu[(size_t)(unsigned)(a.fu_t * i)];
        o.cx = cx[(size_t)(unsigned)(n * i)]; o.cu = cu[i]; o.u = ug[i];
        if (CTV) {
            o.cxx = cxx[(size_t)(unsigned)(a.cxx_t * i) + e]; o.cxxT = cxx[(size_t)(unsigned)(a.cxx_t * i) + et];
            o.cxuc = cxu[(size_t)(unsigned)(a.cxu_t * i) + r]; o.cxur = cxu[(size_t)(unsigned)(a.cxu_t * i) + c];
            o.cuu = cuu[(size_t)(unsigned)(a.cuu_t * i)];
        }
    };
    auto step = [&](int i, Q4In &o) __attribute__((always_inline)) {
        if (!CTV) { o.cxx = cst.cxx; o.cxxT = cst.cxxT; o.cxuc = cst.cxuc; o.cxur = cst.cxur; o.cuu = cst.cuu; }
        Q4Out out;
        q4_step<LIMS, REG2, EXP, Q4NoMid, GPS>(i, o, s, out, par);
        // ---- stores (a diverged trajectory keeps writing; its range is zero-filled after the loop)
        if (act && !(EXP & 1)) Vxxg[(size_t)(unsigned)(16 * i)] = out.Vn;
        const double v0 = st_K ? out.Kc : st_Vx ? out.vx : st_k ? out.kk : out.Quu;       // :75-76
        if (st_on && !(EXP & 1)) *(double *)((char *)st_base + (size_t)((unsigned)i * st_stride)) = v0;
    };

    if (N >= 2) {
        Q4In ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { const int i = N - 2 - d; fetch(i >= 0 ? i : 0, ring[d]); }
        int i0 = N - 2;
        for (; i0 - (2 * D - 1) >= 0; i0 -= D) {                // branch-free groups of D steps
#pragma unroll
            for (int d = 0; d < D; ++d) {
                step(i0 - d, ring[d]);
                if (!(EXP & 2)) fetch(i0 - d - D, ring[d]);
            }
        }
        for (; i0 >= 0; i0 -= D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int i = i0 - d;
                if (i >= 0) {
                    step(i, ring[d]);
                    if (!(EXP & 2)) fetch(i - D >= 0 ? i - D : 0, ring[d]);
                }
            }
        }
        if (s.diverge && act) q4_zero_fill(a, b, q16, s.diverge);
    }
    if (act && q16 == 0) { a.dV[2 * b] = s.dV0; a.dV[2 * b + 1] = s.dV1; a.diverge[b] = s.diverge; }
}

// ---- two time steps per memory instruction (even N, time-invariant cost)
template <bool LIMS, bool REG2, int EXP = 0>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_q4p_kernel(Q4Args a)
{
    constexpr int n = 4, DP = Q4_DP;                            // ring of DP pairs
    const int N = a.N, NP = N / 2;
    const int lane = threadIdx.x, r = lane >> 4, blk = (lane >> 2) & 3, c = lane & 3, q16 = 4 * r + c;
    long tb = (long)blockIdx.x * 4 + blk;
    const bool valid = tb < a.B;
    if (!valid) tb = a.B - 1;
    const int b = (int)tb;
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;
    const int e = r + 4 * c, et = c + 4 * r;
    const int odd = r & 1, re = r & ~1;                         // odd rows serve the even step 2j of a pair, even rows step 2j+1

    // pair p = steps (2p+1, 2p); per-lane bases already point at this lane's step of pair 0
    const double *fxp = a.fx + a.fx_b * b + (re + 4 * c) + (odd ? 0 : a.fx_t);
    const double *fup = a.fu + a.fu_b * b + re + (odd ? 0 : a.fu_t);
    const double *cxp = a.cx + (size_t)n * N * b + re + (odd ? 0 : n);
    // cu and u pairs come with ONE load: rows 0, 1 of a block fetch {cu[2p], cu[2p+1]}, rows 2, 3 {u[2p], u[2p+1]}; v_permlane32_swap
    // hands both to every lane
    const double *cup = (LIMS && r >= 2) ? a.u + (size_t)N * b : a.cu + (size_t)N * b;
    const double *cxx = a.cxx + a.cxx_b * b, *cuu = a.cuu + a.cuu_b * b, *cxu = a.cxu + a.cxu_b * b;
    // stores are UNCONDITIONAL: a lane without an output (inactive trajectory, lanes outside the merged store) writes its own 16 bytes
    // of the sink instead (stride 0) — an `if` around a store costs s_and_saveexec + branch + s_or per store and step
    double *Vxxp = act ? a.Vxx + (size_t)16 * N * b + (re + 4 * c) + (odd ? 0 : 16) : a.sink + 2 * lane;
    const unsigned vxx_stride = act ? 32u * 8u : 0u;            // bytes per pair
    // merged 16-byte store per pair: lanes [.][0]: K pairs, [.][1]: Vx pairs (row parity selects the step like the loads),
    // lane [0][2]: {k[2p], k[2p+1]}, lane [0][3]: {Quu[2p], Quu[2p+1]}
    const bool st_K = c == 0, st_Vx = c == 1, st_s = (r == 0 && c >= 2);
    const bool st_on = act && (st_K || st_Vx || st_s);
    double *st_base = !st_on ? a.sink + 2 * lane
                      : st_K ? a.K + (size_t)n * N * b + re + (odd ? 0 : n) : st_Vx ? a.Vx + (size_t)n * N * b + re + (odd ? 0 : n)
                      : (c == 2) ? a.k + (size_t)N * b : a.Quu + (size_t)N * b;
    const unsigned st_stride = !st_on ? 0u : (st_K || st_Vx) ? 2u * n * 8u : 16u;      // bytes per pair

    Q4Par par;
    par.lam = a.lambda[b]; par.limlo = 0.0; par.limhi = 0.0; par.nolims = true; par.ieta = 1.0;
    if (LIMS) { par.limlo = a.lims[0]; par.limhi = a.lims[1]; par.nolims = par.limlo > par.limhi; }

    Q4In cst;
    cst.cxx = cxx[e]; cst.cxxT = cxx[et]; cst.cxuc = cxu[r]; cst.cxur = cxu[c]; cst.cuu = cuu[0];

    struct Pair { d2 fx, fu, cx, cu; };
    auto fetch = [&](int p, Pair &o) {
        o.fx = *(const d2 *)(fxp + (size_t)(unsigned)(2 * a.fx_t * p));
        o.fu = *(const d2 *)(fup + (size_t)(unsigned)(2 * a.fu_t * p));
        o.cx = *(const d2 *)(cxp + (size_t)(unsigned)(2 * n * p));
        o.cu = *(const d2 *)(cup + (size_t)(unsigned)(2 * p));
    };
    auto unpack = [&](const Pair &o, Q4In &A, Q4In &Bs) {
        row_swap(o.fx.x, o.fx.y, A.fx, Bs.fx);
        row_swap(o.fu.x, o.fu.y, A.fu, Bs.fu);
        row_swap(o.cx.x, o.cx.y, A.cx, Bs.cx);
        half_swap(o.cu.y, A.cu, A.u);                           // lower half of the wave holds cu, upper half u
        half_swap(o.cu.x, Bs.cu, Bs.u);
        A.cxx = Bs.cxx = cst.cxx; A.cxxT = Bs.cxxT = cst.cxxT; A.cxuc = Bs.cxuc = cst.cxuc; A.cxur = Bs.cxur = cst.cxur;
        A.cuu = Bs.cuu = cst.cuu;
    };
    auto store_pair = [&](int p, const Q4Out &oa, const Q4Out &ob) {
        if (EXP & 1) return;
        double x, y;
        row_swap(oa.Vn, ob.Vn, x, y);
        *(d2 *)((char *)Vxxp + (size_t)((unsigned)p * vxx_stride)) = d2{x, y};
        // lanes [0][2], [0][3] store {value of step 2p, value of step 2p+1}: after the swap row 0 holds x = a(row 0), y = a(row 1)
        const double sa = (c == 2) ? (r == 0 ? ob.kk : oa.kk) : (r == 0 ? ob.Quu : oa.Quu);
        const double va = st_K ? oa.Kc : st_Vx ? oa.vx : sa;
        const double vb = st_K ? ob.Kc : ob.vx;
        row_swap(va, vb, x, y);
        *(d2 *)((char *)st_base + (size_t)((unsigned)p * st_stride)) = d2{x, y};
    };

    Q4State s;
    s.kprev = 0.0; s.dV0 = 0.0; s.dV1 = 0.0; s.diverge = 0;
    Pair ring[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) { const int p = NP - 1 - d; fetch(p >= 0 ? p : 0, ring[d]); }
    // ---- first pair: its upper member is the terminal step (backward_pass.jl:21-23 / :234-236), see back_pass_q4_kernel
    {
        Q4In A, Bs;
        unpack(ring[0], A, Bs);
        Q4Out oa, ob;
        s.V = cst.cxx; s.VT = cst.cxxT; s.vxc = A.cx;
        oa.Vn = s.V; oa.Kc = 0.0; oa.vx = A.cx; oa.kk = 0.0; oa.Quu = cst.cuu;
        q4_step<LIMS, REG2, EXP>(N - 2, Bs, s, ob, par);
        store_pair(NP - 1, oa, ob);
        fetch(NP - 1 - DP >= 0 ? NP - 1 - DP : 0, ring[0]);
    }
    auto pair_step = [&](int p, Pair &slot, int pnext) __attribute__((always_inline)) {
        Q4In A, Bs;
        unpack(slot, A, Bs);
        Q4Out oa, ob;
        q4_step<LIMS, REG2, EXP>(2 * p + 1, A, s, oa, par);
        q4_step<LIMS, REG2, EXP>(2 * p, Bs, s, ob, par);
        store_pair(p, oa, ob);
        if (!(EXP & 2)) fetch(pnext, slot);
    };
    // slot of pair p is (NP-1-p) % DP; the first pair used slot 0
    int p0 = NP - 2;
    // leading group so that the main loop starts at slot 0 again: slots 1 .. DP-1.  Unconditional (the launcher guarantees NP >= DP):
    // a conditionally issued load does not count for the compiler's s_waitcnt vmcnt(n) bookkeeping, and a small n at the top of
    // the main loop would drain the whole prefetch ring in every iteration (measured: a third of the kernel time).
#pragma unroll
    for (int d = 1; d < DP; ++d) {
        const int p = NP - 1 - d;
        pair_step(p, ring[d], p - DP >= 0 ? p - DP : 0);
    }
    p0 = NP - 1 - DP;
    for (; p0 - (2 * DP - 1) >= 0; p0 -= DP) {                  // branch-free groups of DP pairs
#pragma unroll
        for (int d = 0; d < DP; ++d) pair_step(p0 - d, ring[d], p0 - d - DP);
    }
    for (; p0 >= 0; p0 -= DP) {
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            const int p = p0 - d;
            if (p >= 0) pair_step(p, ring[d], p - DP >= 0 ? p - DP : 0);
        }
    }
    if (s.diverge && act) q4_zero_fill(a, b, q16, s.diverge);
    if (act && q16 == 0) { a.dV[2 * b] = s.dV0; a.dV[2 * b + 1] = s.dV1; a.diverge[b] = s.diverge; }
}

// ---- CHUNKS OF EIGHT TIME STEPS THROUGH THE LDS (N a multiple of 8, time-varying fx/fu, time-invariant cost)
// A vector-memory instruction costs a lone wave ~60 issue cycles and the pair kernel above still spends ~50 further instructions
// per step on addresses, lane swaps and the select tree of its merged store.  Here the global side moves whole chunks: seven
// direct-to-LDS loads (global_load_lds_dwordx4: fx in four pieces, fu, cx, cu|u) fetch the operands of 8 steps x 4 trajectories one
// chunk ahead, seven 16-byte stores write their results back; per step the wave only issues LDS instructions whose addresses are
// one register + an immediate:  3 reads (fx; fu, cx; cu, u) and 3 writes (Vxx; K, Vx; k, Quu).  Replicated values are written by
// their first lane only; the other lanes aim at a dump area, so no select and no exec mask is needed.
constexpr int Q4L_CH = 8;
constexpr int Q4L_IN = 896;          // doubles per input buffer: fx [4 pieces][4 traj][32] | fu [4][32] | cx [4][32] | cu,u [4][32] (8 + 8 used)
constexpr int Q4L_IFU = 512, Q4L_ICX = 640, Q4L_ICU = 768;
// time-varying cost (CTV; back_pass_gps: the combined c̃xx, c̃xu, c̃uu of the prepass): cxx like fx [4 pieces][4 traj][32] | cxu [4][32] | cuu [4][32] (8 used)
constexpr int Q4L_IXX = Q4L_IN, Q4L_IXU = Q4L_IXX + 512, Q4L_IUU = Q4L_IXU + 128, Q4L_INC = Q4L_IUU + 128;
constexpr int Q4L_OK = 512;          // outputs: Vxx [4][4][32] | K [4][32] + dump 96 | Vx [4][32] + dump 96 | k,Quu [4][16] + dump 144
constexpr int Q4L_KD = 224;          // distance K -> Vx block (the two offsets of one ds_write2_b64)
constexpr int Q4L_OKQ = Q4L_OK + 2 * Q4L_KD, Q4L_OUT = Q4L_OKQ + 64 + 144;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <bool LIMS, bool REG2, bool CTV = false, bool GPS = false>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_q4l_kernel(Q4Args a)
{
    constexpr int n = 4, CH = Q4L_CH;
    __shared__ __attribute__((aligned(16))) double lin[2][CTV ? Q4L_INC : Q4L_IN];
    __shared__ __attribute__((aligned(16))) double lout[Q4L_OUT];
    const int N = a.N, NC = N / CH;
    const int lane = threadIdx.x, r = lane >> 4, blk = (lane >> 2) & 3, c = lane & 3, q16 = 4 * r + c;
    long tb = (long)blockIdx.x * 4 + blk;
    const bool valid = tb < a.B;
    if (!valid) tb = a.B - 1;
    const int b = (int)tb;
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;
    const int e = r + 4 * c, et = c + 4 * r;
    // the lane's trajectory on the global side (lane-linear LDS image: 16 lanes x 16 bytes per trajectory and piece)
    const int tl = lane >> 4, q = lane & 15;
    long tbd = (long)blockIdx.x * 4 + tl;
    const bool validd = tbd < a.B;
    if (!validd) tbd = a.B - 1;
    const int bd = (int)tbd;
    const bool actd = validd && !(a.active && a.active[bd] == 0);

    const double *gfx = a.fx + a.fx_b * bd + 2 * q, *gfu = a.fu + a.fu_b * bd + 2 * q, *gcx = a.cx + (size_t)n * N * bd + 2 * q;
    const double *gcu = ((LIMS && q >= 4) ? a.u : a.cu) + (size_t)N * bd + 2 * (q & 3);
    const double *gxx = a.cxx + a.cxx_b * bd + 2 * q, *gxu = a.cxu + a.cxu_b * bd + 2 * q, *guu = a.cuu + a.cuu_b * bd + 2 * (q & 3);     // (CTV)
    auto dma = [&](int ch, double *in) {
        if (CTV) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((glb_void *)(gxx + (size_t)ch * (16 * CH) + 32 * j), (lds_void *)(in + Q4L_IXX + 128 * j), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(gxu + (size_t)ch * (4 * CH)), (lds_void *)(in + Q4L_IXU), 16, 0, 0);
            if (q < 4) __builtin_amdgcn_global_load_lds((glb_void *)(guu + (size_t)ch * CH), (lds_void *)(in + Q4L_IUU), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((glb_void *)(gfx + (size_t)ch * (16 * CH) + 32 * j), (lds_void *)(in + 128 * j), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(gfu + (size_t)ch * (4 * CH)), (lds_void *)(in + Q4L_IFU), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(gcx + (size_t)ch * (4 * CH)), (lds_void *)(in + Q4L_ICX), 16, 0, 0);
        if (q < 8) __builtin_amdgcn_global_load_lds((glb_void *)(gcu + (size_t)ch * CH), (lds_void *)(in + Q4L_ICU), 16, 0, 0);
    };
    // global side of the results: a lane without an output (inactive trajectory) writes its 16 bytes of the sink
    double *gV = actd ? a.Vxx + (size_t)16 * N * bd + 2 * q : a.sink + 2 * lane;
    double *gK = actd ? a.K + (size_t)n * N * bd + 2 * q : a.sink + 2 * lane;
    double *gX = actd ? a.Vx + (size_t)n * N * bd + 2 * q : a.sink + 2 * lane;
    double *gS = actd ? ((q >= 4) ? a.Quu : a.k) + (size_t)N * bd + 2 * (q & 3) : a.sink + 2 * lane;
    const size_t on = actd ? 1 : 0;
    auto drain = [&](int ch) {
        d2 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *(const d2 *)(lout + 128 * j + 2 * lane);
        const d2 vk = *(const d2 *)(lout + Q4L_OK + 2 * lane), vx = *(const d2 *)(lout + Q4L_OK + Q4L_KD + 2 * lane);
        const d2 vs = *(const d2 *)(lout + Q4L_OKQ + 16 * tl + 2 * (q & 7));
#pragma unroll
        for (int j = 0; j < 4; ++j) *(d2 *)(gV + on * ((size_t)ch * (16 * CH) + 32 * j)) = v[j];
        *(d2 *)(gK + on * ((size_t)ch * (4 * CH))) = vk;
        *(d2 *)(gX + on * ((size_t)ch * (4 * CH))) = vx;
        if (q < 8) *(d2 *)(gS + on * ((size_t)ch * CH)) = vs;
    };

    Q4Par par;
    par.lam = GPS ? 0.0 : a.lambda[b]; par.limlo = 0.0; par.limhi = 0.0; par.nolims = true;      // back_pass_gps: η is the only regularisation
    par.ieta = GPS ? 1.0 / a.eta[b] : 1.0;
    if (LIMS) { par.limlo = a.lims[0]; par.limhi = a.lims[1]; par.nolims = par.limlo > par.limhi; }
    const double *cxx = a.cxx + a.cxx_b * b, *cuu = a.cuu + a.cuu_b * b, *cxu = a.cxu + a.cxu_b * b;
    Q4In cst;
    cst.cxx = cxx[e]; cst.cxxT = cxx[et]; cst.cxuc = cxu[r]; cst.cxur = cxu[c]; cst.cuu = cuu[0];      // (time-invariant cost; CTV reads the image instead)

    // per-lane LDS offsets (doubles) of the compute side
    const int ofx = 32 * blk + e, ofu = Q4L_IFU + 32 * blk + r, ocu = Q4L_ICU + 32 * blk;
    double *wV = lout + 32 * blk + e;
    double *wK = lout + Q4L_OK + ((c == 0) ? 32 * blk + r : 128 + lane);                        // K_i[0, r] | Vx_i[r] at + Q4L_KD
    double *wS = lout + Q4L_OKQ + ((q16 == 0) ? 16 * blk : 64 + (lane & 7) + 16 * (lane >> 3)); // k_i | Quu_i at + 8
    auto readin = [&](const double *in, int sidx, Q4In &o) __attribute__((always_inline)) {
        o.fx = in[ofx + 128 * (sidx >> 1) + 16 * (sidx & 1)];
        o.fu = in[ofu + 4 * sidx];
        o.cx = in[ofu + (Q4L_ICX - Q4L_IFU) + 4 * sidx];
        o.cu = in[ocu + sidx];
        o.u = LIMS ? in[ocu + 8 + sidx] : 0.0;
        if (CTV) {
            const int st = 128 * (sidx >> 1) + 16 * (sidx & 1);
            o.cxx = in[Q4L_IXX + 32 * blk + e + st]; o.cxxT = in[Q4L_IXX + 32 * blk + et + st];
            o.cxuc = in[Q4L_IXU + 32 * blk + r + 4 * sidx]; o.cxur = in[Q4L_IXU + 32 * blk + c + 4 * sidx];
            o.cuu = in[Q4L_IUU + 32 * blk + sidx];
        } else {
            o.cxx = cst.cxx; o.cxxT = cst.cxxT; o.cxuc = cst.cxuc; o.cxur = cst.cxur; o.cuu = cst.cuu;
        }
    };
    auto writeout = [&](int sidx, const Q4Out &o) __attribute__((always_inline)) {
        wV[128 * (sidx >> 1) + 16 * (sidx & 1)] = o.Vn;
        wK[4 * sidx] = o.Kc; wK[Q4L_KD + 4 * sidx] = o.vx;
        wS[sidx] = o.kk; wS[8 + sidx] = o.Quu;
    };

    Q4State s;
    s.kprev = 0.0; s.dV0 = 0.0; s.dV1 = 0.0; s.diverge = 0;
    double *cur = lin[0], *nxt = lin[1];
    dma(NC - 1, cur);
    if (NC > 1) dma(NC - 2, nxt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Q4In in;
    Q4Out prev;
    readin(cur, CH - 1, in);
    for (int ch = NC - 1; ch >= 0; --ch) {
#pragma unroll
        for (int sidx = CH - 1; sidx >= 0; --sidx) {
            Q4In nx;
            if (sidx > 0) readin(cur, sidx - 1, nx);
            else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next chunk was requested a whole chunk ago
                readin(nxt, CH - 1, nx);
            }
            Q4Out o;
            if (sidx == CH - 1 && ch == NC - 1) {
                // terminal step (backward_pass.jl:21-23 / :234-236), see back_pass_q4_kernel
                s.V = in.cxx; s.VT = in.cxxT; s.vxc = in.cx;
                o.Vn = s.V; o.Kc = 0.0; o.vx = in.cx; o.kk = 0.0; o.Quu = in.cuu;
            } else if (sidx == CH - 1)
                q4_step<LIMS, REG2, 0, Q4NoMid, GPS>(CH * ch + sidx, in, s, o, par);
            else {
                auto mid = [&]() __attribute__((always_inline)) { writeout(sidx + 1, prev); };
                q4_step<LIMS, REG2, 0, decltype(mid), GPS>(CH * ch + sidx, in, s, o, par, mid);
            }
            if (sidx == 0) { writeout(0, o); }
            prev = o;
            in = nx;
        }
        drain(ch);
        if (ch >= 2) dma(ch - 2, cur);
        double *t = cur; cur = nxt; nxt = t;
    }
    if (s.diverge && act) q4_zero_fill(a, b, q16, s.diverge);
    if (act && q16 == 0) { a.dV[2 * b] = s.dV0; a.dV[2 * b + 1] = s.dV1; a.diverge[b] = s.diverge; }
}


// ---- CHUNKS THROUGH THE LDS, second layout: TRAJECTORY-INTERLEAVED image, chunk length a template parameter, back_pass_gps' combination
// c̃• = c•/η + c•kl folded into the operand fetch.
// Why a second layout (profiles/r03_c3_pmc.txt): in the image of back_pass_q4l_kernel the four trajectories of a wave lie 32 doubles
// (= one whole row of LDS banks) apart, so the four lanes that read the same element of their trajectories always meet in one bank:
// SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 4.  Here the trajectory index is the FASTEST index of the 16-byte pieces:
//   matrix array (16 doubles / step / trajectory): piece p = 64 (step/2) + 32 (step&1) + 16 rh + 4 c + blk  holds elements (2rh, c), (2rh+1, c);
//     lane (r, blk, c) reads double 2p + (r&1): the 32 lanes of a half-wave touch 32 consecutive doubles — no conflict
//     (the transposed element, read for cxx' only: 2-way);
//   vector array (4 doubles):  piece p = 8 step + 4 rh + blk   — column form (element r) and row form (element c) conflict-free;
//   scalar array:              piece p = 4 (step/2) + blk      holds steps (2 step/2, +1).
// A direct-to-LDS load writes lane L's 16 bytes to piece L of its 64-piece window, so the per-lane SOURCE address decides the layout;
// every lane of every load / store belongs to trajectory blk = L & 3.  Loads of a kind are packed: with CH = 8 a load carries one
// vector array or four scalar arrays, with CH = 4 two vector arrays or eight scalar arrays.
// back_pass_gps (GPS): the image also takes cxx, cxu, cuu and the five KL terms as they are in memory (no prepass kernel, no c̃• round trip
// through HBM: 1.1 GB per pass at C5); a step forms c̃• = c• ieta + c•kl while it fetches (7 multiply-adds), the terminal step takes the raw cx,
// cxx (backward_pass.jl:281-283), Quui = 1/Quu leaves with k and Quu.  Its image is 9.3 KB per 4 steps: CH = 4 keeps a wave under the
// 40 KB that four waves per CU (B = 4 096) leave.
template <int CH, bool GPS>
struct Q4C {
    static constexpr int MAT = 64 * CH, VEC = 16 * CH, SCA = 4 * CH;                 // doubles per array and chunk (4 trajectories)
    // input image
    static constexpr int NMAT = GPS ? 3 : 1, NVEC = GPS ? 5 : 2, NSCA = GPS ? 5 : 2;  // fx [cxx kcxx] | fu cx [cxu kcx kcxu] | cu u [cuu kcu kcuu]
    static constexpr int I_MAT = 0, I_VEC = NMAT * MAT, I_SCA = I_VEC + NVEC * VEC;
    static constexpr int SCA_PAD = ((NSCA * SCA + 127) / 128) * 128;                  // whole 64-piece windows
    static constexpr int IN = I_SCA + SCA_PAD;
    static constexpr int LD_MAT = NMAT * MAT / 128, LD_VEC = (NVEC * VEC + 127) / 128, LD_SCA = SCA_PAD / 128;     // loads per chunk
    static constexpr int NLD = LD_MAT + LD_VEC + LD_SCA;
    // output image: Vxx | K Vx | k Quu [Quui], then the cells lanes without an output write to
    static constexpr int O_VEC = MAT, O_SCA = O_VEC + 2 * VEC, NOS = GPS ? 3 : 2;
    static constexpr int OSCA_PAD = ((NOS * SCA + 127) / 128) * 128;
    static constexpr int O_DUMP = O_SCA + OSCA_PAD, OUT = O_DUMP + 64 + 16 * CH + 16;
    static constexpr int ST_MAT = MAT / 128, ST_VEC = (2 * VEC + 127) / 128, ST_SCA = OSCA_PAD / 128, NST = ST_MAT + ST_VEC + ST_SCA;
    static constexpr int NBUF = GPS ? 3 : 2;                                          // input images (see the kernel)
};

template <bool LIMS, bool REG2, bool GPS, int CH, int EXP = 0>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_q4c_kernel(Q4Args a, const ddp_kl_cost_terms kl)
{
    typedef Q4C<CH, GPS> L;
    constexpr int n = 4;
    static_assert(CH == 4 || CH == 8, "chunk length");
    __shared__ __attribute__((aligned(16))) double lin[L::NBUF][L::IN];
    __shared__ __attribute__((aligned(16))) double lout[L::OUT];
    const int N = a.N, NC = N / CH;
    const int lane = threadIdx.x, r = lane >> 4, blk = (lane >> 2) & 3, c = lane & 3, q16 = 4 * r + c;
    long tb = (long)blockIdx.x * 4 + blk;
    const bool valid = tb < a.B;
    if (!valid) tb = a.B - 1;
    const int b = (int)tb;
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;
    // the lane's trajectory on the global side: every piece of every load / store belongs to trajectory lane & 3
    long tbd = (long)blockIdx.x * 4 + (lane & 3);
    const bool validd = tbd < a.B;
    if (!validd) tbd = a.B - 1;
    const size_t bd = (size_t)tbd;
    const bool actd = validd && !(a.active && a.active[bd] == 0);

    // ---- sources of the loads (chunk 0; a chunk further on is CH * {16, 4, 1} doubles further)
    const double *src[L::NLD];
    bool ldon[L::NLD];
    {
        // matrix arrays: fx, [cxx, kcxx]
        const double *mat[3] = {a.fx + a.fx_b * bd, GPS ? a.cxx + a.cxx_b * bd : a.fx, GPS ? kl.cxx + (size_t)16 * N * bd : a.fx};
#pragma unroll
        for (int i = 0; i < L::LD_MAT; ++i) {
            const int g = 64 * i + lane, arr = g / (32 * CH), p = g % (32 * CH);
            const int step = 2 * (p >> 6) + ((p >> 5) & 1), rh = (p >> 4) & 1, cc = (p >> 2) & 3;
            src[i] = mat[arr] + 16 * step + 2 * rh + 4 * cc;
            ldon[i] = true;
        }
        // vector arrays: fu, cx, [cxu, kcx, kcxu]
        const double *vec[5] = {a.fu + a.fu_b * bd, a.cx + (size_t)n * N * bd, GPS ? a.cxu + a.cxu_b * bd : a.cx, GPS ? kl.cx + (size_t)n * N * bd : a.cx,
                                GPS ? kl.cxu + (size_t)n * N * bd : a.cx};
#pragma unroll
        for (int i = 0; i < L::LD_VEC; ++i) {
            const int g = 64 * i + lane, arr = g / (8 * CH), p = g % (8 * CH);
            const int step = p >> 3, rh = (p >> 2) & 1;
            ldon[L::LD_MAT + i] = arr < L::NVEC;
            src[L::LD_MAT + i] = vec[arr < L::NVEC ? arr : 0] + 4 * step + 2 * rh;
        }
        // scalar arrays: cu, u, [cuu, kcu, kcuu]
        const double *sca[5] = {a.cu + (size_t)N * bd, (LIMS ? a.u : a.cu) + (size_t)N * bd, GPS ? a.cuu + a.cuu_b * bd : a.cu, GPS ? kl.cu + (size_t)N * bd : a.cu,
                                GPS ? kl.cuu + (size_t)N * bd : a.cu};
#pragma unroll
        for (int i = 0; i < L::LD_SCA; ++i) {
            const int g = 64 * i + lane, arr = g / (2 * CH), p = g % (2 * CH);
            ldon[L::LD_MAT + L::LD_VEC + i] = arr < L::NSCA;
            src[L::LD_MAT + L::LD_VEC + i] = sca[arr < L::NSCA ? arr : 0] + 2 * (p >> 2);
        }
    }
    auto dma_one = [&](int ch, double *in, int i) __attribute__((always_inline)) {
        const int per = i < L::LD_MAT ? 16 : (i < L::LD_MAT + L::LD_VEC ? 4 : 1);
        const int off = i < L::LD_MAT ? 128 * i : (i < L::LD_MAT + L::LD_VEC ? L::I_VEC + 128 * (i - L::LD_MAT) : L::I_SCA + 128 * (i - L::LD_MAT - L::LD_VEC));
        if (ldon[i]) __builtin_amdgcn_global_load_lds((glb_void *)(src[i] + (size_t)ch * (CH * per)), (lds_void *)(in + off), 16, 0, 0);
    };
    auto dma = [&](int ch, double *in) {
#pragma unroll
        for (int i = 0; i < L::NLD; ++i) dma_one(ch, in, i);
    };
    // ---- destinations of the stores: Vxx | K, Vx | k, Quu [, Quui]; a lane without an output writes its 16 bytes of the sink
    double *dst[L::NST];
    bool ston[L::NST];
    {
#pragma unroll
        for (int i = 0; i < L::ST_MAT; ++i) {
            const int p = 64 * i + lane, step = 2 * (p >> 6) + ((p >> 5) & 1), rh = (p >> 4) & 1, cc = (p >> 2) & 3;
            dst[i] = a.Vxx + (size_t)16 * N * bd + 16 * step + 2 * rh + 4 * cc;
            ston[i] = true;
        }
        double *vec[2] = {a.K + (size_t)n * N * bd, a.Vx + (size_t)n * N * bd};
#pragma unroll
        for (int i = 0; i < L::ST_VEC; ++i) {
            const int g = 64 * i + lane, arr = g / (8 * CH), p = g % (8 * CH);
            ston[L::ST_MAT + i] = arr < 2;
            dst[L::ST_MAT + i] = vec[arr < 2 ? arr : 0] + 4 * (p >> 3) + 2 * ((p >> 2) & 1);
        }
        double *sca[3] = {a.k + (size_t)N * bd, a.Quu + (size_t)N * bd, GPS ? a.Quui + (size_t)N * bd : a.k};
#pragma unroll
        for (int i = 0; i < L::ST_SCA; ++i) {
            const int g = 64 * i + lane, arr = g / (2 * CH), p = g % (2 * CH);
            ston[L::ST_MAT + L::ST_VEC + i] = arr < L::NOS;
            dst[L::ST_MAT + L::ST_VEC + i] = sca[arr < L::NOS ? arr : 0] + 2 * (p >> 2);
        }
    }
    auto drain = [&](int ch) {
        d2 v[L::NST];
#pragma unroll
        for (int i = 0; i < L::NST; ++i) {
            const int off = i < L::ST_MAT ? 128 * i : (i < L::ST_MAT + L::ST_VEC ? L::O_VEC + 128 * (i - L::ST_MAT) : L::O_SCA + 128 * (i - L::ST_MAT - L::ST_VEC));
            v[i] = *(const d2 *)(lout + off + 2 * lane);
        }
#pragma unroll
        for (int i = 0; i < L::NST; ++i) {
            const int per = i < L::ST_MAT ? 16 : (i < L::ST_MAT + L::ST_VEC ? 4 : 1);
            double *g = (actd && ston[i]) ? dst[i] + (size_t)ch * (CH * per) : a.sink + 2 * lane;
            *(d2 *)g = v[i];
        }
    };

    Q4Par par;
    par.lam = GPS ? 0.0 : a.lambda[b]; par.limlo = 0.0; par.limhi = 0.0; par.nolims = true;      // back_pass_gps: η is the only regularisation
    par.ieta = GPS ? 1.0 / a.eta[b] : 1.0;
    if (LIMS) { par.limlo = a.lims[0]; par.limhi = a.lims[1]; par.nolims = par.limlo > par.limhi; }
    Q4In cst;
    if (!GPS) {
        const double *cxx = a.cxx + a.cxx_b * b, *cuu = a.cuu + a.cuu_b * b, *cxu = a.cxu + a.cxu_b * b;
        const int e = r + 4 * c, et = c + 4 * r;
        cst.cxx = cxx[e]; cst.cxxT = cxx[et]; cst.cxuc = cxu[r]; cst.cxur = cxu[c]; cst.cuu = cuu[0];
    }
    // ---- per-lane offsets (doubles) of the compute side
    const int omat = 32 * (r >> 1) + 8 * c + 2 * blk + (r & 1), omatT = 32 * (c >> 1) + 8 * r + 2 * blk + (c & 1);       // element (r, c) / (c, r)
    const int ocol = 8 * (r >> 1) + 2 * blk + (r & 1), orow = 8 * (c >> 1) + 2 * blk + (c & 1);                            // vector element r / c
    const int osca = 2 * blk;
    auto smat = [](int sidx) { return 128 * (sidx >> 1) + 64 * (sidx & 1); };
    auto ssca = [](int sidx) { return 8 * (sidx >> 1) + (sidx & 1); };
    // outputs: V by every lane; K_i[0, r] by the lanes c == 0, Vx_i[r] by c == 1, k_i / Quu_i / Quui_i by lanes (0, 0) / (0, 1) / (0, 2) of a block
    double *wV = lout + omat;
    double *wK = lout + ((c == 0) ? L::O_VEC + ocol : (c == 1) ? L::O_VEC + L::VEC + ocol : L::O_DUMP + lane);
    const int wKs = (c <= 1) ? 16 : 0;                                                                              // doubles per step (dump cells: none)
    double *wS = lout + ((r == 0 && c < L::NOS) ? L::O_SCA + L::SCA * c + osca : L::O_DUMP + 64 + (lane & 15));
    const bool sreal = (r == 0 && c < L::NOS);
    auto readin = [&](const double *in, int sidx, Q4In &o, bool terminal) __attribute__((always_inline)) {
        o.fx = in[L::I_MAT + omat + smat(sidx)];
        o.fu = in[L::I_VEC + ocol + 16 * sidx];
        const double cxr = in[L::I_VEC + L::VEC + ocol + 16 * sidx];
        const double cur = in[L::I_SCA + osca + ssca(sidx)];
        o.u = LIMS ? in[L::I_SCA + L::SCA + osca + ssca(sidx)] : 0.0;
        if (GPS) {
            const double ie = par.ieta;
            const double xx = in[L::I_MAT + L::MAT + omat + smat(sidx)], xxT = in[L::I_MAT + L::MAT + omatT + smat(sidx)];
            const double kxx = in[L::I_MAT + 2 * L::MAT + omat + smat(sidx)], kxxT = in[L::I_MAT + 2 * L::MAT + omatT + smat(sidx)];
            const double xuc = in[L::I_VEC + 2 * L::VEC + ocol + 16 * sidx], xur = in[L::I_VEC + 2 * L::VEC + orow + 16 * sidx];
            const double kx = in[L::I_VEC + 3 * L::VEC + ocol + 16 * sidx];
            const double kxuc = in[L::I_VEC + 4 * L::VEC + ocol + 16 * sidx], kxur = in[L::I_VEC + 4 * L::VEC + orow + 16 * sidx];
            const double uu = in[L::I_SCA + 2 * L::SCA + osca + ssca(sidx)], ku = in[L::I_SCA + 3 * L::SCA + osca + ssca(sidx)],
                         kuu = in[L::I_SCA + 4 * L::SCA + osca + ssca(sidx)];
            // c̃• = c•/η + c•kl (backward_pass.jl:293-299); the terminal step keeps cx, cxx as they are (:281) and Quu_N = cuu/η + cuukl (:282)
            o.cx = terminal ? cxr : fma(cxr, ie, kx);
            o.cu = fma(cur, ie, ku);
            o.cxx = terminal ? xx : fma(xx, ie, kxx);
            o.cxxT = terminal ? xxT : fma(xxT, ie, kxxT);
            o.cxuc = fma(xuc, ie, kxuc); o.cxur = fma(xur, ie, kxur);
            o.cuu = fma(uu, ie, kuu);
        } else {
            o.cx = cxr; o.cu = cur;
            o.cxx = cst.cxx; o.cxxT = cst.cxxT; o.cxuc = cst.cxuc; o.cxur = cst.cxur; o.cuu = cst.cuu;
        }
    };
    auto writeout = [&](int sidx, const Q4Out &o) __attribute__((always_inline)) {
        wV[smat(sidx)] = o.Vn;
        wK[wKs * sidx] = (c == 0) ? o.Kc : o.vx;
        const double sv = (c == 0) ? o.kk : ((c == 1 || !GPS) ? o.Quu : ddp_rcp_nr(o.Quu));       // Quui_i = inv(Quu_i) (:283, :344)
        wS[sreal ? ssca(sidx) : 0] = sv;
    };

    Q4State s;
    s.kprev = 0.0; s.dV0 = 0.0; s.dV1 = 0.0; s.diverge = 0;
    // NBUF images: chunk ch lives in image (NC - 1 - ch) % NBUF and is requested NBUF - 1 chunks ahead.  With three images the wait
    // at the end of a chunk is COUNTED: vmcnt(NLD) leaves the loads of the chunk after next and the result stores of the last two
    // chunks in flight — at most NLD operations outstanding means at least the OLDER half of the 2 NLD loads has landed, whatever
    // the stores do (loads return in order among themselves; stores may pass them).  With vmcnt(0) and two images a chunk of four
    // steps (~4 us) waited for its predecessor's result stores to reach memory: 0.13 of 0.58 ms (DDP_Q4_EXP, C5).
    constexpr int NBUF = L::NBUF;
    dma(NC - 1, lin[0]);
    if (NC > 1) dma(NC - 2, lin[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (NBUF > 2 && NC > 2) dma(NC - 3, lin[2]);
    Q4In in;
    Q4Out prev;
    readin(lin[0], CH - 1, in, true);
    int ib = 0;                                                 // image of the current chunk
    for (int ch = NC - 1; ch >= 0; --ch) {
        const double *cur = lin[ib];
        const int inx = ib + 1 < NBUF ? ib + 1 : 0;
        const double *nxt = lin[inx];
#pragma unroll
        for (int sidx = CH - 1; sidx >= 0; --sidx) {
            Q4In nx;
            if (sidx > 0) readin(cur, sidx - 1, nx, false);
            else {
                // three images: the loads of the chunk after next (the newest NLD loads) may stay in flight.  (Tried and dropped, C5:
                // vmcnt(NLD + NST) with a content check of the last load so that the newest stores stay in flight too, and the loads
                // spread over the steps of a chunk instead of one burst at its end: 0.56 -> 0.65 ms.)
                if (NBUF > 2) { if (ch >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::NLD) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next chunk was requested a whole chunk ago
                readin(nxt, CH - 1, nx, false);
            }
            Q4Out o;
            if (sidx == CH - 1 && ch == NC - 1) {
                // terminal step (backward_pass.jl:21-23 / :234-236 / :281-283), see back_pass_q4_kernel
                s.V = in.cxx; s.VT = in.cxxT; s.vxc = in.cx;
                o.Vn = s.V; o.Kc = 0.0; o.vx = in.cx; o.kk = 0.0; o.Quu = in.cuu;
            } else if (sidx == CH - 1)
                q4_step<LIMS, REG2, 0, Q4NoMid, GPS>(CH * ch + sidx, in, s, o, par);
            else {
                auto mid = [&]() __attribute__((always_inline)) { writeout(sidx + 1, prev); };
                q4_step<LIMS, REG2, 0, decltype(mid), GPS>(CH * ch + sidx, in, s, o, par, mid);
            }
            if (sidx == 0) { writeout(0, o); }
            prev = o;
            in = nx;
        }
        // the image of this chunk is free: it takes the chunk NBUF further on
        if (ch >= NBUF && !(EXP & 2)) dma(ch - NBUF, lin[ib]);
        if (!(EXP & 1)) drain(ch);
        ib = inx;
    }
    if (s.diverge && act) {
        q4_zero_fill(a, b, q16, s.diverge);
        if (GPS) for (size_t t = q16; t < (size_t)s.diverge; t += 16) a.Quui[(size_t)N * b + t] = 0.0;       // :344 is not reached at and below the failing step
    }
    if (act && q16 == 0) { a.dV[2 * b] = s.dV0; a.dV[2 * b + 1] = s.dV1; a.diverge[b] = s.diverge; }
}


// ---- back_pass_gps for n = 4, m = 1 on the same kernel (backward_pass.jl:259-350).  c̃• = c•/η + c•kl for every step but the last,
// where the reference takes Vx[:,N] = cx[:,N], Vxx[:,:,N] = cxx[:,:,N] as they are and Quu[:,:,N] = cuu/η + cuukl (:281-283).
struct GpsCombine {
    int N, B;
    int cxx_t, cxu_t, cuu_t;                                    // element strides of the cost Hessians per time step / trajectory
    long cxx_b, cxu_b, cuu_b;
    const double *cx, *cu, *cxx, *cxu, *cuu, *kcx, *kcu, *kcxx, *kcxu, *kcuu, *eta;
    double *ox, *ou, *oxx, *oxu, *ouu;
};
__global__ __launch_bounds__(256) void gps_combine_kernel(GpsCombine a)
{
    const long NB = (long)a.N * a.B;
    const int which = blockIdx.y;                               // 0 cx (4), 1 cu (1), 2 cxx (16), 3 cxu (4), 4 cuu (1)
    const int len = which == 0 ? 4 : which == 2 ? 16 : which == 3 ? 4 : 1;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= NB * len) return;
    const long tb = g / len;
    const int e = (int)(g % len), t = (int)(tb % a.N);
    const long b = tb / a.N;
    const bool last = t == a.N - 1;
    const double ie = 1.0 / a.eta[b];
    switch (which) {
    case 0: a.ox[g] = last ? a.cx[g] : a.cx[g] * ie + a.kcx[g]; break;
    case 1: a.ou[g] = a.cu[g] * ie + a.kcu[g]; break;
    case 2: { const double c = a.cxx[a.cxx_b * b + (long)a.cxx_t * t + e]; a.oxx[g] = last ? c : c * ie + a.kcxx[g]; } break;
    case 3: a.oxu[g] = a.cxu[a.cxu_b * b + (long)a.cxu_t * t + e] * ie + a.kcxu[g]; break;          // m = 1: cxukl[1,4] and cxu[4,1] share their order
    default: a.ouu[g] = a.cuu[a.cuu_b * b + (long)a.cuu_t * t] * ie + a.kcuu[g]; break;
    }
}
// Quui[:,:,i] = inv(Quu[:,:,i]) (:283,344) for the steps the pass completed: every step without a failure, the steps after the failing
// one otherwise (it returns before :344 at the failing step; the caller's array is zero-filled)
__global__ __launch_bounds__(256) void gps_quui_kernel(int N, long NB, const double *__restrict__ Quu, const int32_t *__restrict__ diverge,
                                                       const int32_t *__restrict__ active, double *__restrict__ Quui)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= NB) return;
    const long b = g / N;
    if (active && active[b] == 0) return;
    const int t = (int)(g % N), dv = diverge[b];                // dv = 1-based failing step, 0: none
    Quui[g] = (dv == 0 || t > dv - 1) ? 1.0 / Quu[g] : 0.0;
}

}   // namespace

// returns 1 if this shape has no such kernel (caller falls back), 0 launched, <0 error
int ddp_launch_back_pass_q4(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                            const double *cxx, const double *cxu, const double *cuu, const double *fx,
                            const double *fu, const double *lambda, const double *lims, const double *u,
                            const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                            double *Vxx, double *dV, int32_t *diverge)
{
    if (!(d->n == 4 && d->m == 1)) return 1;
    Q4Args a;
    const long N = d->N;
    a.N = d->N; a.B = d->B; a.regType = d->regType;
    a.fx_t = d->fx_tv ? 16 : 0; a.fx_b = d->fx_batched ? 16 * (d->fx_tv ? N : 1) : 0;
    a.fu_t = d->fx_tv ? 4 : 0; a.fu_b = d->fx_batched ? 4 * (d->fx_tv ? N : 1) : 0;
    a.cxx_t = d->cost_tv ? 16 : 0; a.cxx_b = d->cost_batched ? 16 * (d->cost_tv ? N : 1) : 0;
    a.cxu_t = d->cost_tv ? 4 : 0; a.cxu_b = d->cost_batched ? 4 * (d->cost_tv ? N : 1) : 0;
    a.cuu_t = d->cost_tv ? 1 : 0; a.cuu_b = d->cost_batched ? (d->cost_tv ? N : 1) : 0;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.lims = lims;
    a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    a.sink = (double *)h->sink;
    const dim3 grid((unsigned)((d->B + 3) / 4)), block(DDP_WAVE);
    // DDP_Q4_EXP selects "removal experiment" kernels (no result stores / no operand loads: wrong results by design, for profiles/q4_exp.sh);
    // they exist only in a profiling build (-DDDP_PROFILE_BUILD, profiles/build_variant.sh) — the production library ignores the switch
#ifdef DDP_PROFILE_BUILD
    const char *ex = ddp_env(h, ENV_Q4_EXP);
    const int exp = ex ? atoi(ex) : 0;
#else
    const int exp = 0;
#endif
    const char *sg = ddp_env(h, ENV_Q4_SINGLE);                  // 1: force the one-step-at-a-time kernel (tests)
    const bool aligned16 = ((((uintptr_t)fx | (uintptr_t)fu | (uintptr_t)cx | (uintptr_t)cu | (uintptr_t)K | (uintptr_t)k | (uintptr_t)Quu |
                              (uintptr_t)Vx | (uintptr_t)Vxx | (uintptr_t)(d->has_lims ? u : cu)) & 15) == 0);
    const bool paired = !d->cost_tv && (d->N % 2 == 0) && d->N >= 4 * Q4_DP && aligned16 && h->sink != nullptr && !(sg && sg[0] == '1');
#define Q4P(L_, R_, E_) hipLaunchKernelGGL((back_pass_q4p_kernel<L_, R_, E_>), grid, block, 0, h->stream, a)
#define Q4S(L_, C_, R_) hipLaunchKernelGGL((back_pass_q4_kernel<L_, C_, R_>), grid, block, 0, h->stream, a)
    const bool reg2 = d->regType == 2;
    const char *lv = ddp_env(h, ENV_Q4_LDS);                     // 0 / 1: never / whenever possible the LDS-group kernel (A/B, tests)
    // the LDS-group kernel is the latency kernel: 24 KB of LDS per wave let 6 of them share a CU, and from two waves per SIMD on the
    // pair kernel hides its issue gaps behind the other wave (B = 8192: 0.66 ms against 0.90 ms; B = 6144: 0.61 against 0.53, B = 4096: 0.44 against 0.39)
    const bool few = lv ? (lv[0] == '1' || lv[0] == 'o') : d->B <= 6144;
    const bool c3exp = exp >= 11 && exp <= 13 && d->has_lims && d->regType == 2;      // DDP_Q4_EXP=11|12|13: removal experiments of the chunked kernel (C3 shape)
    const bool chunked = paired && d->fx_tv && (d->N % Q4L_CH == 0) && d->N >= 2 * Q4L_CH && (exp == 0 || c3exp) && few;
    a.eta = nullptr; a.Quui = nullptr;
    if (chunked && lv && lv[0] == 'o') {                        // DDP_Q4_LDS=o: the first chunked layout (A/B timing against q4c)
#define Q4L(L_, R_) hipLaunchKernelGGL((back_pass_q4l_kernel<L_, R_>), grid, block, 0, h->stream, a)
        if (d->has_lims && reg2) Q4L(true, true); else if (d->has_lims) Q4L(true, false); else if (reg2) Q4L(false, true); else Q4L(false, false);
#undef Q4L
    } else if (chunked) {
        const ddp_kl_cost_terms nokl = {};
#ifdef DDP_PROFILE_BUILD
        if (c3exp) {
            if (exp == 11) hipLaunchKernelGGL((back_pass_q4c_kernel<true, true, false, 8, 1>), grid, block, 0, h->stream, a, nokl);
            else if (exp == 12) hipLaunchKernelGGL((back_pass_q4c_kernel<true, true, false, 8, 2>), grid, block, 0, h->stream, a, nokl);
            else hipLaunchKernelGGL((back_pass_q4c_kernel<true, true, false, 8, 3>), grid, block, 0, h->stream, a, nokl);
            DDP_HIP(hipGetLastError());
            return 0;
        }
#endif
#define Q4C_(L_, R_) hipLaunchKernelGGL((back_pass_q4c_kernel<L_, R_, false, 8>), grid, block, 0, h->stream, a, nokl)
        if (d->has_lims && reg2) Q4C_(true, true); else if (d->has_lims) Q4C_(true, false); else if (reg2) Q4C_(false, true); else Q4C_(false, false);
#undef Q4C_
    } else if (paired) {
        if (d->has_lims && reg2) {
#ifdef DDP_PROFILE_BUILD
            switch (exp) {
            case 1: Q4P(true, true, 1); break; case 2: Q4P(true, true, 2); break; case 3: Q4P(true, true, 3); break;
            case 4: Q4P(true, true, 4); break; case 7: Q4P(true, true, 7); break; default: Q4P(true, true, 0);
            }
#else
            Q4P(true, true, 0);
#endif
        } else if (d->has_lims) Q4P(true, false, 0);
        else if (reg2) Q4P(false, true, 0);
        else Q4P(false, false, 0);
    } else {
        const int key = (d->has_lims ? 4 : 0) | (d->cost_tv ? 2 : 0) | (reg2 ? 1 : 0);
        switch (key) {
        case 0: Q4S(false, false, false); break; case 1: Q4S(false, false, true); break;
        case 2: Q4S(false, true, false); break;  case 3: Q4S(false, true, true); break;
        case 4: Q4S(true, false, false); break;  case 5: Q4S(true, false, true); break;
        case 6: Q4S(true, true, false); break;   case 7: Q4S(true, true, true); break;
        }
    }
#undef Q4S
#undef Q4P
    DDP_HIP(hipGetLastError());
    return 0;
}

// back_pass_gps for n = 4, m = 1 with one η per trajectory: prepass (c̃• = c•/η + c•kl into the handle's pad buffer), the matrix-core
// kernel above with the 1/η factors on the products with V, Quui = 1/Quu.  Returns 1 when the shape is not handled here.
int ddp_launch_back_pass_gps_q4(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                                const double *cxx, const double *cxu, const double *cuu, const double *fx,
                                const double *fu, const ddp_kl_cost_terms *kl, const double *lims, const double *u,
                                const int32_t *active, double *K, double *k, double *Quu, double *Quui, double *Vx,
                                double *Vxx, double *dV, int32_t *diverge)
{
    if (!(d->n == 4 && d->m == 1) || d->N < 2 || !d->fx_tv || !d->cost_tv || kl->eta_tv) return 1;
    const char *q4e = ddp_env(h, ENV_GPS_Q4);                      // 0: never (the lane-per-trajectory kernel instead; A/B timing, tests)
    if (q4e && q4e[0] == '0') return 1;
    const long N = d->N, B = d->B, NB = N * B;
    {   // chunks of four steps through the LDS with c̃• = c•/η + c•kl formed in the kernel (back_pass_q4c_kernel): ONE launch, no c̃• buffers.
        // DDP_GPS_Q4L=0: the prepass + one-step kernel below (A/B timing, cross-check in the tests); =o: prepass + the first chunked kernel
        const char *le0 = ddp_env(h, ENV_GPS_Q4L);
        const bool al16f = ((((uintptr_t)fx | (uintptr_t)fu | (uintptr_t)cx | (uintptr_t)cu | (uintptr_t)cxx | (uintptr_t)cxu | (uintptr_t)cuu | (uintptr_t)kl->cx |
                              (uintptr_t)kl->cu | (uintptr_t)kl->cxx | (uintptr_t)kl->cxu | (uintptr_t)kl->cuu | (uintptr_t)K | (uintptr_t)k | (uintptr_t)Quu |
                              (uintptr_t)Quui | (uintptr_t)Vx | (uintptr_t)Vxx | (uintptr_t)(d->has_lims ? u : fx)) & 15) == 0);
        if (!le0 && d->N % 4 == 0 && d->N >= 8 && al16f && h->sink != nullptr && d->B <= 6144 && Quui) {
            Q4Args a;
            a.N = d->N; a.B = d->B; a.regType = 1;
            a.fx_t = 16; a.fx_b = d->fx_batched ? 16 * N : 0;
            a.fu_t = 4; a.fu_b = d->fx_batched ? 4 * N : 0;
            a.cxx_t = 16; a.cxx_b = d->cost_batched ? 16 * N : 0; a.cxu_t = 4; a.cxu_b = d->cost_batched ? 4 * N : 0; a.cuu_t = 1; a.cuu_b = d->cost_batched ? N : 0;
            a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = nullptr; a.lims = lims;
            a.u = u; a.active = active;
            a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
            a.eta = kl->eta; a.Quui = Quui; a.sink = (double *)h->sink;
            const dim3 grid((unsigned)((d->B + 3) / 4)), block(DDP_WAVE);
#ifdef DDP_PROFILE_BUILD
            const char *ex = ddp_env(h, ENV_Q4_EXP);            // removal experiments (profiles/q4_exp.sh): 1 no result stores, 2 no operand loads after the first two chunks
            const int exp = ex ? atoi(ex) : 0;
            if (d->has_lims && exp == 1) hipLaunchKernelGGL((back_pass_q4c_kernel<true, false, true, 4, 1>), grid, block, 0, h->stream, a, *kl);
            else if (d->has_lims && exp == 2) hipLaunchKernelGGL((back_pass_q4c_kernel<true, false, true, 4, 2>), grid, block, 0, h->stream, a, *kl);
            else if (d->has_lims && exp == 3) hipLaunchKernelGGL((back_pass_q4c_kernel<true, false, true, 4, 3>), grid, block, 0, h->stream, a, *kl);
            else
#endif
            if (d->has_lims) hipLaunchKernelGGL((back_pass_q4c_kernel<true, false, true, 4>), grid, block, 0, h->stream, a, *kl);
            else hipLaunchKernelGGL((back_pass_q4c_kernel<false, false, true, 4>), grid, block, 0, h->stream, a, *kl);
            DDP_HIP(hipGetLastError());
            return 0;
        }
    }
    auto al = [](size_t b_) { return (b_ + 255) & ~(size_t)255; };
    const size_t s4 = al((size_t)4 * NB * 8), s1 = al((size_t)NB * 8), s16 = al((size_t)16 * NB * 8), bytes = 2 * s4 + 2 * s1 + s16;
    if (bytes > h->pad_bytes) {
        DDP_HIP(hipStreamSynchronize(h->stream));
        if (h->pad) DDP_HIP(hipFree(h->pad));
        h->pad = nullptr; h->pad_bytes = 0;
        DDP_HIP(hipMalloc(&h->pad, bytes));
        h->pad_bytes = bytes;
    }
    char *q = (char *)h->pad;
    auto tk = [&](size_t b_) { double *r_ = (double *)q; q += b_; return r_; };
    GpsCombine c;
    c.N = d->N; c.B = d->B;
    c.cxx_t = 16; c.cxu_t = 4; c.cuu_t = 1;
    c.cxx_b = d->cost_batched ? 16 * N : 0; c.cxu_b = d->cost_batched ? 4 * N : 0; c.cuu_b = d->cost_batched ? N : 0;
    c.cx = cx; c.cu = cu; c.cxx = cxx; c.cxu = cxu; c.cuu = cuu;
    c.kcx = kl->cx; c.kcu = kl->cu; c.kcxx = kl->cxx; c.kcxu = kl->cxu; c.kcuu = kl->cuu; c.eta = kl->eta;
    c.ox = tk(s4); c.ou = tk(s1); c.oxx = tk(s16); c.oxu = tk(s4); c.ouu = tk(s1);
    hipLaunchKernelGGL(gps_combine_kernel, dim3((unsigned)((16 * NB + 255) / 256), 5), dim3(256), 0, h->stream, c);
    Q4Args a;
    a.N = d->N; a.B = d->B; a.regType = 1;
    a.fx_t = 16; a.fx_b = d->fx_batched ? 16 * N : 0;
    a.fu_t = 4; a.fu_b = d->fx_batched ? 4 * N : 0;
    a.cxx_t = 16; a.cxx_b = 16 * N; a.cxu_t = 4; a.cxu_b = 4 * N; a.cuu_t = 1; a.cuu_b = N;
    a.cx = c.ox; a.cu = c.ou; a.cxx = c.oxx; a.cxu = c.oxu; a.cuu = c.ouu; a.fx = fx; a.fu = fu; a.lambda = nullptr; a.lims = lims;
    a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    a.eta = kl->eta; a.Quui = nullptr; a.sink = (double *)h->sink;
    const dim3 grid((unsigned)((d->B + 3) / 4)), block(DDP_WAVE);
    // chunks of eight steps through the LDS (the q4l scheme, here with the time-varying c̃xx, c̃xu, c̃uu in the image): 13 direct-to-LDS loads
    // and 7 stores per 8 steps instead of 10 + 2 vector-memory instructions per step.  DDP_GPS_Q4L=0: the one-step kernel (A/B, tests)
    const char *le = ddp_env(h, ENV_GPS_Q4L);
    const bool al16 = ((((uintptr_t)fx | (uintptr_t)fu | (uintptr_t)K | (uintptr_t)k | (uintptr_t)Quu | (uintptr_t)Vx | (uintptr_t)Vxx |
                         (uintptr_t)(d->has_lims ? u : fx)) & 15) == 0);
    const bool chunked = !(le && le[0] == '0') && d->N % Q4L_CH == 0 && d->N >= 2 * Q4L_CH && al16 && h->sink != nullptr && d->B <= 6144;
    if (chunked && d->has_lims) hipLaunchKernelGGL((back_pass_q4l_kernel<true, false, true, true>), grid, block, 0, h->stream, a);
    else if (chunked) hipLaunchKernelGGL((back_pass_q4l_kernel<false, false, true, true>), grid, block, 0, h->stream, a);
    else if (d->has_lims) hipLaunchKernelGGL((back_pass_q4_kernel<true, true, false, 0, true>), grid, block, 0, h->stream, a);
    else hipLaunchKernelGGL((back_pass_q4_kernel<false, true, false, 0, true>), grid, block, 0, h->stream, a);
    hipLaunchKernelGGL(gps_quui_kernel, dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, h->stream, d->N, NB, (const double *)Quu,
                       (const int32_t *)diverge, active, Quui);
    DDP_HIP(hipGetLastError());
    return 0;
}

#ifdef Q4_COUNT_SLOW
extern "C" int ddp_q4_slow_count(unsigned long long *out2)      /* out2: 7 entries */
{
    if (hipMemcpyFromSymbol(&out2[0], HIP_SYMBOL(q4_slow_count), 8) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&out2[1], HIP_SYMBOL(q4_step_count), 8) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&out2[2], HIP_SYMBOL(q4_slow_why), 40) != hipSuccess) return -1;
    return 0;
}
#endif
