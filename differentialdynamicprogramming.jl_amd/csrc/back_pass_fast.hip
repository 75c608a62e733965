// back_pass_fast.hip — LDS-lean, branch-free backward pass for the unconstrained (Cholesky) path with m = 2
// and even n <= 10 (the BASELINE headline shape n=10, m=2).  Same arithmetic as back_pass.hip /
// src/backward_pass.jl:162-252 + :28-42,:64-76; what changes is WHERE operands live and HOW lanes are
// assigned, because the first kernel was LDS-bandwidth / issue bound (profiles/r01: 105 ds_read2_b64 and
// ~920 instructions per time step).
//
// Lane map (n=10: 6 groups of 10 lanes + 4 spare lanes; lane = 10*g + r):
//   every lane keeps TWO columns (2g, 2g+1) of the stacked Jacobian F = [fx fu] (n x p, p = n+2) in registers
//   (loaded once for LTI dynamics, refreshed from an LDS copy per step for LTV) and uses them in both products:
//     P1  lane (g, r)      : W[r, 2g], W[r, 2g+1]      = Vxx[r,:]·F[:,2g..2g+1]          reads row r of Vxx
//     P2  lane (g, r>=2g)  : G[2g, r], G[2g+1, r]      = F[:,2g..2g+1]'·W[:,r]           Qxx upper-triangular
//                                                         2x1 blocks, kept in registers until P4
//         lane (5, r)      : column r of Qux   (the two u-rows),  spare lanes 61,62: columns of Quu
//         lane (g, 0), g>=1, spare 60 (g=0), spare 63 (g=5): [Qx;Qu] pair 2g,2g+1 — Vx is stored as an
//                            extra column of W so the same dot-product loop serves all roles
//   P3  the 2x2 Quu/QuuF and Qu travel by v_readlane (SGPR broadcast): no LDS round trip, no hand-off;
//       every lane factorises QuuF redundantly (reciprocal square roots by v_rsq_f64 + 2 Newton steps),
//       k and dV are wave-uniform, lane (5, r) solves column r of K in registers
//   P4  lane (g, r>=2g): Vxx[2g..2g+1, r] (+ mirrors);  [Qx;Qu]-pair lanes: Vx[2g..2g+1]
// Three LDS hand-offs per step (W | K,T,Qux | Vxx,Vx), all operand reads are 16-byte ds_read_b128, no
// divergent branches in the steady state except the predicated global stores.
// One wavefront per trajectory (see back_pass.hip for why).
#include "ddp_internal.h"

namespace {

struct BPFArgs {
    int N, B;
    int fx_batched, cost_batched, regType;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
};

constexpr int FTC = 8;   // time steps per cx/cu prefetch chunk

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double bcast(double v, int srclane)
{   // wave-uniform broadcast through SGPRs (2 x v_readlane_b32); srclane is a compile-time constant
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_nr(double x)
{   // 1/sqrt(x): hardware estimate (~2^-26) + two Newton steps -> ~1 ulp; x > 0 is checked by the caller
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}

template <int NS>
struct FLds {       // LDS carve-up in doubles (all offsets even => 16-byte aligned)
    static constexpr int n = NS, p = NS + 2;
    static constexpr int Vs = 0,
                         Ws = Vs + n * n,             // n x (p+1): column p is Vx
                         Ks = Ws + n * (p + 1), Ts = Ks + 2 * n, Xs = Ts + 2 * n,
                         Fs = Xs + 2 * n,             // n x p   stacked Jacobian of the current step
                         Cst = Fs + n * p,            // [cxx n*n | cxu' 2*n (pairs) | cuu 4]
                         cbuf = Cst + n * n + 2 * n + 4,
                         Dmy = cbuf + 2 * FTC * p,    // lane-private dump slots for role-less lanes
                         total = Dmy + 2 * DDP_WAVE;
};

template <int NS, bool FXTV, bool CTV>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_fast_kernel(BPFArgs a)
{
    constexpr int n = NS, m = 2, p = n + m, NG = n / 2 + 1;       // NG groups of n lanes
    static_assert(NS % 2 == 0 && NS >= 2 && NG * NS + 4 <= DDP_WAVE, "even n <= 10 only");
    constexpr int SB = NG * n;                                      // first spare lane
    constexpr int LQ0 = SB, LU0 = SB + 1, LU1 = SB + 2, LQ5 = SB + 3;
    using L = FLds<NS>;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (a.active && a.active[b] == 0) return;
    const int N = a.N;

    __shared__ __attribute__((aligned(16))) double lds[L::total];
    double *Vs = lds + L::Vs, *Ws = lds + L::Ws, *vs = lds + L::Ws + n * p, *Ks = lds + L::Ks, *Ts = lds + L::Ts,
           *Xs = lds + L::Xs, *Fs = lds + L::Fs, *Cst = lds + L::Cst, *cbuf = lds + L::cbuf, *Dmy = lds + L::Dmy;

    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = 4;
    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    const double *fx = a.fx + (a.fx_batched ? nn * (FXTV ? N : 1) * b : 0);
    const double *fu = a.fu + (a.fx_batched ? nm * (FXTV ? N : 1) * b : 0);
    const double *cxx = a.cxx + (a.cost_batched ? nn * (CTV ? N : 1) * b : 0);
    const double *cxu = a.cxu + (a.cost_batched ? nm * (CTV ? N : 1) * b : 0);
    const double *cuu = a.cuu + (a.cost_batched ? mm * (CTV ? N : 1) * b : 0);
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + mm * N * b,
           *Vxg = a.Vx + (size_t)n * N * b, *Vxxg = a.Vxx + nn * N * b;
    const double lam = a.lambda[b];
    const bool reg2 = a.regType == 2;

    // ---- lane roles (all loop-invariant)
    const bool main_lane = lane < SB;
    const int g = main_lane ? lane / n : (lane == LQ0 ? 0 : NG - 1);   // column pair (2g, 2g+1) held in F0/F1
    const int r = main_lane ? lane % n : 0;
    const bool isX = main_lane && g < NG - 1 && r >= 2 * g;            // Qxx / Vxx block rows 2g,2g+1, column r
    const bool isXb = isX && r != 2 * g;                               // second row is not the mirror of another lane's entry
    const bool isUK = main_lane && g == NG - 1;                        // column r of Qux -> column r of K
    const bool isUQ = lane == LU0 || lane == LU1;                      // columns of Quu
    const bool isQ = (main_lane && g >= 1 && g < NG - 1 && r == 0) || lane == LQ0 || lane == LQ5;   // [Qx;Qu] pair g
    const bool isQx = isQ && g < NG - 1;                               // pairs that are Vx entries
    const int wcol = isX ? r : (isUK ? r : (isUQ ? n + (lane - LU0) : (isQ ? p : 0)));
    const int i0 = 2 * g, jj = isX ? r : (i0 < n ? i0 : 0);            // P4 indices (clamped for role-less lanes)
    const int i0c = (i0 < n) ? i0 : 0;
    // LDS destinations (role-less lanes dump into their private slot)
    const int dmy = L::Dmy + 2 * lane;
    const int wdst0 = main_lane ? L::Ws + r + n * (2 * g) : dmy, wdst1 = main_lane ? L::Ws + r + n * (2 * g + 1) : dmy + 1;
    const int kdst = isUK ? 2 * r : -1;
    const int vd0 = isX ? L::Vs + i0 + n * r : (isQx ? L::Ws + n * p + i0 : dmy);
    const int vd1 = isX ? L::Vs + r + n * i0 : (isQx ? L::Ws + n * p + i0 + 1 : dmy + 1);
    const int vd2 = isXb ? L::Vs + i0 + 1 + n * r : dmy, vd3 = isXb ? L::Vs + r + n * (i0 + 1) : dmy + 1;
    // constant-term source inside Cst: X: cxx[2g..2g+1, r]; UK: cxu[r, 0..1]; UQ: cuu[0..1, bb]
    const int coff = isX ? (i0 + n * r) : (isUK ? n * n + 2 * r : (isUQ ? n * n + 2 * n + 2 * (lane - LU0) : 0));
    // [Qx;Qu] pair: offset of (c_2g, c_2g+1) inside one time step of the chunk buffer
    const int qbase = (i0 < n) ? i0 : FTC * n, qstride = (i0 < n) ? n : m;

    // ---- terminal step (backward_pass.jl:234-236 / :197-199), staging of loop-invariant operands
    const size_t tl = (size_t)(N - 1), t2 = (N >= 2) ? (size_t)(N - 2) : 0;
    for (int e = lane; e < n * n; e += DDP_WAVE) {
        const double v = cxx[(CTV ? nn * tl : 0) + e];
        Vs[e] = v;
        Vxxg[nn * tl + e] = v;
        Cst[e] = CTV ? cxx[nn * t2 + e] : v;
    }
    for (int e = lane; e < n; e += DDP_WAVE) {
        const double v = cx[(size_t)n * tl + e];
        vs[e] = v;
        Vxg[(size_t)n * tl + e] = v;
    }
    if (lane < 4) {
        Quug[mm * tl + lane] = cuu[(CTV ? mm * tl : 0) + lane];
        Cst[n * n + 2 * n + lane] = cuu[(CTV ? mm * t2 : 0) + lane];
    }
    for (int e = lane; e < 2 * n; e += DDP_WAVE) {
        Kg[nm * tl + e] = 0.0;
        Cst[n * n + 2 * (e % n) + e / n] = cxu[(CTV ? nm * t2 : 0) + e];        // cxu'[a, j] pairs
    }
    if (lane < m) kg[(size_t)m * tl + lane] = 0.0;
    if (N < 2) {
        if (lane == 0) { a.dV[2 * b] = 0.0; a.dV[2 * b + 1] = 0.0; a.diverge[b] = 0; }
        return;
    }
    {
        const size_t off = FXTV ? t2 : 0;
        for (int e = lane; e < n * n; e += DDP_WAVE) Fs[e] = fx[nn * off + e];
        for (int e = lane; e < n * m; e += DDP_WAVE) Fs[n * n + e] = fu[nm * off + e];
    }
    // cx|cu chunks: chunk c = time steps [c*FTC, c*FTC+FTC), layout [FTC*n cx | FTC*m cu]
    auto chunk_elem = [&](int c, int e) -> double {
        const int t0 = c * FTC;
        if (e < FTC * n) return (t0 + e / n < N) ? cx[(size_t)t0 * n + e] : 0.0;
        e -= FTC * n;
        return (t0 + e / m < N) ? cu[(size_t)t0 * m + e] : 0.0;
    };
    constexpr int CE = FTC * p, RC = (CE + DDP_WAVE - 1) / DDP_WAVE;
    double pfc[RC];
    {
        const int c0 = (N - 2) / FTC;
#pragma unroll
        for (int q = 0; q < RC; ++q) {
            const int e = lane + DDP_WAVE * q;
            if (e < CE) cbuf[(c0 & 1) * CE + e] = chunk_elem(c0, e);
            pfc[q] = (c0 > 0 && e < CE) ? chunk_elem(c0 - 1, e) : 0.0;
        }
    }
    wave_sync();

    // ---- register-resident operands
    double F0[n], F1[n];
    auto load_F = [&]() {
#pragma unroll
        for (int l = 0; l < n; l += 2) {
            const d2 f0 = *(const d2 *)(Fs + (2 * g) * n + l), f1 = *(const d2 *)(Fs + (2 * g + 1) * n + l);
            F0[l] = f0.x; F0[l + 1] = f0.y; F1[l] = f1.x; F1[l + 1] = f1.y;
        }
    };
    load_F();
    d2 cterm = *(const d2 *)(Cst + coff);                              // X: cxx pair, U: cxu'/cuu pair
    // regularisation added to the u-rows: regType 1 -> λ on the diagonal of Quu; regType 2 -> λ·F_u'F (:245-247)
    d2 radd = d2{0.0, 0.0};
    auto load_radd = [&]() {
        if (reg2) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int l = 0; l < n; ++l) { const double f = Fs[(wcol < p ? wcol : 0) * n + l]; s0 += F0[l] * f; s1 += F1[l] * f; }
            radd = (isUK || isUQ) ? d2{lam * s0, lam * s1} : d2{0.0, 0.0};
        } else {
            radd = d2{lane == LU0 ? lam : 0.0, lane == LU1 ? lam : 0.0};
        }
    };
    load_radd();

    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    double pfF[FXTV ? 2 : 1], pfxx[CTV ? 2 : 1], pfxu = 0.0;
    for (int i = N - 2; i >= 0; --i) {
        const int cc = i / FTC, so = i - cc * FTC;
        const double *cb = cbuf + (cc & 1) * CE;
        // ---- next step's time-varying operands (one step ahead, land during this step)
        if (FXTV && i > 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = lane + DDP_WAVE * q;
                if (e < n * p) pfF[q] = (e < n * n) ? fx[nn * (i - 1) + e] : fu[nm * (i - 1) + (e - n * n)];
            }
        }
        if (CTV && i > 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = lane + DDP_WAVE * q;
                if (e < n * n) pfxx[q] = cxx[nn * (i - 1) + e];
            }
            if (lane < 2 * n) pfxu = cxu[nm * (i - 1) + lane];
            else if (lane < 2 * n + 4) pfxu = cuu[mm * (i - 1) + (lane - 2 * n)];
        }

        // ================= P1: W[r, 2g..2g+1] = Vxx[r,:]·F[:, 2g..2g+1] ==============================
        {
            double w0 = 0.0, w1 = 0.0;
#pragma unroll
            for (int l = 0; l < n; l += 2) {
                const d2 v = *(const d2 *)(Vs + r * n + l);            // row r == column r (symmetric)
                w0 += v.x * F0[l]; w1 += v.x * F1[l];
                w0 += v.y * F0[l + 1]; w1 += v.y * F1[l + 1];
            }
            lds[wdst0] = w0;
            lds[wdst1] = w1;
        }
        wave_sync();

        // ================= P2: (g0,g1) = F[:,2g..2g+1]'·W[:,wcol]  (W column p is Vx) =================
        double g0 = 0.0, g1 = 0.0;
        {
            const double *wc = Ws + wcol * n;
#pragma unroll
            for (int l = 0; l < n; l += 2) {
                const d2 w = *(const d2 *)(wc + l);
                g0 += F0[l] * w.x; g1 += F1[l] * w.x;
                g0 += F0[l + 1] * w.y; g1 += F1[l + 1] * w.y;
            }
        }
        const d2 cq = *(const d2 *)(cb + qbase + so * qstride);         // (cx|cu)[2g..2g+1, i] for the [Qx;Qu] lanes
        const double gu0 = g0 + (isQ ? cq.x : cterm.x), gu1 = g1 + (isQ ? cq.y : cterm.y);   // Qxx | Qux,Quu | Qx,Qu
        const double gr0 = gu0 + radd.x, gr1 = gu1 + radd.y;            // regularised u-rows (Qux_reg, QuuF)

        // ================= P3: gains — SGPR broadcast of the 2x2 system, no LDS =========================
        const double Quu00 = bcast(gu0, LU0), Quu10 = bcast(gu1, LU0), Quu01 = bcast(gu0, LU1), Quu11 = bcast(gu1, LU1);
        const double F00 = bcast(gr0, LU0), F01 = bcast(gr0, LU1), F11 = bcast(gr1, LU1);   // upper triangle of QuuF
        const double Qu0 = bcast(gu0, LQ5), Qu1 = bcast(gu1, LQ5);
        // cholesky(Hermitian(QuuF)) (:35): R = [r00 r01; 0 r11], kept as reciprocals
        const double ir00 = rsqrt_nr(F00);
        const double r01 = F01 * ir00;
        const double d11 = F11 - r01 * r01;
        if (!(F00 > 0.0) || !(d11 > 0.0)) {                             // wave-uniform: diverge = i (:37-38)
            diverge = i + 1;
            if (lane == LU0) { Quug[mm * i] = gu0; Quug[mm * i + 1] = gu1; }
            if (lane == LU1) { Quug[mm * i + 2] = gu0; Quug[mm * i + 3] = gu1; }
            break;
        }
        const double ir11 = rsqrt_nr(d11);
        // x = -(R'R)\b :  y0 = b0/r00, y1 = (b1 - r01 y0)/r11, x1 = y1/r11, x0 = (y0 - r01 x1)/r00
        auto solve = [&](double b0, double b1, double &x0, double &x1) {
            const double y0 = b0 * ir00;
            const double y1 = (b1 - r01 * y0) * ir11;
            x1 = y1 * ir11;
            x0 = (y0 - r01 * x1) * ir00;
            x0 = -x0; x1 = -x1;
        };
        double k0, k1;
        solve(Qu0, Qu1, k0, k1);                                        // k_i (wave-uniform) (:41)
        const double Quuk0 = Quu00 * k0 + Quu01 * k1, Quuk1 = Quu10 * k0 + Quu11 * k1;     // (:64)
        dV0 += k0 * Qu0 + k1 * Qu1;                                     // (:68)
        dV1 += 0.5 * (k0 * Quuk0 + k1 * Quuk1);
        {                                                               // K_i[:, r] (:42), T = Quu·K + Qux
            double K0, K1;
            solve(gr0, gr1, K0, K1);
            const double T0 = gu0 + Quu00 * K0 + Quu01 * K1, T1 = gu1 + Quu10 * K0 + Quu11 * K1;
            double *kd = (kdst >= 0) ? Ks + kdst : lds + dmy, *td = (kdst >= 0) ? Ts + kdst : lds + dmy,
                   *xd = (kdst >= 0) ? Xs + kdst : lds + dmy;
            *(d2 *)kd = d2{K0, K1};
            *(d2 *)td = d2{T0, T1};
            *(d2 *)xd = d2{gu0, gu1};
            if (isUK) *(d2 *)(Kg + nm * i + 2 * r) = d2{K0, K1};        // (:76)
        }
        if (isUQ) *(d2 *)(Quug + mm * i + 2 * (lane - LU0)) = d2{gu0, gu1};
        if (lane == LQ5) *(d2 *)(kg + (size_t)m * i) = d2{k0, k1};      // (:75)
        wave_sync();

        // ================= P4: value update (:69-72) ====================================================
        {
            const d2 Ka = *(const d2 *)(Ks + 2 * i0c), Kb = *(const d2 *)(Ks + 2 * i0c + 2), Kj = *(const d2 *)(Ks + 2 * jj);
            const d2 Ta = *(const d2 *)(Ts + 2 * i0c), Tb = *(const d2 *)(Ts + 2 * i0c + 2), Tj = *(const d2 *)(Ts + 2 * jj);
            const d2 Xa = *(const d2 *)(Xs + 2 * i0c), Xb = *(const d2 *)(Xs + 2 * i0c + 2), Xj = *(const d2 *)(Xs + 2 * jj);
            // M_ij = Qxx_ij + K_i·T_j + Qux_i·K_j ;  M_ji = Qxx_ij + K_j·T_i + Qux_j·K_i   (Qxx_ji := Qxx_ij)
            const double ma = gu0 + (Ka.x * Tj.x + Ka.y * Tj.y) + (Xa.x * Kj.x + Xa.y * Kj.y);
            const double mat = gu0 + (Kj.x * Ta.x + Kj.y * Ta.y) + (Xj.x * Ka.x + Xj.y * Ka.y);
            const double mb = gu1 + (Kb.x * Tj.x + Kb.y * Tj.y) + (Xb.x * Kj.x + Xb.y * Kj.y);
            const double mbt = gu1 + (Kj.x * Tb.x + Kj.y * Tb.y) + (Xj.x * Kb.x + Xj.y * Kb.y);
            const double va = (ma + mat) * 0.5, vb = (mb + mbt) * 0.5;  // (:71-72)
            // Vx_i (:69) for the [Qx;Qu] pair lanes
            const double h0 = Quuk0 + Qu0, h1 = Quuk1 + Qu1;
            const double vx0 = gu0 + (Ka.x * h0 + Ka.y * h1) + (Xa.x * k0 + Xa.y * k1);
            const double vx1 = gu1 + (Kb.x * h0 + Kb.y * h1) + (Xb.x * k0 + Xb.y * k1);
            const double o0 = isQ ? vx0 : va, o1 = isQ ? vx1 : vb, o01 = isQ ? vx1 : va;
            lds[vd0] = o0; lds[vd1] = o01; lds[vd2] = o1; lds[vd3] = o1;
            if (isX) {
                Vxxg[nn * i + i0 + n * r] = va; Vxxg[nn * i + r + n * i0] = va;
                if (isXb) { Vxxg[nn * i + i0 + 1 + n * r] = vb; Vxxg[nn * i + r + n * (i0 + 1)] = vb; }
            }
            if (isQx) *(d2 *)(Vxg + (size_t)n * i + i0) = d2{vx0, vx1};
        }
        // ---- hand the prefetched operands of step i-1 over (their readers are the loads after the hand-off)
        if (FXTV && i > 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = lane + DDP_WAVE * q;
                if (e < n * p) Fs[e] = pfF[q];
            }
        }
        if (CTV && i > 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = lane + DDP_WAVE * q;
                if (e < n * n) Cst[e] = pfxx[q];
            }
            if (lane < 2 * n) Cst[n * n + 2 * (lane % n) + lane / n] = pfxu;
            else if (lane < 2 * n + 4) Cst[n * n + 2 * n + (lane - 2 * n)] = pfxu;
        }
        if (i > 0 && so == 0) {                                         // leaving chunk cc: publish cc-1, fetch cc-2
            double *nb = cbuf + ((cc - 1) & 1) * CE;
#pragma unroll
            for (int q = 0; q < RC; ++q) {
                const int e = lane + DDP_WAVE * q;
                if (e < CE) nb[e] = pfc[q];
            }
            if (cc >= 2) {
#pragma unroll
                for (int q = 0; q < RC; ++q) {
                    const int e = lane + DDP_WAVE * q;
                    if (e < CE) pfc[q] = chunk_elem(cc - 2, e);
                }
            }
        }
        wave_sync();
        if (FXTV && i > 0) { load_F(); load_radd(); }
        if (CTV && i > 0) cterm = *(const d2 *)(Cst + coff);
    }
    if (diverge) {   // outputs earlier in time than the failing step are zero (backward_pass.jl:226-229)
        const size_t ie = (size_t)diverge;          // = i + 1
        for (size_t e = lane; e < nm * ie; e += DDP_WAVE) Kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)m * ie; e += DDP_WAVE) kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)n * ie; e += DDP_WAVE) Vxg[e] = 0.0;
        for (size_t e = lane; e < nn * ie; e += DDP_WAVE) Vxxg[e] = 0.0;
        for (size_t e = lane; e < mm * (ie - 1); e += DDP_WAVE) Quug[e] = 0.0;
    }
    if (lane == 0) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
}

template <int NS>
int launch_fast(ddp_handle h, const ddp_bp_desc *d, const BPFArgs &a)
{
    const dim3 grid(d->B), block(DDP_WAVE);
    const int key = (d->fx_tv ? 2 : 0) | (d->cost_tv ? 1 : 0);
    switch (key) {
    case 0: hipLaunchKernelGGL((back_pass_fast_kernel<NS, false, false>), grid, block, 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((back_pass_fast_kernel<NS, false, true>), grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((back_pass_fast_kernel<NS, true, false>), grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL((back_pass_fast_kernel<NS, true, true>), grid, block, 0, h->stream, a); break;
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// returns 1 if this shape has no fast kernel (caller falls back to the general kernel), 0 launched, <0 error
int ddp_launch_back_pass_fast(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                              const double *cxx, const double *cxu, const double *cuu, const double *fx,
                              const double *fu, const double *lambda, const int32_t *active, double *K,
                              double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge)
{
    if (d->has_lims || d->m != 2 || d->n != 10) return 1;
    BPFArgs a;
    a.N = d->N; a.B = d->B; a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched; a.regType = d->regType;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    return launch_fast<10>(h, d, a);
}
