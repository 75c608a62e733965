// back_pass_mfma_kernel.h — backward pass for the BASELINE config-4 shape n = 64, m = 8 with every product of the
// Riccati step on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).  Same arithmetic and failure semantics as
// back_pass.hip / back_pass_big.hip (src/backward_pass.jl:179-215 + :28-79).
//
// One 256-thread work-group (4 waves = the 4 SIMDs of a CU) per trajectory.  LDS:
//     Vs [64 x 64]  Vxx_{i+1}, symmetric, leading dimension LDV = 65  (A operand of W = Vxx·F)
//     Fs [64 x 80]  F = [fx fu 0], state index fastest, leading dimension LDK = 66 — the coalesced global
//                   layout goes to LDS untransposed (conflict-free writes) and LDK ≡ 2 (mod 32) makes the
//                   k-major operand reads (B of W = Vxx·F, A of G = F'W) conflict-free too
//     WT [64 x 64]  W' (column index of W fastest, LD = 80)
//     PT            per-wave partial sums of the u/Vx columns of G = F'W (9 of 16 tile columns, compact)
// The dependent chain of a step is  Vxx_{i+1} -> W_u = Vxx·fu -> Quu, Qux, Qu -> gains -> Vxx_i;  the phases are cut so
// that the serial gain computation (one wave, VALU) runs beside the bulk of W = Vxx·fx (three waves, matrix cores):
//   phase 1  wave w: rows 16w.. of W for the column tiles {u | Vx} and 0 (32 MFMAs, A shared); the u tile never
//            leaves the registers: with the k index permuted to the accumulator layout it is the B operand of the
//            wave's k-slice of ALL FIVE row tiles of G[:, u | Vx] (20 MFMAs); column 72 of W is Vx_{i+1}, so
//            G[:,72] = F'Vx.  Vxx_{i+1} streams to global and the next Jacobian is requested under these MFMAs.
//   phase 2  wave 0 sums the four partial tiles and computes the gains (every lane factorises QuuF, lane c solves
//            column c of K; or the boxQP)  |  wave w = 1..3: column tile w of W for all four row tiles (64 MFMAs, B shared)
//   phase 3  the 10 upper tiles of Vxx_i = cxx + fx'W + ½(K'Y + Y'K): 16 + 4 MFMAs per tile (the rank-16 update
//            [K;Y]'·½[Y;K] rides on the same accumulator).  The tiles of a wave SHARE an operand (waves 0-2 a row of the
//            upper triangle, wave 3 the column-3 left-overs): with all four waves in a product phase it is the LDS that
//            saturates at two operand reads per MFMA (measured 115 cycles per MFMA against 72 of the pipe), not the
//            matrix cores.  The upper triangle is mirrored (exactly symmetric Vxx); Vx_i on wave 3
//   then the next Jacobian goes to Fs (two more barriers: everybody is done reading Fs / before it is read again).
// With control limits (EARLY3) the box-QP makes wave 0's phase 2 the longest part of the step, so the cut is different there: the
// tiles of Vxx_i belong to the waves 1..3 by W column, each goes from its W column straight to cxx + fx'W of its tiles (phase 3a, no
// barrier in between), phase 3b behind the gains barrier is only the rank-16 update, and Fs is free one barrier earlier.
// Measured (profiles/microbench/mfma_f64_bench.hip): 30 ns per MFMA per wave with ONE wave per SIMD (66-70 TF/s of the
// 78.6 TF/s peak) — unlike the fp64 VALU, the matrix pipe does not need several waves to fill.
// Included by back_pass_mfma.hip (no control limits; built with -amdgpu-mfma-vgpr-form) and back_pass_mfma_lims.hip
// (boxQP variant; its register pressure crashes the compiler's AGPR rewrite pass under that flag, so it is a
// separate translation unit built without it).
#pragma once
#include "ddp_internal.h"
#include "boxqp_dev.h"
#include <type_traits>
#include "boxqp_rows.h"

struct BPMArgs {
    int N, B;
    int fx_tv, fx_batched, cost_tv, cost_batched, regType, has_lims;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
};

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256, n = 64, m = 8, p = 72, PP = 80, LD = 80, LDK = 66, LDV = 65, KS = 10, PTS = 36;   // PTS: lanes (l4, l15 < 9) of a partial tile register
constexpr int oVs = 0, oFs = oVs + n * LDV, oWT = oFs + PP * LDK, ovs = oWT + n * LD, oQs = ovs + n, oXs = oQs + PP,
              oXrs = oXs + m * n, oQuus = oXrs + m * n, oRadd = oQuus + m * m, oKs = oRadd + m * m, oYs = oKs + KS * n,
              oks = oYs + KS * n, oQuuks = oks + m, oPT = oQuuks + m, oFlag = oPT + 4 * 5 * 4 * PTS, oTot = oFlag + 2;

#ifdef DDP_MFPROF     // per-phase cycle counts (s_memtime) of block 0, printed per wave: profiling builds only
#define MFP_DECL long long mfp_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mfp_t = __builtin_amdgcn_s_memtime()
#define MFP(k) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_amdgcn_s_memtime(); mfp_[k] += t_ - mfp_t; mfp_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define MFP_PRINT do { if (b == 0 && lane == 0) printf("MFPROF wave %d steps %d: ph1 %lld bar %lld ph2 %lld bar %lld ph3 %lld bar %lld stF %lld bar %lld | ph1: gemm %lld fused %lld; p3: red %lld chol+k %lld Ksolve %lld qp_iter_sum %lld qp_setup %lld qp_solve %lld\n", wv, N - 1, \
    mfp_[0] / (N - 1), mfp_[1] / (N - 1), mfp_[2] / (N - 1), mfp_[3] / (N - 1), mfp_[4] / (N - 1), mfp_[5] / (N - 1), mfp_[6] / (N - 1), mfp_[7] / (N - 1), \
    mfp_[8] / (N - 1), mfp_[9] / (N - 1), mfp_[12] / (N - 1), mfp_[13] / (N - 1), mfp_[14] / (N - 1), mfp_[15], mfp_[10] / (N - 1), mfp_[11] / (N - 1)); } while (0)
#else
#define MFP_DECL
#define MFP(k)
#define MFP_PRINT
#endif

#define MF_SB __builtin_amdgcn_sched_barrier(0)
#ifndef C4_EXP
#define C4_EXP 0          // timing experiments (profiles/ab_c4.sh): bit 0 no operand fetch in the phase-2 products, bit 1 no global traffic there
#endif
__device__ __forceinline__ d4 mf(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }

// NTL MFMA chains that share one operand (A if SHA, else B): 1 + NTL LDS reads per k-step instead of 2·NTL — with all four
// waves in a product phase the LDS, not the matrix pipe, is what saturates at two reads per MFMA.
template <int NK, int PF, int NTL, int SS, int SO, bool SHA>
__device__ __forceinline__ void mfma_chain_shared(const double *sp, const double *const *op, d4 *c)
{
    double r[PF + 1][1 + NTL];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        r[j][0] = sp[SS * j];
#pragma unroll
        for (int u = 0; u < NTL; ++u) r[j][1 + u] = op[u][SO * j];
    }
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        // each product is followed by one piece of the fetch PF k-steps ahead: it issues in the product's shadow
        const int q = kk % (PF + 1), j = (kk + PF) % (PF + 1);
        const bool pf = kk + PF < NK;
#pragma unroll
        for (int u = 0; u < NTL; ++u) {
            MF_SB;
            c[u] = SHA ? mf(r[q][0], r[q][1 + u], c[u]) : mf(r[q][1 + u], r[q][0], c[u]);
            MF_SB;
            if (pf && !(C4_EXP & 4)) {
                if (u == 0) r[j][0] = sp[SS * (kk + PF)];
                r[j][1 + u] = op[u][SO * (kk + PF)];
            }
        }
    }
    MF_SB;
}

template <bool LIMS, bool CTV>
__global__ __launch_bounds__(NT) void back_pass_mfma_kernel(BPMArgs a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.active && a.active[b] == 0) return;
    const int N = a.N, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Vs = lds + oVs, *Fs = lds + oFs, *WT = lds + oWT, *vs = lds + ovs, *Qs = lds + oQs, *Xs = lds + oXs, *Xadd = lds + oXrs,
           *Quus = lds + oQuus, *Radd = lds + oRadd, *Ks = lds + oKs, *Ys = lds + oYs, *ks = lds + oks, *Quuks = lds + oQuuks,
           *PT = lds + oPT, *flag = lds + oFlag;

    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const bool FXTV = a.fx_tv;
    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    const double *ug = LIMS ? a.u + (size_t)m * N * b : nullptr;
    const double *fx = a.fx + (a.fx_batched ? nn * (FXTV ? N : 1) * b : 0);
    const double *fu = a.fu + (a.fx_batched ? nm * (FXTV ? N : 1) * b : 0);
    const double *cxx = a.cxx + (a.cost_batched ? nn * (CTV ? N : 1) * b : 0);
    const double *cxu = a.cxu + (a.cost_batched ? nm * (CTV ? N : 1) * b : 0);
    const double *cuu = a.cuu + (a.cost_batched ? mm * (CTV ? N : 1) * b : 0);
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + mm * N * b,
           *Vxg = a.Vx + (size_t)n * N * b, *Vxxg = a.Vxx + nn * N * b;
    const double lam = a.lambda[b];
    const int regType = a.regType;
    bool nolims = true;
    double limlo[m], limhi[m];
    if (LIMS) {
        nolims = a.lims[0] > a.lims[m];                             // backward_pass.jl:31
#pragma unroll
        for (int q = 0; q < m; ++q) { limlo[q] = a.lims[q]; limhi[q] = a.lims[q + m]; }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35
    // the QP runs with one coordinate per lane (boxqp_rows.h): this lane's bounds, and u[:, i] one step ahead of its use (an HBM
    // round trip on the serial part of every step otherwise)
    const int qcoord = (l15 < m) ? l15 : 0;
    double qlo = 0.0, qhi = 0.0, qu_next = 0.0;
    if (LIMS) {
        qlo = nolims ? -HUGE_VAL : a.lims[qcoord]; qhi = nolims ? HUGE_VAL : a.lims[qcoord + m];
        if (N >= 2) qu_next = ug[(size_t)m * (N - 2) + qcoord];
    }

    // ---- terminal step (backward_pass.jl:197-199); Vxx_{N-1} itself is streamed out by the first step below
    for (int e = tid; e < n * n; e += NT) Vs[(e & 63) + LDV * (e >> 6)] = cxx[(CTV ? nn * (N - 1) : 0) + e];
    if (tid < n) { const double v = cx[(size_t)n * (N - 1) + tid]; vs[tid] = v; Vxg[(size_t)n * (N - 1) + tid] = v; }
    if (tid < m * m) Quug[mm * (N - 1) + tid] = cuu[(CTV ? mm * (N - 1) : 0) + tid];
    for (int e = tid; e < m * n; e += NT) Kg[nm * (N - 1) + e] = 0.0;
    if (tid < m) { kg[(size_t)m * (N - 1) + tid] = 0.0; ks[tid] = 0.0; }
    for (int e = tid; e < (PP - p) * LDK; e += NT) Fs[p * LDK + e] = 0.0;      // zero padding columns 72..79
    if (N < 2) {
        for (int e = tid; e < n * n; e += NT) Vxxg[nn * (N - 1) + e] = cxx[(CTV ? nn * (N - 1) : 0) + e];
        if (tid == 0) { a.dV[2 * b] = 0.0; a.dV[2 * b + 1] = 0.0; a.diverge[b] = 0; }
        return;
    }
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int RF = n * p / NT / 2;                // 9 pairs (16-byte loads) of F per thread: 8 of fx, 1 of fu
    auto load_F1 = [&](int i, int q) -> d2 {          // pair q: elements e, e+1 with e = 2*(tid + NT*q), consecutive state rows
        const int e = 2 * (tid + NT * q);
        return *(const d2 *)((q < RF - 1) ? fx + nn * (FXTV ? i : 0) + e : fu + nm * (FXTV ? i : 0) + (e - n * n));
    };
    auto store_F = [&](const d2 (&r)[RF]) {           // F[k, c] (k fastest in memory) -> Fs[k + LDK*c]; LDK even: 16-byte aligned
#pragma unroll
        for (int q = 0; q < RF; ++q) {
            const int e = 2 * (tid + NT * q);
            *(d2 *)(Fs + (e & 63) + LDK * (e >> 6)) = r[q];
        }
    };
    d2 pfF[RF];
#pragma unroll
    for (int q = 0; q < RF; ++q) pfF[q] = load_F1(N - 2, q);
    store_F(pfF);
    __syncthreads();

    // upper-triangle tile t = 0..9 of the 4 x 4 Qxx tiling, column-major: (0,0) (0,1) (1,1) (0,2) ...
    // phase-3 tiles of a wave share an operand: wave w < 3 takes row w of the upper triangle ((0,0) (0,1) (0,2) | (1,1) (1,2) (1,3) |
    // (2,2) (2,3)), A = F[:, 16w..] shared; wave 3 the two left-over tiles of column 3, (0,3) and (3,3), B = W[:, 48..] shared
    // EARLY3 (control limits: the box-QP makes wave 0's serial part ~5 us, longer than the products of the other waves): the tiles belong
    // to the waves 1..3 BY COLUMN — wave c computes column tile c of W = Vxx·F in phase 2 and goes straight on, without a barrier, to
    // the K-independent part cxx + fx'W of the tiles (ti, c), ti <= c, whose B operand is that column (shared by the chains of the
    // wave); wave 1 also takes (0,0), whose W column comes from phase 1.  All of that runs beside the QP; only the rank-16 update
    // ½(K'Y + Y'K) waits for the gains.  Without limits the gains are shorter than phase 2 and the four-wave split above is faster
    // (measured: 7.7 ms against 9.1 ms with the column split, which leaves wave 0 idle).
    constexpr bool EARLY3 = LIMS;
    auto tile_w = [](int w, int u, int &ti, int &tj) -> bool {
        if (EARLY3) {
            if (w == 1) { ti = (u == 2) ? 1 : 0; tj = (u == 0) ? 0 : 1; return u < 3; }          // (0,0) | (0,1) (1,1)
            if (w == 2) { ti = min(u, 2); tj = 2; return u < 3; }                              // (0,2) (1,2) (2,2)
            if (w == 3) { ti = min(u, 3); tj = 3; return u < 4; }                              // (0,3) (1,3) (2,3) (3,3)
            ti = 0; tj = 0; return false;
        }
        if (u >= 3) { ti = 0; tj = 0; return false; }
        if (w < 3) { ti = w; tj = min(w + u, 3); return u < (w == 2 ? 2 : 3); }
        ti = (u == 0) ? 0 : 3; tj = 3; return u < 2;
    };
    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    double cxxr[4][4], cxur[m], preq[2];              // cost-Hessian operands of this thread (reloaded per step only if CTV)
    auto load_cost = [&](int i) {
        const double *cxxi = cxx + (CTV ? nn * i : 0), *cxui = cxu + (CTV ? nm * i : 0), *cuui = cuu + (CTV ? mm * i : 0);
#pragma unroll
        for (int u = 0; u < (EARLY3 ? 4 : 3); ++u) { // C operands of this wave's Vxx tiles (phase 3)
            int ti, tj;
            tile_w(wv, u, ti, tj);
            const double *cp = cxxi + 16 * ti + l4 + n * (16 * tj + l15);
            cxxr[u][0] = cp[0]; cxxr[u][1] = cp[4]; cxxr[u][2] = cp[8]; cxxr[u][3] = cp[12];
        }
        if (wv == 0) {                              // wave 0 adds cxu (column `lane`), cuu to the reduced partial tiles (phase 2)
#pragma unroll
            for (int q = 0; q < m; ++q) cxur[q] = cxui[lane + n * q];
#pragma unroll
            for (int r = 0; r < 2; ++r) preq[r] = (l15 < m) ? cuui[l4 + 4 * r + m * l15] : 0.0;
        }
    };
    if (!CTV) load_cost(0);
    MFP_DECL;
    for (int i = N - 2; i >= 0; --i) {
        const bool ldF = FXTV && i > 0;                 // next step's Jacobian is fetched under this step's first product
        if (CTV) load_cost(i);
        double gxc = 0.0, gu[2] = {0.0, 0.0};           // gradient entries: cx[lane], cu (wave 0)

        // ================= phase 1: W[16w.., {u|Vx, 0}] = Vxx·F; partial G[:, u|Vx] from the registers ==========
        {
            d4 accU = d4{0.0, 0.0, 0.0, 0.0}, acc0 = d4{0.0, 0.0, 0.0, 0.0};
            const double *ap = Vs + 16 * wv + l15 + LDV * l4;          // A[i][k] = Vxx[16w+i, k]
            const double *bp = Fs + l4 + LDK * l15;                   // B[k][j] = F[k, 16c+j]
            constexpr int PF1 = 2;                                    // operands fetched two k-steps ahead
            double av[PF1 + 1], bUv[PF1 + 1], b0v[PF1 + 1];
#pragma unroll
            for (int j = 0; j < PF1; ++j) { av[j] = ap[LDV * 4 * j]; bUv[j] = bp[LDK * 64 + 4 * j]; b0v[j] = bp[4 * j]; }
            // Global traffic rides in the MFMA shadow, at most one instruction per product: the address unit takes ~16 cycles per
            // wave instruction and all four waves share it (a burst at the top of the step costs ~2k cycles).  Here: columns
            // 0..31 of Vxx_{i+1} (:72 of step i+1; the rest goes out in phase 2), wave 0's share of the next Jacobian.
            const double *vout = Vs + lane + LDV * wv;
            double *gout = Vxxg + nn * (i + 1) + tid;
            double vprev = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int c = kk % (PF1 + 1), j = (kk + PF1) % (PF1 + 1);
                const bool pf = kk + PF1 < 16;
                MF_SB;
                accU = mf(av[c], bUv[c], accU);
                MF_SB;
                if (pf && !(C4_EXP & 8)) { av[j] = ap[LDV * 4 * (kk + PF1)]; bUv[j] = bp[LDK * 64 + 4 * (kk + PF1)]; }
                if (C4_EXP & 16) { }
                else if (kk & 1) gout[NT * (kk >> 1)] = vprev;
                else vprev = vout[LDV * 4 * (kk >> 1)];
                MF_SB;
                acc0 = mf(av[c], b0v[c], acc0);
                MF_SB;
                if (pf && !(C4_EXP & 8)) b0v[j] = bp[4 * (kk + PF1)];
                if (wv == 0 && !(C4_EXP & 16)) {
                    if (kk == 0) gxc = cx[(size_t)n * i + lane];
                    else if (kk < 3) { if (l15 == m) gu[kk - 1] = cu[(size_t)m * i + l4 + 4 * (kk - 1)]; }
                    else if (kk < 3 + RF) { if (ldF) pfF[kk - 3] = load_F1(i - 1, kk - 3); }
                }
            }
            MF_SB;
            MFP(8);
            {                                                         // D[row = l4 + 4r][col = l15] -> WT[col + LD*row]
                double *wp = WT + l15 + LD * (16 * wv + l4);
                wp[0] = acc0.x; wp[LD * 4] = acc0.y; wp[LD * 8] = acc0.z; wp[LD * 12] = acc0.w;
            }
            // The u|Vx tile of W as B operand: k-step r uses k = 16w + l4 + 4r, which is the accumulator register r of this lane
            double bu[4] = {accU.x, accU.y, accU.z, accU.w};
            if (l15 == m) { const double *vp = vs + 16 * wv + l4; bu[0] = vp[0]; bu[1] = vp[4]; bu[2] = vp[8]; bu[3] = vp[12]; }   // column 72 := Vx_{i+1}
            d4 pg[5];
#pragma unroll
            for (int ti = 0; ti < 5; ++ti) pg[ti] = d4{0.0, 0.0, 0.0, 0.0};
            const double *fp = Fs + 16 * wv + l4 + LDK * l15;         // A[i][k] = F[16w + l4 + 4r, 16ti + i]
            double fa[5], fb[5];
#pragma unroll
            for (int ti = 0; ti < 5; ++ti) fa[ti] = fp[LDK * 16 * ti];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) {
                    MF_SB;
                    pg[ti] = mf(fa[ti], bu[r], pg[ti]);
                    MF_SB;
                    if (r < 3 && !(C4_EXP & 8)) fb[ti] = fp[LDK * 16 * ti + 4 * (r + 1)];
                }
                MF_SB;
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) fa[ti] = fb[ti];
            }
            if (l15 <= m) {                                           // columns u | Vx of the five partial tiles, compact
                double *pp = PT + (wv * 5 * 4) * PTS + l4 * 9 + l15;
#pragma unroll
                for (int ti = 0; ti < 5; ++ti) {
                    pp[(ti * 4 + 0) * PTS] = pg[ti].x; pp[(ti * 4 + 1) * PTS] = pg[ti].y;
                    if (ti < 4) { pp[(ti * 4 + 2) * PTS] = pg[ti].z; pp[(ti * 4 + 3) * PTS] = pg[ti].w; }     // tile 4: rows 0..7 only
                }
            }
            MFP(9);
        }
        if (regType == 2) {     // (:205-207): QuuF = Quu + λ·fu'fu, Qux_reg = Qux + λ·fu'fx — the λ terms only, added in phase 2
            for (int e = tid; e < m * n + m * m; e += NT) {
                const bool isx = e < m * n;
                const int q = isx ? (e & 7) : ((e - m * n) & 7), j = isx ? (e >> 3) : n + ((e - m * n) >> 3);
                double s = 0.0;
#pragma unroll 8
                for (int kq = 0; kq < n; ++kq) s += Fs[kq + LDK * (n + q)] * Fs[kq + LDK * j];
                if (isx) Xadd[e] = lam * s;
                else Radd[e - m * n] = lam * s;
            }
        }
        MFP(0);
        __syncthreads();
        MFP(1);

        d4 acc3[4];                                                  // EARLY3: this wave's tiles of Vxx_i (waves 1..3)
        if (wv == 0) {
            // ================= phase 2, wave 0: reduce the partial tiles, gains (backward_pass.jl:30-68) =========
            // This wave is the serial part of the step: a chain of dependent vector instructions that shares its SIMD with the matrix
            // phases of the other work-groups of the CU, and a 64-cycle fp64 MFMA of theirs between two of its instructions is what
            // stretched the 8x8 QP from 4.7 us (alone on a CU, profiles/microbench/boxqp_rows_bench.hip) to ~30 us.  Top priority
            // lets it issue whenever it can.
            __builtin_amdgcn_s_setprio(3);
            double x2[m], xr[m];
            {
                // row `lane` of G[:, u|Vx]: tile lane/16, register (lane%16)/4, tile row lane%4 -> offset 9*lane
                const double *pp = PT + 9 * lane;
                double s[m + 1];
#pragma unroll
                for (int q = 0; q <= m; ++q) s[q] = ((pp[q] + pp[q + 20 * PTS]) + pp[q + 40 * PTS]) + pp[q + 60 * PTS];
#pragma unroll
                for (int q = 0; q < m; ++q) {
                    x2[q] = s[q] + cxur[q];                                       // Qux[q, lane]  (:208)
                    xr[q] = (regType == 2) ? x2[q] + Xadd[q + m * lane] : x2[q];
                }
                Qs[lane] = s[m] + gxc;                                            // Qx[lane]  (:203)
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {                                         // tile 4, rows l4 + 4r < 8
                const int aq = l4 + 4 * r;
                if (l15 <= m) {
                    const double *pp = PT + (16 + r) * PTS + l4 * 9 + l15;
                    const double s = ((pp[0] + pp[20 * PTS]) + pp[40 * PTS]) + pp[60 * PTS];
                    if (l15 < m) Quus[aq + m * l15] = s + preq[r];                // (:209)
                    else Qs[n + aq] = s + gu[r];                                  // (:204)
                }
            }
            wave_sync();
            double H[m * m], R[m * m], kk[m];
            unsigned clamped = 0u;
            MFP(12);
            int fail;
            double ri[m];
            constexpr bool use_ri = !LIMS;                           // division-free factor on the unconstrained path; a kernel compiled for
                                                                     // limits takes the QP also for `lims[1,1] > lims[1,2]` (bounds at ±Inf)
            double qu[m];
#pragma unroll
            for (int q = 0; q < m; ++q) qu[q] = Qs[n + q];
            if constexpr (use_ri) {
#pragma unroll
                for (int e = 0; e < m * m; ++e) H[e] = Quus[e];
                if (regType == 2) {
#pragma unroll
                    for (int e = 0; e < m * m; ++e) H[e] += Radd[e];
                } else {
#pragma unroll
                    for (int q = 0; q < m; ++q) H[q + m * q] += lam;
                }
                fail = ddp_chol_rinv<m>(H, R, ri);                   // cholesky(Hermitian(QuuF))  (:35)
#pragma unroll
                for (int q = 0; q < m; ++q) kk[q] = qu[q];
                ddp_rsolve_neg<m>(R, ri, kk);                        // k_i = -(R\Qu)  (:41)
            } else {
                // boxQP with one coordinate per lane (boxqp_rows.h): lane l15 < m of every 16-lane row holds row and column l15 of QuuF
                bqr::Rows<m> qr;
                const bool qin = l15 < m;
                const int qi = qcoord;
#pragma unroll
                for (int j = 0; j < m; ++j) {
                    double hr = Quus[qi + m * j], hc = Quus[j + m * qi];
                    if (regType == 2) { hr += Radd[qi + m * j]; hc += Radd[j + m * qi]; }
                    else if (j == qi) { hr += lam; hc += lam; }
                    qr.Hrow[j] = qin ? hr : 0.0; qr.Hcol[j] = qin ? hc : 0.0;
                }
                const double uq = qu_next;                           // u[qi, i], requested a step ago
                qu_next = ug[(size_t)m * (i > 0 ? i - 1 : 0) + qi];
                const double gq = qin ? Qs[n + qi] : 0.0, loq = qin ? qlo - uq : 0.0, upq = qin ? qhi - uq : 0.0;   // (:45-46)
                const double x0q = qin ? ks[qi] : 0.0;               // warm start k[:, min(i+1, N-1)] (:49)
                double xq;
                int iters;
                MFP(10);
                const int result = bqr::boxqp_rows<m>(qr, gq, loq, upq, x0q, qpo, l15, xq, clamped, iters);
                MFP(11);
                fail = (result < 1);                                 // (:53)
#ifdef DDP_MFPROF
                mfp_[15] += iters;                                   // profiling builds: QP iterations
#endif
                // every lane solves a column of K with the factor, and needs all of k
                asm volatile("s_nop 1" : "+v"(xq));
                bqr::sfor<0, m>([&](auto qc) { constexpr int q = decltype(qc)::value; kk[q] = bqr::bcast<q>(xq); });
                bqr::sfor<0, m>([&](auto cc) {
                    constexpr int c2 = decltype(cc)::value;
                    ri[c2] = qr.ri[c2];
                    bqr::sfor<0, m>([&](auto kc) {
                        constexpr int k2 = decltype(kc)::value;
                        if constexpr (k2 < c2) R[k2 + m * c2] = bqr::bcast<c2>(qr.Rcol[k2]); else R[k2 + m * c2] = 0.0;     // R[k2][c2] lives in lane c2
                    });
                });
            }
            MFP(13);
            if (lane == 0) flag[0] = fail ? 1.0 : 0.0;
            if (fail) {
                Quug[mm * i + lane] = Quus[lane];
            } else {
                // every LDS read first, every write last: the compiler cannot prove the K/Y writes do not alias Quus
                double col[m];
#pragma unroll
                for (int q = 0; q < m; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : xr[q];
                // Unconstrained regType 1: (Quu + λI)·K = -Qux and (Quu + λI)·k = -Qu hold to the backward error of the solve,
                // so Quu·K and Quu·k need no product with Quu (the other cases take it from LDS again: H is dead, R holds the factor)
                const bool by_residual = use_ri && regType != 2;
                if (!by_residual) {
#pragma unroll
                    for (int e = 0; e < m * m; ++e) H[e] = Quus[e];
                }
                const double quu_l = Quus[lane];
                if (use_ri) ddp_rsolve_neg<m>(R, ri, col);           // K_i column `lane`
                else {
                    chol_solve_ri<m>(m, R, ri, col);
#pragma unroll
                    for (int q = 0; q < m; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : -col[q];
                }
                MFP(14);
                double y[m], quuk[m], kQu = 0.0, kQuuk = 0.0;
#pragma unroll
                for (int q = 0; q < m; ++q) {                        // Y = Quu·K + 2·Qux;  Quu·k, dV (:64-68)
                    if (by_residual) {
                        y[q] = fma(-lam, col[q], x2[q]);                 // Quu·K = -Qux - λK
                        quuk[q] = -fma(lam, kk[q], qu[q]);               // Quu·k = -Qu - λk
                    } else {
                        double t = 2.0 * x2[q], t2 = 0.0;
#pragma unroll
                        for (int q2 = 0; q2 < m; ++q2) {
                            const double hq = H[(q < q2 ? q : q2) + m * (q < q2 ? q2 : q)];  // upper triangle, like the factorisation
                            t += hq * col[q2]; t2 += hq * kk[q2];
                        }
                        y[q] = t; quuk[q] = t2;
                    }
                }
#pragma unroll
                for (int q = 0; q < m; ++q) { kQu += kk[q] * qu[q]; kQuuk += kk[q] * quuk[q]; }
                dV0 += kQu; dV1 += 0.5 * kQuuk;                      // (every lane; lane 0 reports)
#pragma unroll
                for (int q = 0; q < m; ++q) {
                    Xs[q + m * lane] = x2[q];
                    Ks[q + KS * lane] = col[q];
                    Ys[q + KS * lane] = y[q];
                    Kg[nm * i + q + (size_t)m * lane] = col[q];      // (:76)
                }
                Quug[mm * i + lane] = quu_l;
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < m; ++q) { Quuks[q] = quuk[q]; ks[q] = kk[q]; kg[(size_t)m * i + q] = kk[q]; }
                }
            }
            __builtin_amdgcn_s_setprio(0);
        } else {
            // ================= phase 2, wave c = 1..3: column tile c of W = Vxx·F for the four row tiles ==========
            d4 acc[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc[rb] = d4{0.0, 0.0, 0.0, 0.0};
            const double *ap = Vs + l15 + LDV * l4;                   // A[i][k] = Vxx[16rb+i, k]
            const double *bp = Fs + l4 + LDK * (16 * wv + l15);       // B[k][j] = F[k, 16c+j]
            constexpr int PF = 2;                                     // operands fetched PF k-steps ahead (deeper did not help)
            double bq[PF + 1], aq_[PF + 1][4];
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                bq[j] = bp[4 * j];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) aq_[j][rb] = ap[16 * rb + LDV * 4 * j];
            }
            const double *vout = Vs + lane + LDV * (31 + wv);          // columns 32.. of Vxx_{i+1}: 31 + w + 3j, j = 0..10
            double *gout = Vxxg + nn * (i + 1) + lane + n * (31 + wv);
            double vprev = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                // one MFMA, then the pieces of traffic that issue in its 64-cycle shadow: the wave issues in order, so fetches and
                // stores bunched in front of the four products of a k-step leave the matrix pipe idle while they issue
                const int c = kk % (PF + 1), j = (kk + PF) % (PF + 1);
                const bool pf = kk + PF < 16;
                MF_SB;
                acc[0] = mf(aq_[c][0], bq[c], acc[0]);
                MF_SB;
                if (pf && !(C4_EXP & 1)) { bq[j] = bp[4 * (kk + PF)]; aq_[j][0] = ap[LDV * 4 * (kk + PF)]; }
                MF_SB;
                acc[1] = mf(aq_[c][1], bq[c], acc[1]);
                MF_SB;
                if (pf && !(C4_EXP & 1)) aq_[j][1] = ap[16 + LDV * 4 * (kk + PF)];
                if (!(C4_EXP & 2) && kk >= 1 && kk <= 11 && (kk < 11 || wv < 3)) gout[n * 3 * (kk - 1)] = vprev;
                MF_SB;
                acc[2] = mf(aq_[c][2], bq[c], acc[2]);
                MF_SB;
                if (pf && !(C4_EXP & 1)) aq_[j][2] = ap[32 + LDV * 4 * (kk + PF)];
                if (!(C4_EXP & 2) && kk < 11 && (kk < 10 || wv < 3)) vprev = vout[LDV * 3 * kk];
                MF_SB;
                acc[3] = mf(aq_[c][3], bq[c], acc[3]);
                MF_SB;
                if (pf && !(C4_EXP & 1)) aq_[j][3] = ap[48 + LDV * 4 * (kk + PF)];
                if (!(C4_EXP & 2) && kk >= 7 && ldF) pfF[kk - 7] = load_F1(i - 1, kk - 7);
            }
            MF_SB;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                double *wp = WT + 16 * wv + l15 + LD * (16 * rb + l4);
                wp[0] = acc[rb].x; wp[LD * 4] = acc[rb].y; wp[LD * 8] = acc[rb].z; wp[LD * 12] = acc[rb].w;
            }
            if (EARLY3) {
                // ================= phase 3a (no barrier: this wave's own W column): cxx + fx'W of the tiles (ti, c) ==========
                wave_sync();
                const double *apt[4], *bpt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int ti, tj;
                    tile_w(wv, u, ti, tj);
                    acc3[u] = d4{cxxr[u][0], cxxr[u][1], cxxr[u][2], cxxr[u][3]};
                    apt[u] = Fs + l4 + LDK * (16 * ti + l15);             // A[i][k] = F[k, 16ti+i]
                    bpt[u] = WT + 16 * tj + l15 + LD * l4;                // B[k][j] = W[k, 16tj+j]
                }
                if (wv == 1) {
                    mfma_chain_shared<16, 2, 1, LD * 4, 4, false>(bpt[0], apt, acc3);               // (0,0)
                    mfma_chain_shared<16, 2, 2, LD * 4, 4, false>(bpt[1], apt + 1, acc3 + 1);       // (0,1) (1,1): B = W[:, 16..31] shared
                } else if (wv == 2) mfma_chain_shared<16, 2, 3, LD * 4, 4, false>(bpt[0], apt, acc3);
                else mfma_chain_shared<16, 2, 4, LD * 4, 4, false>(bpt[0], apt, acc3);
            }
        }
        MFP(2);
        __syncthreads();
        MFP(3);
        if (flag[0] != 0.0) { diverge = i + 1; break; }              // block-uniform

        // ================= phase 3: Vxx_i = cxx + fx'W + ½(K'Y + Y'K), symmetrised (:69-72, :210); Vx_i ==========
        if (!EARLY3) {
            d4 acc[3];
            const double *apt[3], *bpt[3];
            double kA[3][2], yA[3][2], kB[3][2], yB[3][2], *qp[3], *mp[3];
            bool diag[3], valid[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                int ti, tj;
                valid[u] = tile_w(wv, u, ti, tj);
                diag[u] = ti == tj;
                const int gj = 16 * tj + l15, gi0 = 16 * ti + l4;
                qp[u] = Vs + gj + LDV * gi0;                          // Vxx[gi, gj] stored at (gj, gi): lanes contiguous
                mp[u] = Vs + gi0 + LDV * gj;                          // mirror position (gi, gj)
                acc[u] = d4{cxxr[u][0], cxxr[u][1], cxxr[u][2], cxxr[u][3]};
                apt[u] = Fs + l4 + LDK * (16 * ti + l15);             // A[i][k] = F[k, 16ti+i]
                bpt[u] = WT + gj + LD * l4;                           // B[k][j] = W[k, 16tj+j]
                const int ia = l4 + KS * (16 * ti + l15), ib = l4 + KS * gj;
                kA[u][0] = Ks[ia]; kA[u][1] = Ks[ia + 4]; yA[u][0] = Ys[ia]; yA[u][1] = Ys[ia + 4];
                kB[u][0] = 0.5 * Ks[ib]; kB[u][1] = 0.5 * Ks[ib + 4]; yB[u][0] = 0.5 * Ys[ib]; yB[u][1] = 0.5 * Ys[ib + 4];
            }
            if (wv < 2) mfma_chain_shared<16, 2, 3, 4, LD * 4, true>(apt[0], bpt, acc);           // rows 0, 1: three tiles
            else if (wv == 2) mfma_chain_shared<16, 2, 2, 4, LD * 4, true>(apt[0], bpt, acc);     // row 2: two tiles
            else mfma_chain_shared<16, 2, 2, LD * 4, 4, false>(bpt[0], apt, acc);                // column 3: (0,3), (3,3)
#pragma unroll
            for (int u = 0; u < 3; ++u) acc[u] = mf(kA[u][0], yB[u][0], acc[u]);      // the rank-16 update on the same accumulators
#pragma unroll
            for (int u = 0; u < 3; ++u) acc[u] = mf(kA[u][1], yB[u][1], acc[u]);
#pragma unroll
            for (int u = 0; u < 3; ++u) acc[u] = mf(yA[u][0], kB[u][0], acc[u]);
#pragma unroll
            for (int u = 0; u < 3; ++u) acc[u] = mf(yA[u][1], kB[u][1], acc[u]);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (!valid[u]) continue;
                // Off-diagonal tiles exist once and are mirrored.  A diagonal tile holds both (i,j) and (j,i), equal up to rounding:
                // its upper triangle is mirrored in the same way, so the result is exactly symmetric without an exchange
                // (the reference averages the two halves, (:71-72); the difference is of the order of the rounding error of G).
                const double av[4] = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!diag[u] || l4 + 4 * r <= l15) { qp[u][LDV * 4 * r] = av[r]; mp[u][4 * r] = av[r]; }
                }
            }
        }
        // ================= EARLY3, phase 3b: + ½(K'Y + Y'K) on the accumulators of phase 3a, symmetrised ==========
        auto phase3b = [&](auto wc) __attribute__((always_inline)) {
            constexpr int W_ = decltype(wc)::value;               // the wave as a compile-time constant: its tile list unrolls without tests
            double kA[4][2], yA[4][2], kB[4][2], yB[4][2], *qp[4], *mp[4];
            bool diag[4], valid[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int ti, tj;
                valid[u] = tile_w(W_, u, ti, tj);
                diag[u] = ti == tj;
                if (!valid[u]) continue;
                const int gj = 16 * tj + l15, gi0 = 16 * ti + l4;
                qp[u] = Vs + gj + LDV * gi0;                          // Vxx[gi, gj] stored at (gj, gi): lanes contiguous
                mp[u] = Vs + gi0 + LDV * gj;                          // mirror position (gi, gj)
                const int ia = l4 + KS * (16 * ti + l15), ib = l4 + KS * gj;
                kA[u][0] = Ks[ia]; kA[u][1] = Ks[ia + 4]; yA[u][0] = Ys[ia]; yA[u][1] = Ys[ia + 4];
                kB[u][0] = 0.5 * Ks[ib]; kB[u][1] = 0.5 * Ks[ib + 4]; yB[u][0] = 0.5 * Ys[ib]; yB[u][1] = 0.5 * Ys[ib + 4];
            }
            // the four products of a tile depend on each other: run the tiles of the wave side by side
#pragma unroll
            for (int u = 0; u < 4; ++u) if (valid[u]) acc3[u] = mf(kA[u][0], yB[u][0], acc3[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) if (valid[u]) acc3[u] = mf(kA[u][1], yB[u][1], acc3[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) if (valid[u]) acc3[u] = mf(yA[u][0], kB[u][0], acc3[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) if (valid[u]) acc3[u] = mf(yA[u][1], kB[u][1], acc3[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!valid[u]) continue;
                const double av[4] = {acc3[u].x, acc3[u].y, acc3[u].z, acc3[u].w};      // see the note on diagonal tiles above
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!diag[u] || l4 + 4 * r <= l15) { qp[u][LDV * 4 * r] = av[r]; mp[u][4 * r] = av[r]; }
                }
            }
        };
        if (EARLY3) {
            if (wv == 1) phase3b(std::integral_constant<int, 1>{});
            else if (wv == 2) phase3b(std::integral_constant<int, 2>{});
            else if (wv == 3) phase3b(std::integral_constant<int, 3>{});
        }
        if (wv == (EARLY3 ? 0 : 3)) {                                // Vx_i (:69)
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int q = 0; q < m; ++q) {
                s1 += Ks[q + KS * lane] * Quuks[q];
                s2 += Ks[q + KS * lane] * Qs[n + q];
                s3 += Xs[q + m * lane] * ks[q];
            }
            const double v = ((Qs[lane] + s1) + s2) + s3;
            vs[lane] = v; Vxg[(size_t)n * i + lane] = v;
        }
        MFP(4);
        if (!EARLY3) __syncthreads();                                // every wave is done reading Fs (EARLY3: its last reader, phase 3a, sits in front of the barrier above)
        MFP(5);
        if (ldF) store_F(pfF);
        MFP(6);
        __syncthreads();
        MFP(7);
    }
    MFP_PRINT;
    if (diverge) {   // outputs earlier in time than the failing step are zero (backward_pass.jl:226-229)
        const size_t ie = (size_t)diverge;
        for (size_t e = tid; e < nm * ie; e += NT) Kg[e] = 0.0;
        for (size_t e = tid; e < (size_t)m * ie; e += NT) kg[e] = 0.0;
        for (size_t e = tid; e < (size_t)n * ie; e += NT) Vxg[e] = 0.0;
        for (size_t e = tid; e < nn * ie; e += NT) Vxxg[e] = 0.0;
        for (size_t e = tid; e < mm * (ie - 1); e += NT) Quug[e] = 0.0;
    } else {
        for (int e = tid; e < n * n; e += NT) Vxxg[e] = Vs[(e & 63) + LDV * (e >> 6)];
    }
    if (tid == 0) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
}

}   // namespace

template <bool LIMS, bool CTV>
static int ddp_bpm_launch_tv(ddp_handle h, const BPMArgs &a)
{
    const size_t shmem = (size_t)oTot * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mfma_kernel<LIMS, CTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((back_pass_mfma_kernel<LIMS, CTV>), dim3(a.B), dim3(NT), shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}

// time-invariant cost: its terms are loaded once, before the loop — no load (and no s_waitcnt on the in-order vmcnt) inside
template <bool LIMS>
static int ddp_bpm_launch(ddp_handle h, const BPMArgs &a) { return a.cost_tv ? ddp_bpm_launch_tv<LIMS, true>(h, a) : ddp_bpm_launch_tv<LIMS, false>(h, a); }

int ddp_bpm_launch_lims(ddp_handle h, const BPMArgs &a);      // back_pass_mfma_lims.hip
