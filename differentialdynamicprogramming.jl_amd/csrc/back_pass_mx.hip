// back_pass_mx.hip — backward pass for the unconstrained (Cholesky) path, n = 10, m = 2, on the fp64 MATRIX
// cores: one wavefront per trajectory, every matrix of a time step lives in ONE 16x16 v_mfma_f64_16x16x4_f64
// tile held in registers.  Same arithmetic as src/backward_pass.jl:162-252 + :28-42,:64-76.
//
// Why: at the BASELINE batch (1024 trajectories = one wave per SIMD) the pass is bound by the dependent
// instruction chain of a time step, not by HBM.  The vector kernels (back_pass_fast/_dpp) spend ~300
// instructions per step, most of them moving operands between lanes (LDS reads, DPP broadcasts, readlanes).
// The MFMA register layouts make that movement disappear:
//
//   tile layouts (lane = 16*l4 + l15):   A operand: A[i = l15][k = l4]      B operand: B[k = l4][j = l15]
//                                        accumulator register r: D[row = l4 + 4r][col = l15]
//   Since a product sums over k, the k index may be permuted as long as A and B use the same permutation.
//   With k = l4 + 4s for MFMA step s, accumulator register s of one product IS operand register s of the next:
//
//   GEMM1  W = Vxx·F + C1      A_s = Vxx (accumulator layout; Vxx is symmetric), B_s = F_s                    3 MFMA
//   GEMM2  G = F'·W + H        A_s = F_s (the same registers), B_s = W_s (the accumulator of GEMM1)           3 MFMA
//   VALUE  V = G + [K' Qux']·[T; K]    one k=4 MFMA: k-slices (K0,T0) (K1,T1) (Qux0,K0) (Qux1,K1)             1 MFMA
//   where F = [fx fu] (10 x 12), H = [cxx cxu; cxu' cuu], T = Quu·K + Qux, and column 12 of the tile carries the
//   vectors: W[:,12] := Vx (injected through C1), H[:,12] = [cx;cu], so G[:,12] = [Qx;Qu], the solve of
//   "column 12" is k_i, and V[:,12] = Vx_i — the vector recursion costs no extra instruction.
//   G rows 10,11 (= Qux | Quu | Qu) are replicated to all four 16-lane rows by two gfx950 row swaps
//   (v_permlane32_swap, v_permlane16_swap), the 2x2 Quu is broadcast inside the rows by DPP, the 2x2 system is
//   solved redundantly by every lane, so that K and T are lane-local and feed the value MFMA directly.
//   ½(V + V') (:71-72) takes one trip through a padded LDS tile (write the accumulator, read it transposed).
// Measured on gfx950 (profiles/microbench/chain_latency.hip): the fp64 MFMA occupies the SIMD for 64 cycles and
// vector instructions do not overlap with it, so a step costs  7 MFMA x 64  +  ~5.5 cycles per vector instruction;
// the kernel is therefore written to minimise the INSTRUCTION COUNT of a step (~70 vector instructions).
#include "ddp_internal.h"

#ifndef MX_EXP
#define MX_EXP 0          // timing experiments (wrong results): 1 no transpose round trip, 2 no group write-back, 3 both
#endif

namespace {

#include "back_pass_mx_common.h"

// RT: run-time sizes nr <= 10, mr <= 2 in the same tile (state j in tile row/column j, rows nr..9 exact zeros, the controls in tile rows
// 10, 11; for mr = 1 tile entry (11, 11) of H is 1 so that the 2x2 system stays positive definite with K[1,:] = 0), operands and results
// with run-time strides, results straight to global memory (no step records)
template <bool FXTV, bool CTV, bool REG2, bool LCH, bool RT>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_mx_kernel(BPXArgs a)
{
    const int b = blockIdx.x, lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    if (a.active && a.active[b] == 0) return;
    const int N = a.N;
    static_assert(!(RT && LCH), "the step records have the layout of the exact shape");
    const int nr = RT ? a.n : n, mr = RT ? a.m : m;
    const size_t nn = (size_t)nr * nr, nm = (size_t)nr * mr, mm = (size_t)mr * mr;

    __shared__ __attribute__((aligned(16))) double lds[TLD * 16 + 16];      // transpose tile + zero cells
    __shared__ __attribute__((aligned(16))) double lout[LCH ? LOUT + LDUMP_SZ : 2];
    __shared__ __attribute__((aligned(16))) double leb[2][LCH ? 128 : 2];

    const double *cx = a.cx + (size_t)nr * N * b, *cu = a.cu + (size_t)mr * N * b;
    const double *fx = a.fx + (a.fx_batched ? nn * (FXTV ? N : 1) * b : 0);
    const double *fu = a.fu + (a.fx_batched ? nm * (FXTV ? N : 1) * b : 0);
    const double *cxx = a.cxx + (a.cost_batched ? nn * (CTV ? N : 1) * b : 0);
    const double *cxu = a.cxu + (a.cost_batched ? nm * (CTV ? N : 1) * b : 0);
    const double *cuu = a.cuu + (a.cost_batched ? mm * (CTV ? N : 1) * b : 0);
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)mr * N * b, *Quug = a.Quu + mm * N * b,
           *Vxg = a.Vx + (size_t)nr * N * b, *Vxxg = a.Vxx + nn * N * b;
    const double lam = a.lambda[b];

    // ---- terminal step (backward_pass.jl:234-236 / :197-199)
    const size_t tl = (size_t)(N - 1);
    for (int e = lane; e < nr * nr; e += DDP_WAVE) Vxxg[nn * tl + e] = cxx[(CTV ? nn * tl : 0) + e];
    if (lane < nr) Vxg[(size_t)nr * tl + lane] = cx[(size_t)nr * tl + lane];
    if (lane < mr * mr) Quug[mm * tl + lane] = cuu[(CTV ? mm * tl : 0) + lane];
    if (lane < mr * nr) Kg[nm * tl + lane] = 0.0;
    if (lane < mr) kg[(size_t)mr * tl + lane] = 0.0;
    if (N < 2) {
        if (lane == 0) { a.dV[2 * b] = 0.0; a.dV[2 * b + 1] = 0.0; a.diverge[b] = 0; }
        return;
    }
    for (int e = lane; e < TLD * 16 + 16; e += DDP_WAVE) lds[e] = 0.0;

    // ---- per-lane operand streams ------------------------------------------------------------------------------
    // tile coordinate -> state index (< nr), control index (tile rows n.. = 10, 11), or nothing (-1)
    auto six = [&](int r) { return r < nr ? r : -1; };
    auto uix = [&](int r) { return (r >= n && r < n + mr) ? r - n : -1; };
    const Stream zeroS = Stream{(const char *)mx_zero, 0u, nullptr};
    auto h_stream = [&](int row, int col) -> Stream {       // H = [cxx cxu; cxu' cuu] in tile coordinates, zero outside
        const int sr = six(row), sc = six(col), ur = uix(row), uc = uix(col);
        if (sr >= 0 && sc >= 0) return Stream{(const char *)(cxx + sr + nr * sc), CTV ? (unsigned)(nn * 8) : 0u, nullptr};
        if (sr >= 0 && uc >= 0) return Stream{(const char *)(cxu + sr + nr * uc), CTV ? (unsigned)(nm * 8) : 0u, nullptr};
        if (ur >= 0 && sc >= 0) return Stream{(const char *)(cxu + sc + nr * ur), CTV ? (unsigned)(nm * 8) : 0u, nullptr};
        if (ur >= 0 && uc >= 0) return Stream{(const char *)(cuu + ur + mr * uc), CTV ? (unsigned)(mm * 8) : 0u, nullptr};
        if (RT && row == col && row >= n + mr && row < p) return Stream{(const char *)mx_one, 0u, nullptr};     // the unused control of mr = 1
        return zeroS;
    };
    auto f_stream = [&](int row, int col) -> Stream {       // F = [fx fu] in tile coordinates, zero outside
        const int sr = six(row), sc = six(col), uc = uix(col);
        if (sr >= 0 && sc >= 0) return Stream{(const char *)(fx + sr + nr * sc), FXTV ? (unsigned)(nn * 8) : 0u, nullptr};
        if (sr >= 0 && uc >= 0) return Stream{(const char *)(fu + sr + nr * uc), FXTV ? (unsigned)(nm * 8) : 0u, nullptr};
        return zeroS;
    };
    // Tile rows 12..15 repeat the u-rows 10, 11, 10, 11 (F columns 12..15 of the A operand of GEMM2 repeat columns 10, 11, and
    // so do the rows of H): accumulator register 3 of G then holds, in EVERY 16-lane row, the u-row with the parity of that row.
    const int urow = n + (l4 & 1);
    Stream hS[4], fS[3];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = s < 3 ? l4 + 4 * s : urow;
        hS[s] = l15 == VC ? zeroS : h_stream(row, l15);   // C of GEMM2 (column VC: below)
        if (s < 3) fS[s] = f_stream(l4 + 4 * s, l15 < p ? l15 : n + (l15 & 1));
    }
    // The vector e = [cx; cu] (p entries) of tile column VC costs ONE load per step: lane (l4, j) fetches e[j + l4], so that
    // the entry l4 + 4s that accumulator register s wants in column VC sits in lane 4s of the same 16-lane row for EVERY
    // row — a row broadcast folded into the multiply-add that builds the C operand (register 3: e[10 + (l4&1)] = lane 10
    // of rows 0,1 and lane 8 of rows 2,3).
    const int eidx = l15 + l4 < p ? l15 + l4 : p - 1;
    Stream eS = six(eidx) >= 0 ? Stream{(const char *)(cx + eidx), (unsigned)(nr * 8), nullptr}
                               : (uix(eidx) >= 0 ? Stream{(const char *)(cu + (eidx - n)), (unsigned)(mr * 8), nullptr} : zeroS);

    // ---- loop-invariant lane constants ---------------------------------------------------------------------------
    const double mask12 = l15 == VC ? 1.0 : 0.0;
    const double cB = l4 >= 2 ? 1.0 : -lam;                   // regType 1: T = -λK in the 16-lane rows 0, 1
    const bool odd = (l4 & 1) != 0, hi2 = l4 >= 2;
    const int wr = l4 + TLD * l15;                           // accumulator register s -> tile element (l4+4s, l15)
    const int rdT = l15 == VC ? TZERO : l15 + TLD * l4;      // its transpose (l15, l4+4s): + 4*TLD per register
    const int rdS = l15 == VC ? 0 : 4 * TLD;
    // Vxx | Vx ride on the same three stores: columns < n: Vxx[l4+4s, col]; column VC: Vx[l4+4s]
    const bool v_col = l15 < nr || l15 == VC;
    const bool v_act01 = v_col, v_act2 = v_col && l4 < 2 && (!RT || l4 + 8 < nr);
    const double vscl = l15 == VC ? 1.0 : 0.5;               // registers hold V + V'; column VC holds Vx itself
    char *vst = l15 == VC ? (char *)(Vxg + (size_t)nr * (tl - 1) + l4) : (char *)(Vxxg + nn * (tl - 1) + l4 + nr * (l15 < nr ? l15 : 0));
    const unsigned vst_stride = l15 == VC ? (unsigned)(nr * 8) : (unsigned)(nn * 8);
    // K | k | Quu ride on one store: lanes of rows 2,3: columns <n: K[a, col]; column VC: k[a]; columns n..n+1: Quu[a, col-n]
    const bool quu_lane = hi2 && (l15 == n || l15 == n + 1);
    const int a2 = hi2 ? l4 - 2 : 0;
    const bool kq_act = hi2 && a2 < mr && (l15 < nr || l15 == VC || uix(l15) >= 0);
    const unsigned long long lanes01 = __builtin_amdgcn_ballot_w64(v_act01), lanes2k = __builtin_amdgcn_ballot_w64(hi2 ? kq_act : v_act2);
    // RT: the rows l4 and l4 + 4 of a column are stored separately (either may lie in the padding)
    const unsigned long long lanes0 = __builtin_amdgcn_ballot_w64(v_col && l4 < nr), lanes1 = __builtin_amdgcn_ballot_w64(v_col && l4 + 4 < nr);
    char *kq = !hi2 ? vst + 64
                    : (l15 < nr ? (char *)(Kg + nm * (tl - 1) + a2 + mr * l15)
                                : (l15 == VC ? (char *)(kg + (size_t)mr * (tl - 1) + a2) : (char *)(Quug + mm * (tl - 1) + a2 + mr * (uix(l15) >= 0 ? l15 - n : 0))));
    const unsigned kq_stride = !hi2 ? vst_stride : (l15 < nr ? (unsigned)(nm * 8) : (l15 == VC ? (unsigned)(mr * 8) : (unsigned)(mm * 8)));

    // ---- LCH: where this lane's results go in a step record, and its share of the write-back of a group
    // accumulator registers 0, 1 (and 2 in 16-lane rows 0, 1): Vxx[l4+4s, l15] | column VC: Vx[l4+4s]; register 2 of rows 2, 3: K | k | Quu
    const int w1 = l15 < n ? l4 + n * l15 : (l15 == VC ? R_VX + l4 : LDUMP + lane);
    const int w2 = !hi2 ? w1 + 8
                        : (l15 < n ? R_K + a2 + m * l15 : (l15 == VC ? R_KV + a2 : (l15 < p ? R_QUU + a2 + m * (l15 - n) : LDUMP + lane)));
    // write-back: Vxx of one step = 50 lanes x 16 bytes; Vx, K, k, Quu of three steps = 54 lanes x 16 bytes
    const int msub = lane / 18, mw = lane % 18;
    const int mL = REC * msub + (mw < 5 ? R_VX + 2 * mw : (mw < 15 ? R_K + 2 * (mw - 5) : (mw == 15 ? R_KV : R_QUU + 2 * (mw - 16))));
    const unsigned mstep = mw < 5 ? n * 8u : (mw < 15 ? (unsigned)(nm * 8) : (mw == 15 ? m * 8u : (unsigned)(mm * 8)));     // bytes per time step
    char *pV = nullptr, *pM = nullptr;
    const char *pE = nullptr;
    if constexpr (LCH) {
        const long t0 = (long)N - 2 - (PD - 1);              // lowest step of the first group
        pV = (char *)(Vxxg + (long)nn * t0 + 2 * (lane < 50 ? lane : 0));
        char *mb = mw < 5 ? (char *)(Vxg + 2 * mw) : (mw < 15 ? (char *)(Kg + 2 * (mw - 5)) : (mw == 15 ? (char *)kg : (char *)(Quug + 2 * (mw - 16))));
        pM = mb + (long)mstep * (t0 + (lane < 54 ? msub : 0));
        const int et = lane < 48 ? lane / 6 : 0, ew = lane % 6;                                                             // [cx; cu] of step t0 + et, pair ew
        pE = ew < 5 ? (const char *)(cx + (long)n * (t0 + et) + 2 * ew) : (const char *)(cu + (long)m * (t0 + et));
    }
    const unsigned estep = (lane % 6) < 5 ? n * 8u : m * 8u;

    // ---- register-resident operands -------------------------------------------------------------------------------
    const double hmask = l15 < p ? 0.5 : 0.0;
    double F[3], Fh[3], Hc[4];                               // F_s (A of GEMM2), ½F_s without the repeated columns (B of GEMM1: A carries 2 Vxx), C (H)
    double er[PD], hr[CTV ? PD : 1][4], fr[FXTV ? PD : 1][3];
    const int i0 = N - 2;
#pragma unroll
    for (int j = 0; j < PD; ++j) {
        const int t = i0 - j > 0 ? i0 - j : 0;
        er[j] = LCH ? 0.0 : eS.at(t);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (FXTV && s < 3) fr[j][s] = fS[s].at(t);
            if (CTV) hr[j][s] = hS[s].at(t);
        }
    }
    {
        const int t = i0 - PD > 0 ? i0 - PD : 0;
        eS.seek(t);
#pragma unroll
        for (int s = 0; s < 4; ++s) { hS[s].seek(t); if (s < 3) fS[s].seek(t); }
    }
    if (!FXTV) {
#pragma unroll
        for (int s = 0; s < 3; ++s) { F[s] = fS[s].at(0); Fh[s] = hmask * F[s]; }
    }
    if (!CTV) {
#pragma unroll
        for (int s = 0; s < 4; ++s) Hc[s] = hS[s].at(0);
    }
    const d4 Hc4 = d4{Hc[0], Hc[1], Hc[2], Hc[3]};

    // value function of the terminal step in tile layout: S = 2 Vxx, column VC: Vx
    double S[3];
    {
        const Stream hs[3] = {h_stream(l4, l15), h_stream(l4 + 4, l15), h_stream(l4 + 8, l15)};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int row = l4 + 4 * s;
            S[s] = (l15 < nr && row < nr) ? 2.0 * hs[s].at((int)tl) : ((l15 == VC && row < nr) ? cx[(size_t)nr * tl + row] : 0.0);
        }
    }
    const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
    double *ecur = leb[0], *enxt = leb[1];
    auto dma_e = [&](double *dst) {                 // [cx; cu] of the PD steps pE points at -> PD records of EREC doubles
        if (lane < 48) __builtin_amdgcn_global_load_lds((glb_void *)pE, (lds_void *)dst, 16, 0, 0);
    };
    if (LCH && i0 >= PD - 1) { dma_e(ecur); pE -= (size_t)PD * estep; }
    wave_sync();
    // all set-up loads have landed: the waits inside the loop are then computed from the steady state only
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)

    double dVa = 0.0, dVp = 0.0;                    // Σ k'Qu  and the per-row halves of Σ k'(Quu k + Qu)
    int diverge = 0;
    // One time step.  No exits inside: a diverged trajectory (wave-uniform) keeps stepping through garbage with its
    // stores switched off until the loop around the step looks at `diverge` — an exit edge here would make the waits at
    // the loop head cover the path "just refilled this ring slot -> loop head" and drain the memory queue every time.
    // mode 0: operands and results straight from / to global memory; 1 (LCH groups): [cx;cu] from the LDS, results into the step
    // record of the group; 2 (LCH, the steps left over below the last group): like 0, er[] was loaded for them
    auto step = [&](const int i, auto slot_c, auto mode_c) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value, mode = decltype(mode_c)::value;
        constexpr int tau = PD - 1 - slot;                  // position of the step in its group (ascending time)
        const bool okp = diverge == 0;
        // ---- operands of this step (ring slot `slot`; the slot is refilled at the END of the step, when its old
        //      contents are dead, so that the prefetch lands in the same registers: no copies, no early waits)
        // C operand of GEMM2: H only.  The vector e = [cx;cu] of column VC enters later and in place: G[:,VC] is used by the
        // u-rows (Qu, below) and, linearly, by V[:,VC] = Vx (added after the value MFMA).
        const d4 c = CTV ? d4{hr[slot][0], hr[slot][1], hr[slot][2], hr[slot][3]} : Hc4;
        const double e = mode == 1 ? ecur[EREC * tau + eidx] : er[slot];
        // LCH groups: the entry cu[parity of my 16-lane row] that column VC of the u-rows wants (Qu = cu + fu'Vx) comes by a
        // second LDS read instead of two row broadcasts out of e (an LDS read costs ~8 issue cycles off the chain, the two
        // v_fmac_f64_dpp + their wait states ~40 on it)
        const double eu = mode == 1 ? ecur[EREC * tau + n + (l4 & 1)] : 0.0;
        if (FXTV) {
#pragma unroll
            for (int s = 0; s < 3; ++s) { F[s] = fr[slot][s]; Fh[s] = hmask * F[s]; }
        }
        // ================= GEMM1: W = Vxx·F; column VC := Vx (F[:,VC] = 0, S[:,VC] = Vx) ============================
        d4 w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[0], Fh[0], zero4, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[1], Fh[1], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[2], Fh[2], w, 0, 0, 0);
        const double W[3] = {fma(S[0], mask12, w.x), fma(S[1], mask12, w.y), fma(S[2], mask12, w.z)};
        // ================= GEMM2: G = F'W + H, column VC: [cx;cu] + F'Vx  (:203-210) ================================
        d4 g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], W[0], c, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], W[1], g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2], W[2], g, 0, 0, 0);
        // ================= gains (backward_pass.jl:30-42) =============================================================
        double Z;                                          // G row 10 | 11 (Qux | Quu | Qu) with the parity of my 16-lane row
        if (mode == 1) {
            Z = fma(eu, mask12, g.w);                      // column VC: Qu = cu + fu'Vx
        } else {
            Z = g.w + 0.0;
            fmac_bcast<10, 0x3, true>(Z, e, mask12);       // (e[10 + (l4&1)]: lane 10 of rows 0,1,
            fmac_bcast<8, 0xc>(Z, e, mask12);              //  lane 8 of rows 2,3)
        }
        double Q0, Q1;                                     // row 10, row 11 in every lane
        double F00, F01, F11;                              // QuuF (:205-207)
        if (REG2) {                                        // u-rows of F'(W + λF) + H: Qux_reg, QuuF;  λF = 2λ·(½F)
            const double lam2 = 2.0 * lam;
            d4 gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], fma(lam2, Fh[0], W[0]), c, 0, 0, 0);
            gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], fma(lam2, Fh[1], W[1]), gr, 0, 0, 0);
            gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2], fma(lam2, Fh[2], W[2]), gr, 0, 0, 0);
            double Zr = gr.w + 0.0;                        // column VC of the regularised rows is still Qu (F[:,VC] = 0)
            fmac_bcast<10, 0x3, true>(Zr, e, mask12);
            fmac_bcast<8, 0xc>(Zr, e, mask12);
            spread_pair(Zr, Q0, Q1);
            F00 = row_bcast<n>(Q0); F01 = row_bcast<n + 1>(Q0); F11 = row_bcast<n + 1>(Q1);
        } else {
            spread_pair(Z, Q0, Q1);
            F00 = row_bcast<n>(Q0) + lam; F01 = row_bcast<n + 1>(Q0); F11 = row_bcast<n + 1>(Q1) + lam;
        }
        // x = -(QuuF)\b for every column b = (Q0, Q1)[:, j]: K (:42); column VC: k_i (:41).  Explicit 2x2 inverse:
        // positive definite <=> F00 > 0 and det > 0 — the pivots of cholesky(Hermitian(QuuF)) (:35-38)
        const double det = fma(F00, F11, -(F01 * F01));
        const bool bad = !(F00 > 0.0) || !(det > 0.0);
        // 1/det (hardware estimate + two Newton steps, rcp_nr) with the numerators of the 2x2 solve in the stalls of its dependent
        // chain: the compiler's schedule put them behind it (a dependent fp64 instruction issues ~10 cycles after its producer, an
        // independent one after 6)
        auto pin = [](double &x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); };
        double y = __builtin_amdgcn_rcp(det); pin(y);
        double t01 = F01 * Q1; pin(t01);
        double e1 = fma(-det, y, 1.0); pin(e1);
        double t10 = F01 * Q0; pin(t10);
        y = fma(y, e1, y); pin(y);
        double n0 = fma(F11, Q0, -t01); pin(n0);
        double e2 = fma(-det, y, 1.0); pin(e2);
        double n1 = fma(F00, Q1, -t10); pin(n1);
        y = fma(y, e2, y);
        const double nidet = -y;
        const double K0 = n0 * nidet;
        const double K1 = n1 * nidet;
        // my u-row: K_a, T_a = Qux_a + Quu[a,:]·K  (:64); a = parity of the 16-lane row
        const double Ksel = odd ? K1 : K0;
        double Tsel, Bop;
        if (!REG2) {
            // regType 1: QuuF·K = -Qux with QuuF = Quu + λI, hence T = Quu·K + Qux = -λK — the residual form (as in the n = 64
            // kernel): one multiply instead of two dependent row-broadcast multiply-adds, and the B operand of the value
            // product (T_a in the 16-lane rows 0, 1; K_a in rows 2, 3) is K_a times a lane constant instead of a select
            Bop = Ksel * cB;
            Tsel = Bop;                                    // (only read where it is T: rows 0, 1)
        } else {
            Tsel = Z;
            fmac_bcast<n, 0xf, true>(Tsel, Z, K0);
            fmac_bcast<n + 1>(Tsel, Z, K1);
            Bop = hi2 ? Ksel : Tsel;
        }
        // ================= value update (:69-72): V = G + [K' Qux']·[T; K] ===========================================
        const double Aop = hi2 ? Z : Ksel;
        // (columns 10, 11 of V are junk; Quu_i for the stores is taken from Z below)
        const d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop, Bop, g, 0, 0, 0);
        // ---- while the value MFMA runs: outputs that do not depend on it, bookkeeping
        // wave-uniform branches, not selects: the common path is two multiply-adds
        const bool badu = (__builtin_amdgcn_ballot_w64(!(F00 > 0.0)) | __builtin_amdgcn_ballot_w64(!(det > 0.0))) != 0;
        if (__builtin_expect(badu || !okp, 0)) {
            asm volatile("" ::: "memory");
            if (okp) diverge = i + 1;                      // diverge = i (:37-38)
        } else {                                           // column VC: k'Qu and k_a (Quu k + Qu)_a  (:68)
            asm volatile("" ::: "memory");
            dVa = fma(K1, Q1, fma(K0, Q0, dVa));
            dVp = fma(Ksel, Tsel, dVp);
        }
        // ---- ½(V + V') through the transpose tile; registers keep V + V' (column VC: Vx)
        if (MX_EXP & 1) { S[0] = v.x + v.x; S[1] = v.y + v.y; S[2] = v.z + v.z; }
        else {
        lds[wr] = v.x; lds[wr + 4] = v.y; lds[wr + 8] = v.z;
        wave_sync();
        S[0] = v.x + lds[rdT]; S[1] = v.y + lds[rdT + rdS]; S[2] = v.z + lds[rdT + 2 * rdS];
        }
        fmac_bcast<0>(S[0], e, mask12);                    // column VC: Vx += cx  (e[l4 + 4s]: lane 4s of my 16-lane row)
        fmac_bcast<4>(S[1], e, mask12);
        fmac_bcast<8>(S[2], e, mask12);
        // Stores are unconditional: a diverged trajectory writes garbage into time steps that are zero-filled after the loop
        // (behind a vmcnt(0) wait); the failing step itself leaves the Quu_i the reference returns.
        if (mode == 1) {
            lout[w1 + REC * tau] = vscl * S[0];
            lout[w1 + 4 + REC * tau] = vscl * S[1];
            lout[w2 + REC * tau] = hi2 ? (quu_lane ? Z : Ksel) : vscl * S[2];
        } else {
            if (RT) { store_masked(vst, vscl * S[0], lanes0); store_masked(vst + 32, vscl * S[1], lanes1); }
            else store2_masked(vst, vscl * S[0], vscl * S[1], lanes01);
            // rows 8, 9 of Vxx | Vx (16-lane rows 0,1) and K | k | Quu (:75-76) (rows 2,3) share one store
            store_masked(kq, hi2 ? (quu_lane ? Z : Ksel) : vscl * S[2], lanes2k);
            vst -= vst_stride;
            kq -= kq_stride;
        }
        wave_sync();                                       // the tile is free again
        {   // refill the ring slot with the step PD ahead (clamped: always a valid load)
            const int tp = i - PD > 0 ? i - PD : 0;
            asm volatile("" ::: "memory");
            if (mode == 0) er[slot] = eS.next();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (FXTV && s < 3) fr[slot][s] = fS[s].next();
                if (CTV) hr[slot][s] = hS[s].next();
            }
            if (i - PD > 0) {
                if (mode == 0) eS.back();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (FXTV && s < 3) fS[s].back();
                    if (CTV) hS[s].back();
                }
            }
        }
    };
    int i = i0;
    if (!LCH) {
        while (i >= PD - 1 && diverge == 0) {
            static_for<0, PD>([&](auto sc) __attribute__((always_inline)) { step(i - decltype(sc)::value, sc, IC<0>{}); });
            i -= PD;
        }
        static_for<0, PD - 1>([&](auto sc) __attribute__((always_inline)) {    // the last (N-1) mod PD steps
            if (i >= 0 && diverge == 0) { step(i, sc, IC<0>{}); --i; }
        });
    } else {
        while (i >= PD - 1 && diverge == 0) {
            if (i - PD >= PD - 1) { dma_e(enxt); pE -= (size_t)PD * estep; }       // [cx;cu] of the next group, a whole group ahead
            static_for<0, PD>([&](auto sc) __attribute__((always_inline)) { step(i - decltype(sc)::value, sc, IC<1>{}); });
            // the direct-to-LDS load issued at the top has landed once at most the ring refills of this group are outstanding
            // (loads return in order); the stores below are issued after the wait and never waited for.  The count is the number
            // of refill loads a group issues behind dma_e — one 8-byte load per streamed operand register and step, nothing the
            // compiler could merge (the lanes' addresses are unrelated) — and tests/test_gpu_edge_cases.py checks that this path and
            // the step-by-step path (DDP_MX_LDS=0) agree bit for bit in every FXTV / CTV / regType combination.
            constexpr int REFILLS = PD * ((FXTV ? 3 : 0) + (CTV ? 4 : 0));
            static_assert(REFILLS <= 63, "vmcnt is a 6-bit counter");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(REFILLS) : "memory");
            // ---- write the PD step records back: Vxx one step per instruction, the small arrays three steps per instruction
            if (!(MX_EXP & 2)) if (lane < 50) {
#pragma unroll
                for (int t = 0; t < PD; ++t) *(d2 *)(pV + nn * 8 * t) = *(const d2 *)(lout + REC * t + 2 * lane);
            }
            if (!(MX_EXP & 2)) if (lane < 54) {
                *(d2 *)pM = *(const d2 *)(lout + mL);
                *(d2 *)(pM + 3 * (size_t)mstep) = *(const d2 *)(lout + mL + 3 * REC);
                if (lane < 36) *(d2 *)(pM + 6 * (size_t)mstep) = *(const d2 *)(lout + mL + 6 * REC);
            }
            pV -= nn * 8 * PD; pM -= (size_t)mstep * PD;
            double *t_ = ecur; ecur = enxt; enxt = t_;
            i -= PD;
        }
        // the steps below the last whole group go the direct way
        vst -= (size_t)vst_stride * (unsigned)(i0 - i); kq -= (size_t)kq_stride * (unsigned)(i0 - i);
#pragma unroll
        for (int j = 0; j < PD - 1; ++j) er[j] = eS.at(i - j > 0 ? i - j : 0);
        static_for<0, PD - 1>([&](auto sc) __attribute__((always_inline)) {
            if (i >= 0 && diverge == 0) { step(i, sc, IC<2>{}); --i; }
        });
    }

    if (diverge) {   // outputs earlier in time than the failing step are zero (backward_pass.jl:226-229)
        const size_t ie = (size_t)diverge;          // = i + 1
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): the garbage of the steps after the failure has landed
        for (size_t e = lane; e < nm * ie; e += DDP_WAVE) Kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)mr * ie; e += DDP_WAVE) kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)nr * ie; e += DDP_WAVE) Vxg[e] = 0.0;
        for (size_t e = lane; e < nn * ie; e += DDP_WAVE) Vxxg[e] = 0.0;
        for (size_t e = lane; e < mm * (ie - 1); e += DDP_WAVE) Quug[e] = 0.0;
    }
    {   // dV (:68): [Σ k'Qu, ½ Σ k'Quu k];  k'Quu k = k'(Quu k + Qu) - k'Qu, the two u-rows live in lanes VC and 16+VC
        const int plo = __builtin_amdgcn_readlane(__double2loint(dVp), 16 + VC), phi = __builtin_amdgcn_readlane(__double2hiint(dVp), 16 + VC);
        const double kT = dVp + __hiloint2double(phi, plo);
        if (lane == VC) { a.dV[2 * b] = dVa; a.dV[2 * b + 1] = 0.5 * (kT - dVa); }
    }
    if (lane == 0) a.diverge[b] = diverge;
}

template <bool REG2, bool LCH, bool RT = false>
int launch_mx(ddp_handle h, const ddp_bp_desc *d, const BPXArgs &a)
{
    const dim3 grid(d->B), block(DDP_WAVE);
    const int key = (d->fx_tv ? 2 : 0) | (d->cost_tv ? 1 : 0);
    switch (key) {
    case 0: hipLaunchKernelGGL((back_pass_mx_kernel<false, false, REG2, LCH, RT>), grid, block, 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((back_pass_mx_kernel<false, true, REG2, LCH, RT>), grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((back_pass_mx_kernel<true, false, REG2, LCH, RT>), grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL((back_pass_mx_kernel<true, true, REG2, LCH, RT>), grid, block, 0, h->stream, a); break;
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// returns 1 if this shape is not handled here, 0 launched, <0 error
int ddp_launch_back_pass_mx(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                            const double *cxx, const double *cxu, const double *cuu, const double *fx,
                            const double *fu, const double *lambda, const int32_t *active, double *K,
                            double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge)
{
    if (d->has_lims || d->m != 2 || d->n != 10) return 1;
    BPXArgs a;
    a.N = d->N; a.B = d->B; a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched; a.n = d->n; a.m = d->m;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    // the group write-back needs 16-byte aligned arrays (every per-step size of this shape is a multiple of 16 bytes)
    const char *lv = ddp_env(h, ENV_MX_LDS);                    // 0: results straight to global memory, step by step (A/B, tests)
    const bool al16 = ((((uintptr_t)cx | (uintptr_t)cu | (uintptr_t)K | (uintptr_t)k | (uintptr_t)Quu | (uintptr_t)Vx | (uintptr_t)Vxx) & 15) == 0);
    if (al16 && !(lv && lv[0] == '0')) return d->regType == 2 ? launch_mx<true, true>(h, d, a) : launch_mx<false, true>(h, d, a);
    return d->regType == 2 ? launch_mx<true, false>(h, d, a) : launch_mx<false, false>(h, d, a);
}

// The same tile kernel for any n <= 10, m <= 2 without control limits (run-time sizes inside the (10, 2) tile layout); 1 = not applicable
int ddp_launch_back_pass_mxr(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const int32_t *active, double *K,
                             double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge)
{
    if (d->has_lims || d->m > 2 || d->n > 10) return 1;
    BPXArgs a;
    a.N = d->N; a.B = d->B; a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched; a.n = d->n; a.m = d->m;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    return d->regType == 2 ? launch_mx<true, false, true>(h, d, a) : launch_mx<false, false, true>(h, d, a);
}
