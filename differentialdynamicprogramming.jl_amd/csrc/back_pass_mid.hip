// back_pass_mid.hip — backward pass for the shapes between the 16-lane rows (n <= 14) and the n = 64 matrix-core kernel:
// any n <= 32, m <= 8 (src/backward_pass.jl:162-252 + :28-79), ONE wave per trajectory, the three products of a step on
// v_mfma_f64_16x16x4 with their operands in the LDS.  Before this file these shapes ran on the 64-lane vector kernel of back_pass.hip
// (n = 24, m = 4, N = 300, B = 1 024 with per-trajectory dynamics: 11.5 ms, 0.04 of HBM; n = 32, m = 8: 58 ms).
//
// Padded sizes: NR = 16 NTR >= n rows / contraction length (NTR = 1, 2), PC = 16 PT >= n + m + 1 columns (PT = 2, 3); the LDS images are
// zero outside the actual n, m, so every product over the padded range adds exact zeros.  Per step:
//   W  = Vxx F            NTR x PT tiles, NR / 4 k-steps (A: Vs, B: Fs)                       (:165 / :203 / :240, the products with Vxx)
//   Ws[:, p] := Vx        the column behind the last one of F, so that
//   G  = F' [W | Vx] + [H | c]   PT x PT tiles: Qxx, Qux, Quu and, in column p, Qx and Qu     (:165-169 / :203-210 / :240-244)
//   gains                 m <= 4: every lane the identity-padded 4 x 4 system on registers (boxqp_dev.h), lane c solves column c of K;
//                         4 < m <= 8: one coordinate per lane (boxqp_rows.h), Φ = QuuF^-1 from 8 unit-vector solves, K = -Φ Qux_reg and
//                         Y = Quu K + 2 Qux as 4 NTR more products
//   Vxx_i = Qxx + ½(K'Y + Y'K)    P = Qxx + K'Y on the xx tiles (NTR² products per k-step), then ½(P + P') exactly (:69-72)
// Round 5 (profiles/r05_mid_phases.txt): every global load of a step in ONE straight-line batch behind the second product, consumed at the
// top of the next step; Vxx_{i+1} stored from its LDS image just in front of that batch; no exec-mask branch per element anywhere.
// Operands with run-time strides (one instantiation for the LTI / LTV / TV-cost methods).  Accumulator layout of the instruction:
// register r of lane (l4, l15) holds row 4r + l4, column l15 of a tile; A operand lane = A[i = l15][k = l4], B = B[k = l4][j = l15].
#include "ddp_internal.h"
#include "boxqp_dev.h"
#include "boxqp_rows.h"

// -DMID_PROF: s_memtime ticks per phase of a step, summed over the launch by trajectory 0 and left in the first words of its Vxx
// (profiles/mid_phase_profile.py); a profiling build only
#ifdef MID_PROF
#define MP_DECL unsigned long long mp_t = __builtin_amdgcn_s_memtime(), mp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); mp_acc[k] += t_ - mp_t; mp_t = t_; }
#define MP_DUMP if (b == 0 && lane == 0) { for (int k = 0; k < 12; ++k) Vxxg[k] = (double)mp_acc[k]; }
#else
#define MP_DECL
#define MP(k)
#define MP_DUMP
#endif

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

struct BPMidArgs {
    int n, m, N, B, regType;
    long fx_t, fx_b, fu_t, fu_b, cxx_t, cxx_b, cxu_t, cxu_b, cuu_t, cuu_b;      // element strides per time step / per trajectory (0: shared)
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
};

__device__ __forceinline__ d4 mf(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }

template <int NTR, int PT, int MMX>
struct MidLds {
    static constexpr int NR = 16 * NTR, PC = 16 * PT, LDV = NR + 1, LDF = NR + 1, LDW = PC + 1, MM = MMX, MK = 8;      // MK: k length of the rank update (two k-steps)
    // the W image is dead behind the second product: K, ½Y and the unsymmetrised Vxx_i live there
    static constexpr int WSZ = NR * LDW > 2 * MK * NR + NR * LDV ? NR * LDW : 2 * MK * NR + NR * LDV;
    static constexpr int oV = 0, oF = oV + NR * LDV, oW = oF + LDF * PC, oGu = oW + WSZ, oQx = oGu + (MK + 1) * PC, oVx = oQx + NR,   // (row MK of Gu: a dump row)
                         oQuu = oVx + NR, oRs = oQuu + MK * MK, oRi = oRs + MK * MK, oSink = oRi + 2 * MK, oGv = oSink + DDP_WAVE, oZero = oGv + DDP_WAVE, oUv = oZero + PC, oTot = oUv + 8;
    static constexpr int oK = oW, oY = oW + MK * NR, oVr = oW + 2 * MK * NR;
};

// MMX: the size the m x m system is compiled for (4 for m <= 4: a quarter of the registers of the 8 x 8 arrays)
template <int NTR, int PT, int MMX, bool LIMS, bool CTV>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_mid_kernel(BPMidArgs a)
{
    using L = MidLds<NTR, PT, MMX>;
    constexpr int NR = L::NR, PC = L::PC, LDV = L::LDV, LDF = L::LDF, LDW = L::LDW, MM = L::MM, MK = L::MK, KT = NR / 4;
    const int b = blockIdx.x, lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    if (a.active && a.active[b] == 0) return;
    const int n = a.n, m = a.m, N = a.N, p = n + m;
    extern __shared__ double lds[];
    double *Vs = lds + L::oV, *Fs = lds + L::oF, *Ws = lds + L::oW, *Gu = lds + L::oGu, *qxs = lds + L::oQx, *vxs = lds + L::oVx,
           *Hs = lds + L::oQuu, *Rs = lds + L::oRs, *ris = lds + L::oRi, *Ks = lds + L::oK, *Ys = lds + L::oY, *Vr = lds + L::oVr;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    const double *ug = LIMS ? a.u + (size_t)m * N * b : nullptr;
    const double *fx = a.fx + a.fx_b * b, *fu = a.fu + a.fu_b * b;
    const double *cxx = a.cxx + a.cxx_b * b, *cxu = a.cxu + a.cxu_b * b, *cuu = a.cuu + a.cuu_b * b;
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + mm * N * b,
           *Vxg = a.Vx + (size_t)n * N * b, *Vxxg = a.Vxx + nn * N * b;
    const double lam = a.lambda[b];
    const int regType = a.regType;
    bool nolims = true;
    double limlo[MM], limhi[MM];
#pragma unroll
    for (int q = 0; q < MM; ++q) { limlo[q] = -1.0; limhi[q] = 1.0; }
    if (LIMS) {
        nolims = a.lims[0] > a.lims[m];                             // backward_pass.jl:31
#pragma unroll
        for (int q = 0; q < MM; ++q) if (q < m) { limlo[q] = a.lims[q]; limhi[q] = a.lims[q + m]; }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35

    for (int e = lane; e < L::oTot; e += DDP_WAVE) lds[e] = 0.0;    // the padding stays zero for the whole launch
    wave_sync();
    {   // terminal step (backward_pass.jl:21-23 / :197-199 / :234-236)
        const size_t tl = (size_t)(N - 1);
        for (int e = lane; e < (int)nn; e += DDP_WAVE) { const double v = cxx[a.cxx_t * tl + e]; Vs[(e % n) + LDV * (e / n)] = v; Vxxg[nn * tl + e] = v; }
        if (lane < n) { const double v = cx[(size_t)n * tl + lane]; vxs[lane] = v; Vxg[(size_t)n * tl + lane] = v; }
        if (lane < (int)mm) Quug[mm * tl + lane] = cuu[a.cuu_t * tl + lane];
        for (int e = lane; e < (int)nm; e += DDP_WAVE) Kg[nm * tl + e] = 0.0;
        if (lane < m) kg[(size_t)m * tl + lane] = 0.0;
    }
    double dV0 = 0.0, dV1 = 0.0, kprev[MM], kprev8 = 0.0;
    const double lim_lo = (LIMS && (l15 & 7) < m) ? a.lims[l15 & 7] : -1.0, lim_hi = (LIMS && (l15 & 7) < m) ? a.lims[(l15 & 7) + m] : 1.0;
    int k_src[4];                                                 // element e of K_i (memory order, m x n): its place in the LDS image Ks[q][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int e = lane + DDP_WAVE * r, ee = e < (int)nm ? e : 0; k_src[r] = (ee % m) * NR + ee / m; }
#pragma unroll
    for (int q = 0; q < MM; ++q) kprev[q] = 0.0;
    int diverge = 0;
    // ---- F = [fx fu] by columns: lane (kr, jh) = (lane % NR, lane / NR) moves row kr of the columns CPI r + jh, r = 0, 1, ...: a scalar base +
    // one 32-bit offset per load (rows / columns past the end repeat the last one), the LDS address one per-lane base + an immediate.
    // EVERY global load of the time loop sits in ONE straight-line batch (F, the gradients, time-varying cost terms of step i - 1, behind the
    // second product of step i) and is consumed at the top of the next step: with loads under run-time conditions (the element-linear
    // loop of the first version, the per-row tests of the gradient loads) the compiler could not count what was outstanding and waited for
    // EVERYTHING (s_waitcnt vmcnt(0)) right behind the requests it had just issued — profiles/r05_mid_phases.txt: 7 750 of a step's 22 000
    // cycles in the phase that holds 24 matrix instructions.  Columns n.. of the fx part land on columns the fu part overwrites or nobody reads.
    constexpr int CPI = DDP_WAVE / NR, RA = NR / CPI, RB = 8 / CPI, RS = (NR * NR + DDP_WAVE - 1) / DDP_WAVE;
    const int kr = lane % NR, jh = lane / NR, krc = kr < n ? kr : n - 1;
    unsigned gA[RA], gB[RB];
    int lB[RB];
#pragma unroll
    for (int r = 0; r < RA; ++r) { const int j = CPI * r + jh; gA[r] = 8u * (unsigned)(krc + n * (j < n ? j : n - 1)); }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int q = CPI * r + jh;
        gB[r] = 8u * (unsigned)(krc + n * (q < m ? q : m - 1));
        lB[r] = q < m ? L::oF + kr + LDF * (n + q) : L::oSink + lane;
    }
    double *const FsA = Fs + kr + LDF * jh;
    // the gradient [cx_i; cu_i] through an LDS vector (gv; zeros past n + m): one load per lane and step
    const char *gptr = lane < n ? (const char *)(cx + lane) : (lane < p ? (const char *)(cu + (lane - n)) : (const char *)cx);
    const unsigned gstride = 8u * (unsigned)(lane < n ? n : (lane < p ? m : 0));
    double *const gv = lds + L::oGv;
    int s_a[RS];                                                  // element e of Vxx_i (memory order): its offset in the LDS image
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int e = lane + DDP_WAVE * r, ee = e < (int)nn ? e : (int)nn - 1;
        s_a[r] = (ee % n) + LDV * (ee / n);
    }
    const int quu_src = lane < (int)mm ? (lane % m) * PC + n + lane / m : 0;      // (no division by a run-time m inside the loop)
    double pfA[RA], pfB[RB], pg, pu = 0.0;                        // F, the gradients (and, with limits, u) of the next step, requested a step ahead
    double *const uvec = lds + L::oUv;
    auto load_F = [&](int i) {
        const char *fxi = (const char *)(fx + a.fx_t * i), *fui = (const char *)(fu + a.fu_t * i);
#pragma unroll
        for (int r = 0; r < RA; ++r) pfA[r] = *(const double *)(fxi + gA[r]);
#pragma unroll
        for (int r = 0; r < RB; ++r) pfB[r] = *(const double *)(fui + gB[r]);
        pg = *(const double *)(gptr + (size_t)gstride * i);
        if constexpr (LIMS) pu = ug[(size_t)m * i + (lane < m ? lane : 0)];
    };
    auto store_F = [&]() {
#pragma unroll
        for (int r = 0; r < RA; ++r) FsA[LDF * CPI * r] = pfA[r];
#pragma unroll
        for (int r = 0; r < RB; ++r) lds[lB[r]] = pfB[r];
        gv[lane] = lane < p ? pg : 0.0;
        if constexpr (LIMS) { if (lane < 8) uvec[lane] = lane < m ? pu : 0.0; }
    };
    // Vxx_i from the LDS image, in memory order: all reads first (under a condition the compiler sank each read into its store's branch: 16
    // LDS round trips in a row), the lanes past the end repeat the last element (the same value to the same address: no exec mask)
    const unsigned vmax = 8u * (unsigned)(nn - 1);
    auto store_Vxx = [&](int i) {
        double vv[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) vv[r] = Vs[s_a[r]];
#pragma unroll
        for (int r = 0; r < RS; ++r) asm volatile("" : "+v"(vv[r]));
        char *const base = (char *)(Vxxg + nn * i);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            if (DDP_WAVE * r < (int)nn) {                            // (uniform)
                const unsigned off = min(8u * (unsigned)lane + 512u * r, vmax);
                *(double *)(base + off) = vv[r];
            }
        }
    };
    // where the u rows of G go (Gu, row - n) — a dump row for the rest; ½ / 0 and 1 / 0 factors of the symmetrisation (rows / columns < n)
    int gu_row[PT][4];
    double hrow[NTR][4], cmask[NTR];
#pragma unroll
    for (int ti = 0; ti < PT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = 16 * ti + 4 * r + l4; gu_row[ti][r] = (row >= n && row < p) ? (row - n) * PC : MK * PC; }
#pragma unroll
    for (int ti = 0; ti < NTR; ++ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hrow[ti][r] = (16 * ti + 4 * r + l4 < n) ? 0.5 : 0.0;
        cmask[ti] = (16 * ti + l15 < n) ? 1.0 : 0.0;
    }
    // cost Hessians of this lane's tile elements: registers while they do not vary with time
    double hc[PT][PT][4];
    auto load_H = [&](int i) {
        const double *cxxi = cxx + a.cxx_t * i, *cxui = cxu + a.cxu_t * i, *cuui = cuu + a.cuu_t * i;
#pragma unroll
        for (int ti = 0; ti < PT; ++ti)
#pragma unroll
            for (int cj = 0; cj < PT; ++cj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + l4, col = 16 * cj + l15;
                    double c = 0.0;
                    if (row < n) { if (col < n) c = cxxi[row + n * col]; else if (col < p) c = cxui[row + n * (col - n)]; }       // (:244), Qxu = Qux'
                    else if (row < p) { if (col < n) c = cxui[col + n * (row - n)]; else if (col < p) c = cuui[(row - n) + m * (col - n)]; }   // (:242-243)
                    hc[ti][cj][r] = c;
                }
    };
    // the gradient column (Qx, Qu: column p of G): the lanes l15 == p % 16 of tile column p / 16 add gv[row]; the others read zeros
    const double *const gsel = (l15 == p % 16 ? gv : lds + L::oZero) + l4;
    if (N >= 2) { load_F(N - 2); load_H(CTV ? N - 2 : 0); }
    wave_sync();
    MP_DECL
    for (int i = N - 2; i >= 0; --i) {
        MP(7)
        // ---- F_i = [fx fu] into the LDS (k fastest; the memory order of both arrays); the next one is requested behind the products
        store_F();
        wave_sync();
        MP(0)
        // ================= W = Vxx F ==============================================================================
        {
            d4 acc[NTR][PT];
#pragma unroll
            for (int ri = 0; ri < NTR; ++ri)
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) acc[ri][cj] = d4{0.0, 0.0, 0.0, 0.0};
            // the operands of k-step ks + 1 are read (unconditionally: past n they are zeros) before the products of k-step ks, which a
            // uniform test skips when 4 ks >= n — with the reads inside the test every k-step waited for its own LDS round trip
            double av[2][NTR], bv[2][PT];
            auto rdW = [&](int ks, int sl) __attribute__((always_inline)) {
#pragma unroll
                for (int ri = 0; ri < NTR; ++ri) av[sl][ri] = Vs[(16 * ri + l15) + LDV * (4 * ks + l4)];
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) bv[sl][cj] = Fs[(4 * ks + l4) + LDF * (16 * cj + l15)];
            };
            rdW(0, 0);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                if (ks + 1 < KT) rdW(ks + 1, (ks + 1) & 1);
                if (4 * ks < n) {
#pragma unroll
                    for (int ri = 0; ri < NTR; ++ri)
#pragma unroll
                        for (int cj = 0; cj < PT; ++cj) acc[ri][cj] = mf(av[ks & 1][ri], bv[ks & 1][cj], acc[ri][cj]);
                }
            }
#pragma unroll
            for (int ri = 0; ri < NTR; ++ri)
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) {
                    double *wp = Ws + (16 * ri + l4) * LDW + 16 * cj + l15;
                    wp[0] = acc[ri][cj].x; wp[4 * LDW] = acc[ri][cj].y; wp[8 * LDW] = acc[ri][cj].z; wp[12 * LDW] = acc[ri][cj].w;
                }
        }
        if (lane < n) Ws[lane * LDW + p] = vxs[lane];              // column p of [W | Vx] (behind the tile stores: LDS operations of a wave stay in order)
        wave_sync();
        MP(1)
        // ================= G = F'[W | Vx] + cost terms ================================================================
        d4 g[PT][PT];
#pragma unroll
        for (int ti = 0; ti < PT; ++ti)
#pragma unroll
            for (int cj = 0; cj < PT; ++cj) g[ti][cj] = d4{0.0, 0.0, 0.0, 0.0};
        {
            double fa[2][PT], wb[2][PT];
            auto rdG = [&](int ks, int sl) __attribute__((always_inline)) {
#pragma unroll
                for (int ti = 0; ti < PT; ++ti) fa[sl][ti] = Fs[(4 * ks + l4) + LDF * (16 * ti + l15)];      // A[i][k] = F[k, 16 ti + i]
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) wb[sl][cj] = Ws[(4 * ks + l4) * LDW + 16 * cj + l15];        // B[k][j] = W[k, 16 cj + j]
            };
            rdG(0, 0);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                if (ks + 1 < KT) rdG(ks + 1, (ks + 1) & 1);
                if (4 * ks < n) {
#pragma unroll
                    for (int ti = 0; ti < PT; ++ti)
#pragma unroll
                        for (int cj = 0; cj < PT; ++cj) g[ti][cj] = mf(fa[ks & 1][ti], wb[ks & 1][cj], g[ti][cj]);
                }
            }
        }
        MP(8)
        {   // cost Hessians and gradients; the u rows leave for the gains (Gu; other rows aim at its dump row), the tile column that holds
            // column p goes to a free region whole: Qx_j = Gc[j][p % 16] — no per-element test, no exec-mask branch
            const int cjp = p / 16;
            double *const Gc = Vr + l4 * 17 + l15;                     // (the region of the unsymmetrised Vxx_i: free until the value update)
#pragma unroll
            for (int ti = 0; ti < PT; ++ti) {
                const bool urows = 16 * ti + 16 > n && 16 * ti < p;                    // (uniform)
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) {
                    double v[4] = {g[ti][cj].x, g[ti][cj].y, g[ti][cj].z, g[ti][cj].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += hc[ti][cj][r];
                    if (cj == cjp) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += gsel[16 * ti + 4 * r];       // (zeros outside the lanes of column p; hc is zero in them)
                        if (ti < NTR) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) Gc[(16 * ti + 4 * r) * 17] = v[r];
                        }
                    }
                    if (urows) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Gu[gu_row[ti][r] + 16 * cj + l15] = v[r];
                    }
                    g[ti][cj] = d4{v[0], v[1], v[2], v[3]};
                }
            }
        }
        // Vxx_{i+1} leaves from the image it is still in (its stores have the rest of the step to land before anything waits on the
        // memory counter), then the one batch of loads for step i - 1 (step 0 asks for itself once more: no branch around the requests)
        MP(9)
        store_Vxx(i + 1);
        MP(10)
        load_F(i > 0 ? i - 1 : 0);
        if (CTV) load_H(i > 0 ? i - 1 : 0);
        wave_sync();
        MP(2)
        // ================= gains (backward_pass.jl:30-62) ==============================================================================
        if constexpr (MM == 8) {
        // ---- 4 < m <= 8: the m x m system with ONE COORDINATE PER LANE (boxqp_rows.h: lane i of every 16-lane row holds row / column i of
        // QuuF and of its factor; products, factorisation and solves are runs of v_fmac_f64_dpp row_newbcast), identity past m; the gains
        // K = -Φ Qux_reg and Y = Quu K + 2 Qux on the matrix cores with Φ = QuuF^-1 (masked by the clamped set) from 8 unit-vector solves,
        // four per pass (one per 16-lane row).  The first form let every lane repeat the 8 x 8 system on per-lane arrays: 500-1 500 bytes
        // of scratch per lane, 9 - 13 000 cycles for the gains and as many for K, Y of a step (profiles/r05_mid_phases.txt).
        double *const Ph = Rs, *const kv8 = ris, *const wv8 = ris + 8;
        const int ci = l15 & 7;
        const bool in8 = l15 < 8;
        {   // QuuF (:247): one element per lane, identity past m
            const int r2 = lane & 7, c2 = lane >> 3;
            double v = r2 == c2 ? 1.0 : 0.0;
            if (r2 < m && c2 < m) {
                v = Gu[r2 * PC + n + c2];
                if (regType == 2) {
                    double sv = 0.0;
                    for (int k2 = 0; k2 < n; ++k2) sv += Fs[k2 + LDF * (n + r2)] * Fs[k2 + LDF * (n + c2)];
                    v += lam * sv;
                } else if (r2 == c2) v += lam;
            }
            Hs[lane] = v;
        }
        const double *Qxr = Gu;                                   // Qux_reg, rows q < 8 with pitch PC (:246)
        if (regType == 2) {                                       // Vxx_reg = Vxx + λI: λ fu'fx on Qux_reg — its own image
            double *const Gr = lds + L::oTot;                          // (8 x PC doubles the launcher adds to the allocation for regType 2)
            const int lcq = lane < n ? lane : 0;
            for (int q = 0; q < m; ++q) {
                double sx = 0.0;
                for (int k2 = 0; k2 < n; ++k2) sx += Fs[k2 + LDF * (n + q)] * Fs[k2 + LDF * lcq];
                if (lane < NR) Gr[q * PC + lane] = lane < n ? Gu[q * PC + lane] + lam * sx : 0.0;
            }
            for (int q = m; q < 8; ++q) if (lane < NR) Gr[q * PC + lane] = 0.0;
            Qxr = Gr;
        }
        wave_sync();
        bqr::Rows<8> qr;
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) { qr.Hrow[j2] = Hs[ci + 8 * j2]; qr.Hcol[j2] = Hs[j2 + 8 * ci]; }
        const double Qui = ci < m ? Gu[ci * PC + p] : 0.0;          // Qu_i
        double quurow[8];                                         // Quu[i, :] (unregularised; zeros past m)
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) quurow[j2] = Gu[ci * PC + n + j2];
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) quurow[j2] = (ci < m && j2 < m) ? quurow[j2] : 0.0;
        unsigned clamped = 0u;
        int fail;
        double ki;
        if (!LIMS || nolims) {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) { qr.Rcol[k2] = 0.0; qr.Rrow[k2] = 0.0; qr.ri[k2] = 0.0; }
            fail = bqr::factor<8>(qr, 0u, l15, in8);                 // cholesky(Hermitian(QuuF))  (:35)
            ki = -bqr::solve<8>(qr, in8 ? Qui : 0.0, l15);          // k_i = -(R\(R'\Qu))  (:41)
        } else {
            // coordinates past m: gradient 1 on the interval [0, 0] — clamped in every iteration (x == lower, grad > 0)
            const double uq = uvec[ci];
            const double lo = ci < m ? lim_lo - uq : 0.0, up = ci < m ? lim_hi - uq : 0.0, gq8 = ci < m ? Qui : 1.0;      // (:45-46)
            int iters;
            const int result = bqr::boxqp_rows<8>(qr, gq8, lo, up, kprev8, qpo, l15, ki, clamped, iters);            // (:49)
            fail = (result < 1);                                     // (:53)
        }
        if (lane < (int)mm) Quug[mm * i + lane] = Gu[quu_src];        // assigned before a failure upstream too
        if (fail) { diverge = i + 1; break; }                        // (:37-38, :54-55): wave-uniform
        MP(3)
        kprev8 = ki;
        {   // Φ = (masked QuuF)^-1 without the clamped rows / columns: column c = 4 pass + (lane >> 4) per 16-lane row
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int c = 4 * ps + l4;
                const double y = bqr::solve<8>(qr, (in8 && l15 == c && !((clamped >> c) & 1u)) ? 1.0 : 0.0, l15);
                if (in8) Ph[l15 + 8 * c] = ((clamped >> l15) & 1u) ? 0.0 : y;
            }
        }
        double one = 1.0;
        asm volatile("" : "+v"(one));
        const double Quuki = bqr::rowdot<8>(in8 ? ki : 0.0, quurow);       // (Quu k)_i  (:64)
        dV0 += bqr::rowsum<8>(in8 ? ki * Qui : 0.0, one);                   // (:68)
        dV1 += 0.5 * bqr::rowsum<8>(in8 ? ki * Quuki : 0.0, one);
        if (lane < 8) { kv8[lane] = ki; wv8[lane] = Quuki + Qui; }
        if (lane < m) kg[(size_t)m * i + lane] = ki;                                   // (:75)
        wave_sync();
        {   // K = -Φ Qux_reg, Y = Quu K + 2 Qux: rows 4 r + l4 < 8 of the accumulators (r = 0, 1); K in the accumulator layout IS the B operand of Y
            const double *const phA = in8 ? Ph + l15 : lds + L::oZero;      // A[i = l15][k]: Φ[l15][k], zero rows past 8
            const int phs = in8 ? 8 : 0;
#pragma unroll
            for (int tj = 0; tj < NTR; ++tj) {
                d4 kacc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    kacc = mf(phA[phs * (4 * ks + l4)], Qxr[(4 * ks + l4) * PC + 16 * tj + l15], kacc);
                const double k0 = -kacc.x, k1 = -kacc.y;            // K[l4][col], K[4 + l4][col]
                d4 yacc = d4{2.0 * Gu[l4 * PC + 16 * tj + l15], 2.0 * Gu[(4 + l4) * PC + 16 * tj + l15], 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const double qa = Gu[(in8 ? l15 : 0) * PC + n + 4 * ks + l4];
                    yacc = mf((in8 && 4 * ks + l4 < m) ? qa : 0.0, ks == 0 ? k0 : k1, yacc);
                }
                Ks[l4 * NR + 16 * tj + l15] = k0; Ks[(4 + l4) * NR + 16 * tj + l15] = k1;
                Ys[l4 * NR + 16 * tj + l15] = yacc.x; Ys[(4 + l4) * NR + 16 * tj + l15] = yacc.y;
            }
        }
        wave_sync();
        {   // Vx_i (:69) and K_i in memory order (:76)
            const int lc = lane < n ? lane : n - 1;
            double vx = Vr[lc * 17 + (p & 15)];                      // Qx_j
#pragma unroll
            for (int q = 0; q < 8; ++q) vx += Ks[q * NR + lc] * wv8[q] + Gu[q * PC + lc] * kv8[q];
            if (lane < n) { Vxg[(size_t)n * i + lane] = vx; vxs[lane] = vx; }
            double kvv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) kvv[r] = Ks[k_src[r]];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (lane + DDP_WAVE * r < (int)nm) Kg[nm * i + lane + DDP_WAVE * r] = kvv[r];
        }
        // (columns n.. of the K, Y images carry finite products of the u columns of Gu: the value update's rows / columns past n are zeroed by its factors)
        } else {
        // ---- m <= 4: every lane the (identity-padded) 4 x 4 system on registers (boxqp_dev.h)
        double H[MM * MM], R[MM * MM], Qu[MM], kk[MM], ri[MM];
        double Quu[MM * MM];                                       // every read of the step issued at once, zeros past m
#pragma unroll
        for (int e = 0; e < MM * MM; ++e) Quu[e] = Gu[(e % MM) * PC + n + e / MM];
#pragma unroll
        for (int e = 0; e < MM * MM; ++e) Quu[e] = (e % MM < m && e / MM < m) ? Quu[e] : 0.0;
        auto quu = [&](int q, int q2) __attribute__((always_inline)) -> double { return Quu[q + MM * q2]; };
        unsigned clamped = 0u;
#pragma unroll
        for (int c2 = 0; c2 < MM; ++c2) Qu[c2] = c2 < m ? Gu[c2 * PC + p] : 0.0;
        const int lc = lane < n ? lane : n - 1;                    // (lanes past n repeat column n - 1 and drop the result: no branch around the reads)
        double xr[MM];                                            // Qux_reg[:, lane]; rows q >= m of Gu are never written: zeros
#pragma unroll
        for (int q = 0; q < MM; ++q) xr[q] = Gu[q * PC + lc];
        if (regType == 2) {                                       // Vxx_reg = Vxx + λI: λ fu'fx on Qux_reg (:246)
#pragma unroll
            for (int q = 0; q < MM; ++q) {
                if (q < m) {
                    double sx = 0.0;
                    for (int k2 = 0; k2 < n; ++k2) sx += Fs[k2 + LDF * (n + q)] * Fs[k2 + LDF * (lane < n ? lane : 0)];
                    if (lane < n) xr[q] += lam * sx;
                }
            }
        }
        int fail;
        auto solve = [&](double (&bv)[MM]) __attribute__((always_inline)) { chol_solve_ri<MM>(MM, R, ri, bv); };      // b <- (R'R)\b
        {
#pragma unroll
            for (int c2 = 0; c2 < MM; ++c2)
#pragma unroll
                for (int r2 = 0; r2 < MM; ++r2) H[r2 + MM * c2] = (r2 < m && c2 < m) ? quu(r2, c2) : (r2 == c2 ? 1.0 : 0.0);   // identity past m: the system is solved at its compiled size, no run-time bounds in the factorisation and the solves
            if (regType == 2) {                                   // λ fu'fu on QuuF (:247)
#pragma unroll
                for (int q = 0; q < MM; ++q)
#pragma unroll
                    for (int q2 = 0; q2 < MM; ++q2) {
                        if (q < m && q2 < m) {
                            double sv = 0.0;
                            for (int k2 = 0; k2 < n; ++k2) sv += Fs[k2 + LDF * (n + q)] * Fs[k2 + LDF * (n + q2)];
                            H[q + MM * q2] += lam * sv;
                        }
                    }
            } else {
#pragma unroll
                for (int q = 0; q < MM; ++q) H[q + MM * q] += (q < m) ? lam : 0.0;
            }
            if (!LIMS || nolims) {
                fail = chol_masked_ri<MM>(MM, H, 0u, R, ri);           // cholesky(Hermitian(QuuF))  (:35)
#pragma unroll
                for (int q = 0; q < MM; ++q) kk[q] = Qu[q];
                solve(kk);
#pragma unroll
                for (int q = 0; q < MM; ++q) kk[q] = -kk[q];         // k_i = -(R\Qu)  (:41)
            } else {
                // coordinates past m: gradient 1 on the interval [0, 0] — clamped in every iteration (x == lower, grad > 0), so they add
                // nothing to any sum and the all-clamped exit (:98) still means "every control"
                double lo[MM], up[MM], gqp[MM];
#pragma unroll
                for (int q = 0; q < MM; ++q) {
                    const double uq = uvec[q];                      // (u_i came with the step's batch of loads; zeros past m)
                    lo[q] = q < m ? limlo[q] - uq : 0.0; up[q] = q < m ? limhi[q] - uq : 0.0; gqp[q] = q < m ? Qu[q] : 1.0;      // (:45-46)
                }
                int iters;
                const int result = boxqp_dev_ri<MM>(MM, H, gqp, lo, up, kprev, qpo, kk, R, ri, clamped, iters);    // (:49), warm start k[:, min(i+1, N-1)]
                fail = (result < 1);                                 // (:53)
            }
        }
        if (lane < (int)mm) Quug[mm * i + lane] = Gu[quu_src];     // assigned before a failure upstream too
        if (fail) { diverge = i + 1; break; }                        // (:37-38, :54-55): wave-uniform
        MP(3)
        double Quuk[MM];
#pragma unroll
        for (int q = 0; q < MM; ++q) {
            double t = 0.0;
#pragma unroll
            for (int q2 = 0; q2 < MM; ++q2) t += quu(q, q2) * kk[q2];
            Quuk[q] = t;                                             // (:64)
            kprev[q] = kk[q];
        }
        {
            double kQu = 0.0, kQuuk = 0.0;
#pragma unroll
            for (int q = 0; q < MM; ++q) { kQu += kk[q] * Qu[q]; kQuuk += kk[q] * Quuk[q]; }
            dV0 += kQu; dV1 += 0.5 * kQuuk;                          // (:68)
        }
        {   // K_i column `lane`, Y = Quu K + 2 Qux, Vx_i  (:42 / :57-61, :69)
            double col[MM], x2[MM], yy[MM];
#pragma unroll
            for (int q = 0; q < MM; ++q) { x2[q] = Gu[q * PC + lc]; col[q] = ((clamped >> q) & 1u) ? 0.0 : xr[q]; }
            double vx = Vr[lc * 17 + (p & 15)];                      // Qx_j from the tile column of column p (the epilogue above)
            solve(col);
#pragma unroll
            for (int q = 0; q < MM; ++q) col[q] = (((clamped >> q) & 1u) || q >= m) ? 0.0 : -col[q];
#pragma unroll
            for (int q = 0; q < MM; ++q) {
                double t = 2.0 * x2[q];
#pragma unroll
                for (int q2 = 0; q2 < MM; ++q2) t += quu(q, q2) * col[q2];
                vx += col[q] * (Quuk[q] + Qu[q]) + x2[q] * kk[q];
                yy[q] = t;
            }
            const bool ln = lane < n;
            if (lane < NR) {
#pragma unroll
                for (int q = 0; q < MM; ++q) { Ks[q * NR + lane] = ln ? col[q] : 0.0; Ys[q * NR + lane] = ln ? yy[q] : 0.0; }   // (rows q >= m: zeros)
            }
            if (ln) {
#pragma unroll
                for (int q = 0; q < MM; ++q)
                    if (q < m) Kg[nm * i + q + (size_t)m * lane] = col[q];       // (:76)
                Vxg[(size_t)n * i + lane] = vx; vxs[lane] = vx;
            }
            if (lane < m) {
                double kv = kk[0];
#pragma unroll
                for (int q = 1; q < MM; ++q) kv = (lane == q) ? kk[q] : kv;
                kg[(size_t)m * i + lane] = kv;                                           // (:75)
            }
        }
        }
        wave_sync();
        MP(4)
        // ================= Vxx_i = Qxx + ½(K'Y + Y'K)  (:70-72): P = Qxx + K'Y on the xx tiles, then ½(P + P') — the symmetric part of
        // K'Y is ½(K'Y + Y'K), so ONE product per tile and k-step instead of two ============================================
#pragma unroll
        for (int ks = 0; ks < MM / 4; ++ks) {                           // (rows q >= MM of K, Y do not exist: MM / 4 k-steps)
            double ka[NTR], ya[NTR];
#pragma unroll
            for (int t = 0; t < NTR; ++t) { ka[t] = Ks[(4 * ks + l4) * NR + 16 * t + l15]; ya[t] = Ys[(4 * ks + l4) * NR + 16 * t + l15]; }
#pragma unroll
            for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                for (int tj = 0; tj < NTR; ++tj) g[ti][tj] = mf(ka[ti], ya[tj], g[ti][tj]);      // K'Y: its symmetric part is ½(K'Y + Y'K)
        }
        wave_sync();                                                   // K, Y have been read: the region takes the unsymmetrised Vxx_i
#pragma unroll
        for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
            for (int tj = 0; tj < NTR; ++tj) {
                double *vp = Vr + (16 * ti + l4) + LDV * (16 * tj + l15);
                vp[0] = g[ti][tj].x; vp[4] = g[ti][tj].y; vp[8] = g[ti][tj].z; vp[12] = g[ti][tj].w;
            }
        wave_sync();
        MP(5)
        // ½(P + P') in the accumulator layout: the transposed element of (row, col) = (16 ti + 4 r + l4, 16 tj + l15) is one LDS read at a
        // per-lane base + an immediate; rows / columns past n become exact zeros by their ½ / 0 and 1 / 0 factors (the image stays padded)
        {
            const double *tp = Vr + l15 + LDV * l4;
#pragma unroll
            for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                for (int tj = 0; tj < NTR; ++tj) {
                    const double own[4] = {g[ti][tj].x, g[ti][tj].y, g[ti][tj].z, g[ti][tj].w};
                    double *vp = Vs + (16 * ti + l4) + LDV * (16 * tj + l15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) vp[4 * r] = ((own[r] + tp[16 * tj + LDV * (16 * ti + 4 * r)]) * hrow[ti][r]) * cmask[tj];
                }
        }
        wave_sync();
        MP(6)
    }
    if (!diverge) store_Vxx(0);                                       // (the last image; the terminal one when N = 1)
    MP_DUMP
    if (diverge) {                                                      // outputs earlier in time than a failing step are zero (:37-38 with :226-229)
        const size_t ie = (size_t)diverge;
        for (size_t e = lane; e < nm * ie; e += DDP_WAVE) Kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)m * ie; e += DDP_WAVE) kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)n * ie; e += DDP_WAVE) Vxg[e] = 0.0;
        for (size_t e = lane; e < nn * ie; e += DDP_WAVE) Vxxg[e] = 0.0;
        for (size_t e = lane; e < mm * (ie - 1); e += DDP_WAVE) Quug[e] = 0.0;
    }
    if (lane == 0) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
}

template <int NTR, int PT, int MMX, bool LIMS, bool CTV>
int launch_mid2(ddp_handle h, const ddp_bp_desc *d, const BPMidArgs &a)
{
    const size_t bytes = ((size_t)MidLds<NTR, PT, MMX>::oTot + ((MMX == 8 && a.regType == 2) ? 8 * MidLds<NTR, PT, MMX>::PC : 0)) * sizeof(double);
    const dim3 grid((unsigned)d->B), block(DDP_WAVE);
    DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mid_kernel<NTR, PT, MMX, LIMS, CTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((back_pass_mid_kernel<NTR, PT, MMX, LIMS, CTV>), grid, block, bytes, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}

template <int NTR, int PT, int MMX>
int launch_mid(ddp_handle h, const ddp_bp_desc *d, const BPMidArgs &a)
{
    const bool ctv = a.cxx_t != 0 || a.cxu_t != 0 || a.cuu_t != 0;       // cost terms that vary with time: requested per step with the batch
    if (d->has_lims) return ctv ? launch_mid2<NTR, PT, MMX, true, true>(h, d, a) : launch_mid2<NTR, PT, MMX, true, false>(h, d, a);
    return ctv ? launch_mid2<NTR, PT, MMX, false, true>(h, d, a) : launch_mid2<NTR, PT, MMX, false, false>(h, d, a);
}

}   // namespace

// returns 1 if the shape is not handled here (n > 32, m > 8, n + m + 1 > 48), 0 launched, < 0 error
int ddp_launch_back_pass_mid(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge)
{
    const int n = d->n, m = d->m;
    if (n < 1 || m < 1 || n > 32 || m > 8) return 1;
    const long N = d->N;
    BPMidArgs a;
    a.n = n; a.m = m; a.N = d->N; a.B = d->B; a.regType = d->regType;
    const long nn = (long)n * n, nm = (long)n * m, mm = (long)m * m;
    a.fx_t = d->fx_tv ? nn : 0; a.fx_b = d->fx_batched ? nn * (d->fx_tv ? N : 1) : 0;
    a.fu_t = d->fx_tv ? nm : 0; a.fu_b = d->fx_batched ? nm * (d->fx_tv ? N : 1) : 0;
    a.cxx_t = d->cost_tv ? nn : 0; a.cxx_b = d->cost_batched ? nn * (d->cost_tv ? N : 1) : 0;
    a.cxu_t = d->cost_tv ? nm : 0; a.cxu_b = d->cost_batched ? nm * (d->cost_tv ? N : 1) : 0;
    a.cuu_t = d->cost_tv ? mm : 0; a.cuu_b = d->cost_batched ? mm * (d->cost_tv ? N : 1) : 0;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.lims = lims;
    a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    const int ntr = n <= 16 ? 1 : 2, pt = (n + m + 1 + 15) / 16;           // pt = 1 only for n + m <= 15: the row kernels' range, padded to 2 here
    if (m <= 4) {
        if (ntr == 1) return launch_mid<1, 2, 4>(h, d, a);
        return pt <= 2 ? launch_mid<2, 2, 4>(h, d, a) : launch_mid<2, 3, 4>(h, d, a);
    }
    if (ntr == 1) return launch_mid<1, 2, 8>(h, d, a);
    return pt <= 2 ? launch_mid<2, 2, 8>(h, d, a) : launch_mid<2, 3, 8>(h, d, a);
}
