// back_pass_mid8.hip — the 8 x 8 instantiations (4 < m <= 8) of the mid-size backward kernel in their first form: element-linear staging of
// [fx fu], the cost terms and gradients requested per row, every lane repeating the m x m factorisation on an LDS image.  The 4 x 4
// instantiations moved on (back_pass_mid.hip: one straight-line batch of loads per step, no exec-mask branches around the staging, one
// product for the value update); carried over unchanged here, those changes made THIS size slower (n = 32, m = 8, N = 300, B = 1 024:
// 8.9 -> 11.5 ms) — its 8 x 8 per-lane arrays already spill, and every reload of a spilled register waits on the same counter as the
// global loads.  What this size needs is the m x m system with one coordinate per lane (boxqp_rows.h) and K = -Φ Qux on the matrix cores.
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
//   gains                 every lane factorises QuuF (run-time-sized routines of boxqp_dev.h, as the vector kernel), lane c solves column c of K
//   Vxx_i = Qxx + ½(K'Y + Y'K), Y = Quu K + 2 Qux     rank-2m update on the xx tiles: 4 NTR² more products (:69-72), then ½(V + V') exactly
// Operands with run-time strides (one instantiation for the LTI / LTV / TV-cost methods).  Accumulator layout of the instruction:
// register r of lane (l4, l15) holds row 4r + l4, column l15 of a tile; A operand lane = A[i = l15][k = l4], B = B[k = l4][j = l15].
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

struct BPMid8Args {
    int n, m, N, B, regType;
    long fx_t, fx_b, fu_t, fu_b, cxx_t, cxx_b, cxu_t, cxu_b, cuu_t, cuu_b;      // element strides per time step / per trajectory (0: shared)
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
};

__device__ __forceinline__ d4 mf(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }

template <int NTR, int PT, int MMX>
struct Mid8Lds {
    static constexpr int NR = 16 * NTR, PC = 16 * PT, LDV = NR + 1, LDF = NR + 1, LDW = PC + 1, MM = MMX, MK = 8;      // MK: k length of the rank update (two k-steps)
    // the W image is dead behind the second product: K, ½Y and the unsymmetrised Vxx_i live there
    static constexpr int WSZ = NR * LDW > 2 * MK * NR + NR * LDV ? NR * LDW : 2 * MK * NR + NR * LDV;
    static constexpr int oV = 0, oF = oV + NR * LDV, oW = oF + LDF * PC, oGu = oW + WSZ, oQx = oGu + MK * PC, oVx = oQx + NR,
                         oQuu = oVx + NR, oRs = oQuu + MK * MK, oRi = oRs + MK * MK, oTot = oRi + MK + 2;
    static constexpr int oK = oW, oY = oW + MK * NR, oVr = oW + 2 * MK * NR;
};

// MMX: the size the m x m system is compiled for (4 for m <= 4: a quarter of the registers of the 8 x 8 arrays)
template <int NTR, int PT, int MMX, bool LIMS>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_mid8_kernel(BPMid8Args a)
{
    using L = Mid8Lds<NTR, PT, MMX>;
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
    double dV0 = 0.0, dV1 = 0.0, kprev[MM];
#pragma unroll
    for (int q = 0; q < MM; ++q) kprev[q] = 0.0;
    int diverge = 0;
    // ---- per-lane index tables (no division inside the time loop)
    constexpr int RF = (NR * (NR + 8) + DDP_WAVE - 1) / DDP_WAVE, RS = (NR * NR + DDP_WAVE - 1) / DDP_WAVE;
    int f_lds[RF];                                                // element e = lane + 64 r of [fx fu] (contiguous in fx, then in fu): its LDS offset
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        const int e = lane + DDP_WAVE * r, ee = e < n * p ? e : 0;
        f_lds[r] = e < n * p ? (ee % n) + LDF * (ee / n) : -1;
    }
    int s_a[RS], s_b[RS];                                         // element e of Vxx_i: its LDS offset and the transposed one
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int e = lane + DDP_WAVE * r, ee = e < (int)nn ? e : 0;
        s_a[r] = (ee % n) + LDV * (ee / n); s_b[r] = (ee / n) + LDV * (ee % n);
    }
    double pfF[RF];                                               // F of the next step, requested a step ahead
    auto load_F = [&](int i) {
        const double *fxi = fx + a.fx_t * i, *fui = fu + a.fu_t * i;
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            const int e = lane + DDP_WAVE * r, ee = e < n * p ? e : 0;
            pfF[r] = ee < (int)nn ? fxi[ee] : fui[ee - (int)nn];
        }
    };
    // cost Hessians of this lane's tile elements: registers while they do not vary with time
    const bool ctv = a.cxx_t != 0 || a.cxu_t != 0 || a.cuu_t != 0;
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
    // the gradient column (Qx, Qu: column p of G): the lanes l15 == p % 16 of tile column p / 16 hold it; requested a step ahead like F
    const bool gcol = l15 == p % 16;
    double gq[PT][4];
    auto load_g = [&](int i) {
        const double *cxi = cx + (size_t)n * i, *cui = cu + (size_t)m * i;
#pragma unroll
        for (int ti = 0; ti < PT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + 4 * r + l4;
                gq[ti][r] = !gcol ? 0.0 : (row < n ? cxi[row] : (row < p ? cui[row - n] : 0.0));     // Qx (:241), Qu (:240)
            }
    };
    if (N >= 2) { load_F(N - 2); load_g(N - 2); if (!ctv) load_H(0); }
    wave_sync();
    for (int i = N - 2; i >= 0; --i) {
        // ---- F_i = [fx fu] into the LDS (k fastest; the memory order of both arrays); the next one is requested behind the products
#pragma unroll
        for (int r = 0; r < RF; ++r) if (f_lds[r] >= 0) Fs[f_lds[r]] = pfF[r];
        if (ctv) load_H(i);
        wave_sync();
        // ================= W = Vxx F ==============================================================================
        {
            d4 acc[NTR][PT];
#pragma unroll
            for (int ri = 0; ri < NTR; ++ri)
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) acc[ri][cj] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                double av[NTR], bv[PT];
#pragma unroll
                for (int ri = 0; ri < NTR; ++ri) av[ri] = Vs[(16 * ri + l15) + LDV * (4 * ks + l4)];
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) bv[cj] = Fs[(4 * ks + l4) + LDF * (16 * cj + l15)];
#pragma unroll
                for (int ri = 0; ri < NTR; ++ri)
#pragma unroll
                    for (int cj = 0; cj < PT; ++cj) acc[ri][cj] = mf(av[ri], bv[cj], acc[ri][cj]);
            }
#pragma unroll
            for (int ri = 0; ri < NTR; ++ri)
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) {
                    double *wp = Ws + (16 * ri + l4) * LDW + 16 * cj + l15;
                    wp[0] = acc[ri][cj].x; wp[4 * LDW] = acc[ri][cj].y; wp[8 * LDW] = acc[ri][cj].z; wp[12 * LDW] = acc[ri][cj].w;
                }
        }
        wave_sync();
        if (lane < n) Ws[lane * LDW + p] = vxs[lane];              // column p of [W | Vx]
        wave_sync();
        // ================= G = F'[W | Vx] + cost terms ================================================================
        d4 g[PT][PT];
#pragma unroll
        for (int ti = 0; ti < PT; ++ti)
#pragma unroll
            for (int cj = 0; cj < PT; ++cj) g[ti][cj] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            double fa[PT], wb[PT];
#pragma unroll
            for (int ti = 0; ti < PT; ++ti) fa[ti] = Fs[(4 * ks + l4) + LDF * (16 * ti + l15)];      // A[i][k] = F[k, 16 ti + i]
#pragma unroll
            for (int cj = 0; cj < PT; ++cj) wb[cj] = Ws[(4 * ks + l4) * LDW + 16 * cj + l15];        // B[k][j] = W[k, 16 cj + j]
#pragma unroll
            for (int ti = 0; ti < PT; ++ti)
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) g[ti][cj] = mf(fa[ti], wb[cj], g[ti][cj]);
        }
        if (i > 0) load_F(i - 1);                                     // (Fs has been read for the last time unless regType 2 needs it: it stays untouched)
        {   // cost Hessians and gradients; the u rows and column p leave for the gains
            const int cjp = p / 16;
#pragma unroll
            for (int ti = 0; ti < PT; ++ti)
#pragma unroll
                for (int cj = 0; cj < PT; ++cj) {
                    const int col = 16 * cj + l15;
                    double v[4] = {g[ti][cj].x, g[ti][cj].y, g[ti][cj].z, g[ti][cj].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ti + 4 * r + l4;
                        v[r] += (cj == cjp && gcol) ? gq[ti][r] : hc[ti][cj][r];
                        if (row >= n && row < p) Gu[(row - n) * PC + col] = v[r];
                        if (row < n && col == p) qxs[row] = v[r];
                    }
                    g[ti][cj] = d4{v[0], v[1], v[2], v[3]};
                }
            if (i > 0) load_g(i - 1);
        }
        wave_sync();
        // ================= gains (backward_pass.jl:30-62), every lane the m x m system ==================================
        // RL: the m x m system of the 8 x 8 instantiation WITHOUT limits lives in the LDS (Hs, Rs, ris: every lane runs the same scalar
        // factorisation on them — identical values to identical addresses — and solves its own right-hand side with broadcast reads):
        // three 64-element register arrays per lane spilled to scratch (n = 32, m = 8: 35 us per step).  The 4 x 4 instantiation and the
        // box-QP keep the register routines of boxqp_dev.h.
        constexpr bool RL = MM == 8 && !LIMS;
        double H[RL ? 1 : MM * MM], R[RL ? 1 : MM * MM], Qu[MM], kk[MM], ri[RL ? 1 : MM];      // (Quu itself stays in the LDS: Gu[q][n + q2])
        unsigned clamped = 0u;
#pragma unroll
        for (int c2 = 0; c2 < MM; ++c2) Qu[c2] = c2 < m ? Gu[c2 * PC + p] : 0.0;
        double xr[MM];                                            // Qux_reg[:, lane]
#pragma unroll
        for (int q = 0; q < MM; ++q) xr[q] = (q < m && lane < n) ? Gu[q * PC + lane] : 0.0;
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
        // forward / back substitution with the factor in registers or in the LDS; b <- (R'R)\b
        auto solve = [&](double (&bv)[MM]) __attribute__((always_inline)) {
            if constexpr (RL) {
#pragma unroll
                for (int i2 = 0; i2 < MM; ++i2) {
                    if (i2 < m) {
                        double sv = bv[i2];
#pragma unroll
                        for (int k2 = 0; k2 < i2; ++k2) sv -= Rs[k2 + MM * i2] * bv[k2];
                        bv[i2] = sv * ris[i2];
                    }
                }
#pragma unroll
                for (int i2 = MM - 1; i2 >= 0; --i2) {
                    if (i2 < m) {
                        double sv = bv[i2];
#pragma unroll
                        for (int k2 = i2 + 1; k2 < MM; ++k2)
                            if (k2 < m) sv -= Rs[i2 + MM * k2] * bv[k2];
                        bv[i2] = sv * ris[i2];
                    }
                }
            } else {
                chol_solve_ri<MM>(m, R, ri, bv);
            }
        };
        if constexpr (RL) {
            {   // QuuF (:247): one element per lane
                const int r2 = lane & 7, c2 = lane >> 3;
                double v = 0.0;
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
            wave_sync();
            fail = 0;                                              // chol_masked_ri's statements (boxqp_dev.h) on the LDS image, nothing clamped;
            // unrolled with guards like the original: the reads of a column are independent and go out together (run-time loops made
            // every one of the ~100 inner iterations an LDS round trip)
#pragma unroll
            for (int j2 = 0; j2 < MM; ++j2) {
                if (j2 < m) {
                    double cj_[MM];                               // column j2 of R above the diagonal
#pragma unroll
                    for (int k2 = 0; k2 < j2; ++k2) cj_[k2] = Rs[k2 + MM * j2];
                    double ajj = Hs[j2 + MM * j2];
#pragma unroll
                    for (int k2 = 0; k2 < j2; ++k2) ajj -= cj_[k2] * cj_[k2];
                    if (!(ajj > 0.0) && fail == 0) fail = j2 + 1;
                    const double rr = ddp_rsqrt(ajj);
                    ris[j2] = rr;
                    Rs[j2 + MM * j2] = ajj * rr;
#pragma unroll
                    for (int i2 = j2 + 1; i2 < MM; ++i2) {
                        if (i2 < m) {
                            double sv = Hs[j2 + MM * i2];
#pragma unroll
                            for (int k2 = 0; k2 < j2; ++k2) sv -= cj_[k2] * Rs[k2 + MM * i2];
                            Rs[j2 + MM * i2] = sv * rr;
                        }
                    }
                }
            }
            wave_sync();
#pragma unroll
            for (int q = 0; q < MM; ++q) kk[q] = Qu[q];
            solve(kk);
#pragma unroll
            for (int q = 0; q < MM; ++q) kk[q] = -kk[q];             // k_i = -(R\Qu)  (:41)
        } else {
#pragma unroll
            for (int c2 = 0; c2 < MM; ++c2)
#pragma unroll
                for (int r2 = 0; r2 < MM; ++r2) H[r2 + MM * c2] = (r2 < m && c2 < m) ? Gu[r2 * PC + n + c2] : 0.0;
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
                fail = chol_masked_ri<MM>(m, H, 0u, R, ri);            // cholesky(Hermitian(QuuF))  (:35)
#pragma unroll
                for (int q = 0; q < MM; ++q) kk[q] = Qu[q];
                solve(kk);
#pragma unroll
                for (int q = 0; q < MM; ++q) kk[q] = -kk[q];         // k_i = -(R\Qu)  (:41)
            } else {
                double lo[MM], up[MM];
#pragma unroll
                for (int q = 0; q < MM; ++q) { const double uq = q < m ? ug[(size_t)m * i + q] : 0.0; lo[q] = limlo[q] - uq; up[q] = limhi[q] - uq; }   // (:45-46)
                int iters;
                const int result = boxqp_dev_ri<MM>(m, H, Qu, lo, up, kprev, qpo, kk, R, ri, clamped, iters);      // (:49), warm start k[:, min(i+1, N-1)]
                fail = (result < 1);                                 // (:53)
            }
        }
        if (lane < (int)mm) Quug[mm * i + lane] = Gu[(lane % m) * PC + n + lane / m];     // assigned before a failure upstream too
        if (fail) { diverge = i + 1; break; }                        // (:37-38, :54-55): wave-uniform
        double Quuk[MM];
#pragma unroll
        for (int q = 0; q < MM; ++q) {
            double t = 0.0;
#pragma unroll
            for (int q2 = 0; q2 < MM; ++q2) t += (q2 < m ? Gu[q * PC + n + q2] : 0.0) * kk[q2];
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
            double col[MM], x2[MM];
#pragma unroll
            for (int q = 0; q < MM; ++q) { x2[q] = (q < m && lane < n) ? Gu[q * PC + lane] : 0.0; col[q] = ((clamped >> q) & 1u) ? 0.0 : xr[q]; }
            solve(col);
#pragma unroll
            for (int q = 0; q < MM; ++q) col[q] = (((clamped >> q) & 1u) || q >= m || lane >= n) ? 0.0 : -col[q];
            double vx = lane < n ? qxs[lane] : 0.0;
#pragma unroll
            for (int q = 0; q < MM; ++q) {
                double t = 2.0 * x2[q];
#pragma unroll
                for (int q2 = 0; q2 < MM; ++q2) t += (q2 < m ? Gu[q * PC + n + q2] : 0.0) * col[q2];
                vx += col[q] * (Quuk[q] + Qu[q]) + x2[q] * kk[q];
                if (lane < NR) { Ks[q * NR + lane] = col[q]; Ys[q * NR + lane] = (q < m && lane < n) ? 0.5 * t : 0.0; }
                if (q < m && lane < n) Kg[nm * i + q + (size_t)m * lane] = col[q];       // (:76)
            }
            if (lane < n) { Vxg[(size_t)n * i + lane] = vx; vxs[lane] = vx; }
            if (lane < m) {
                double kv = kk[0];
#pragma unroll
                for (int q = 1; q < MM; ++q) kv = (lane == q) ? kk[q] : kv;
                kg[(size_t)m * i + lane] = kv;                                           // (:75)
            }
        }
        wave_sync();
        // ================= Vxx_i = Qxx + ½(K'Y + Y'K)  (:70-72): rank-2m update of the xx tiles, then ½(V + V') ==========
#pragma unroll
        for (int ks = 0; ks < MM / 4; ++ks) {                           // (rows q >= MM of K, Y do not exist: MM / 4 k-steps)
            double ka[NTR], ya[NTR];
#pragma unroll
            for (int t = 0; t < NTR; ++t) { ka[t] = Ks[(4 * ks + l4) * NR + 16 * t + l15]; ya[t] = Ys[(4 * ks + l4) * NR + 16 * t + l15]; }
#pragma unroll
            for (int ti = 0; ti < NTR; ++ti)
#pragma unroll
                for (int tj = 0; tj < NTR; ++tj) {
                    g[ti][tj] = mf(ka[ti], ya[tj], g[ti][tj]);           // K'(½Y)
                    g[ti][tj] = mf(ya[ti], ka[tj], g[ti][tj]);           // (½Y)'K
                }
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
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int e = lane + DDP_WAVE * r;
            if (e < (int)nn) {
                const double v = 0.5 * (Vr[s_a[r]] + Vr[s_b[r]]);
                Vs[s_a[r]] = v;
                Vxxg[nn * i + e] = v;
            }
        }
        wave_sync();
    }
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

template <int NTR, int PT, int MMX>
int launch_mid8(ddp_handle h, const ddp_bp_desc *d, const BPMid8Args &a)
{
    const size_t bytes = (size_t)Mid8Lds<NTR, PT, MMX>::oTot * sizeof(double);
    const dim3 grid((unsigned)d->B), block(DDP_WAVE);
    if (d->has_lims) {
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mid8_kernel<NTR, PT, MMX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL((back_pass_mid8_kernel<NTR, PT, MMX, true>), grid, block, bytes, h->stream, a);
    } else {
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mid8_kernel<NTR, PT, MMX, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL((back_pass_mid8_kernel<NTR, PT, MMX, false>), grid, block, bytes, h->stream, a);
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// returns 1 if the shape is not handled here (n > 32, m > 8, n + m + 1 > 48), 0 launched, < 0 error
int ddp_launch_back_pass_mid8(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge)
{
    const int n = d->n, m = d->m;
    if (n < 1 || m < 1 || n > 32 || m > 8) return 1;
    const long N = d->N;
    BPMid8Args a;
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
    if (m <= 4) return 1;                                                 // (back_pass_mid.hip)
    if (ntr == 1) return launch_mid8<1, 2, 8>(h, d, a);
    return pt <= 2 ? launch_mid8<2, 2, 8>(h, d, a) : launch_mid8<2, 3, 8>(h, d, a);
}
