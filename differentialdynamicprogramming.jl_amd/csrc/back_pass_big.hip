// back_pass_big.hip — backward pass for large states (32 < n <= 64, m <= 8, n and m even): BASELINE config 4
// (n=64, m=8: src/backward_pass.jl:179-215 with per-trajectory fx[n,n,N,B], fu[n,m,N,B]) and any other large shape.
//
// One 256-thread work-group (4 waves, one per SIMD of a CU) per trajectory.  Vxx (n x n), the stacked Jacobian
// F = [fx fu] (n x p) and W = Vxx·F (n x p) live in LDS (106 KB at n=64, m=8: one work-group per CU); both
// products run as 2x2 register blocks over 16-byte LDS operand reads (1 ds_read_b128 per 2 FMAs):
//   P1  W = Vxx·F,   q = [cx;cu] + F'Vx
//   P2  G = F'W: the x-block (both triangles, mirrored) overwrites the Vxx buffer in place — Vxx_{i+1} is dead
//       after P1 and the value update is element-wise on Qxx — the u-rows go to small LDS arrays
//   P3  every thread factorises QuuF (m <= 8) redundantly (or runs boxQP), thread c < n solves column c of K
//   P4  Vxx_i = sym(Qxx) + ½(S+S'), (S+S')[i,j] = Σ_a K[a,i]·Y[a,j] + Y[a,i]·K[a,j], Y = Quu·K + 2·Qux
//       (same algebra as back_pass_dpp.hip), then coalesced stores of Vxx_i, K_i, Vx_i, k_i, Quu_i from LDS.
// fp64 MFMA (v_mfma_f64_16x16x4_f64) is NOT used yet: on MI355X its peak equals the fp64 vector peak, so the gain is
// LDS operand traffic only (0.5 vs 2 B/flop); at ~15 k LDS cycles per step this kernel already sits near the HBM
// time of the step's 113 KB (C4) — MFMA tiles are the round-2 refinement.
// Arithmetic and failure semantics as in back_pass.hip (diverge index, zero-filled earlier outputs).
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

struct BPBArgs {
    int n, m, N, B;
    int fx_tv, fx_batched, cost_tv, cost_batched, regType, has_lims;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
};

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int NT = 256, MMAX = 8;

struct BigLds {            // offsets in doubles (all even)
    int Fs, Vs, Ws, vs, Qs, Xs, Xrs, Quus, QuuFs, Ks, Ys, ks, Quuks, flag, total;
    __host__ __device__ BigLds(int n, int m)
    {
        const int p = n + m;
        int o = 0;
        const int ld = n + 2;          // padded leading dimension: consecutive 2-row blocks land 8 banks apart (not on one bank)
        Fs = o; o += ld * p;
        Vs = o; o += ld * n;
        Ws = o; o += ld * p;
        vs = o; o += n;
        Qs = o; o += p;
        Xs = o; o += m * n;            // Qux
        Xrs = o; o += m * n;           // Qux_reg
        Quus = o; o += m * m;
        QuuFs = o; o += m * m;
        Ks = o; o += m * n;
        Ys = o; o += m * n;
        ks = o; o += m;
        Quuks = o; o += m;
        flag = o; o += 2;
        total = (o + 1) & ~1;
    }
};

__global__ __launch_bounds__(NT) void back_pass_big_kernel(BPBArgs a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.active && a.active[b] == 0) return;
    const int n = a.n, m = a.m, N = a.N, p = n + m, nh = n / 2, ph = p / 2, LD = n + 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const BigLds L(n, m);
    double *Fs = lds + L.Fs, *Vs = lds + L.Vs, *Ws = lds + L.Ws, *vs = lds + L.vs, *Qs = lds + L.Qs, *Xs = lds + L.Xs,
           *Xrs = lds + L.Xrs, *Quus = lds + L.Quus, *QuuFs = lds + L.QuuFs, *Ks = lds + L.Ks, *Ys = lds + L.Ys,
           *ks = lds + L.ks, *Quuks = lds + L.Quuks;

    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const bool FXTV = a.fx_tv, CTV = a.cost_tv, LIMS = a.has_lims;
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
    double limlo[MMAX], limhi[MMAX];
    if (LIMS) {
        nolims = a.lims[0] > a.lims[m];                             // backward_pass.jl:31
#pragma unroll
        for (int q = 0; q < MMAX; ++q) { limlo[q] = (q < m) ? a.lims[q] : 0.0; limhi[q] = (q < m) ? a.lims[q + m] : 0.0; }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35

    // ---- terminal step (backward_pass.jl:197-199 / :234-236)
    for (int e = tid; e < n * n; e += NT) {
        const double v = cxx[(CTV ? nn * (N - 1) : 0) + e];
        Vs[(e % n) + LD * (e / n)] = v;
        Vxxg[nn * (N - 1) + e] = v;
    }
    for (int e = tid; e < n; e += NT) { const double v = cx[(size_t)n * (N - 1) + e]; vs[e] = v; Vxg[(size_t)n * (N - 1) + e] = v; }
    for (int e = tid; e < m * m; e += NT) Quug[mm * (N - 1) + e] = cuu[(CTV ? mm * (N - 1) : 0) + e];
    for (int e = tid; e < m * n; e += NT) Kg[nm * (N - 1) + e] = 0.0;
    for (int e = tid; e < m; e += NT) { kg[(size_t)m * (N - 1) + e] = 0.0; ks[e] = 0.0; }
    if (N < 2) {
        if (tid == 0) { a.dV[2 * b] = 0.0; a.dV[2 * b + 1] = 0.0; a.diverge[b] = 0; }
        return;
    }
    {
        const size_t off = FXTV ? (size_t)(N - 2) : 0;
        for (int e = tid; e < n * n; e += NT) Fs[(e % n) + LD * (e / n)] = fx[nn * off + e];
        for (int e = tid; e < n * m; e += NT) Fs[(e % n) + LD * (n + e / n)] = fu[nm * off + e];
    }
    __syncthreads();

    const int tid_r = tid % n, tid_c = tid / n;       // (row, column) of flat element `tid` of a dense n-row matrix
    constexpr int RF = (64 * 72 + NT - 1) / NT;       // F elements per thread (prefetch registers)
    double pfF[RF];
    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    for (int i = N - 2; i >= 0; --i) {
        const double *cxxi = cxx + (CTV ? nn * i : 0), *cxui = cxu + (CTV ? nm * i : 0), *cuui = cuu + (CTV ? mm * i : 0);
        if (FXTV && i > 0) {                          // next step's Jacobian lands while this step computes
#pragma unroll
            for (int r = 0; r < RF; ++r) {
                const int e = tid + NT * r;
                if (e < n * p) pfF[r] = (e < n * n) ? fx[nn * (i - 1) + e] : fu[nm * (i - 1) + (e - n * n)];
            }
        }
        // ================= P1: W = Vxx·F (2x2 blocks), Qs = [cx;cu] + F'Vx =================================
        for (int blk = tid; blk < nh * ph; blk += NT) {
            const int r0 = 2 * (blk % nh), j0 = 2 * (blk / nh);
            const double *v0 = Vs + r0 * LD, *v1 = v0 + LD, *f0 = Fs + j0 * LD, *f1 = f0 + LD;   // Vxx row r == column r
            double w00 = 0.0, w01 = 0.0, w10 = 0.0, w11 = 0.0;
#pragma unroll 8
            for (int l = 0; l < n; l += 2) {
                const d2 a0 = *(const d2 *)(v0 + l), a1 = *(const d2 *)(v1 + l), b0 = *(const d2 *)(f0 + l), b1 = *(const d2 *)(f1 + l);
                w00 += a0.x * b0.x; w01 += a0.x * b1.x; w10 += a1.x * b0.x; w11 += a1.x * b1.x;
                w00 += a0.y * b0.y; w01 += a0.y * b1.y; w10 += a1.y * b0.y; w11 += a1.y * b1.y;
            }
            *(d2 *)(Ws + r0 + LD * j0) = d2{w00, w10};
            *(d2 *)(Ws + r0 + LD * (j0 + 1)) = d2{w01, w11};
        }
        for (int j = tid; j < p; j += NT) {
            const double *fc = Fs + j * LD;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += fc[l] * vs[l];
            Qs[j] = (j < n ? cx[(size_t)n * i + j] : cu[(size_t)m * i + (j - n)]) + s;       // (:203-204)
        }
        __syncthreads();

        // ================= P2: G = F'W; x-block -> Vs (in place), u-rows -> Xs/Xrs/Quus/QuuFs ===============
        for (int t = tid; t < ph * (ph + 1) / 2; t += NT) {
            int jb = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (jb * (jb + 1) / 2 > t) --jb;
            while ((jb + 1) * (jb + 2) / 2 <= t) ++jb;
            const int ib = t - jb * (jb + 1) / 2;                  // block row <= block column
            const int i0 = 2 * ib, j0 = 2 * jb;
            const double *fa = Fs + i0 * LD, *fb = fa + LD, *wa = Ws + j0 * LD, *wb = wa + LD;
            double g00 = 0.0, g01 = 0.0, g10 = 0.0, g11 = 0.0, s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0;
            const bool urow = i0 >= n;                             // rows of the block are u-rows (need the λ·F'F term)
            const double *fja = Fs + j0 * LD, *fjb = fja + LD;
#pragma unroll 8
            for (int l = 0; l < n; l += 2) {
                const d2 a0 = *(const d2 *)(fa + l), a1 = *(const d2 *)(fb + l), b0 = *(const d2 *)(wa + l), b1 = *(const d2 *)(wb + l);
                g00 += a0.x * b0.x; g01 += a0.x * b1.x; g10 += a1.x * b0.x; g11 += a1.x * b1.x;
                g00 += a0.y * b0.y; g01 += a0.y * b1.y; g10 += a1.y * b0.y; g11 += a1.y * b1.y;
                if (urow && regType == 2) {
                    const d2 c0 = *(const d2 *)(fja + l), c1 = *(const d2 *)(fjb + l);
                    s00 += a0.x * c0.x + a0.y * c0.y; s01 += a0.x * c1.x + a0.y * c1.y;
                    s10 += a1.x * c0.x + a1.y * c0.y; s11 += a1.x * c1.x + a1.y * c1.y;
                }
            }
            if (j0 < n) {                                          // block inside Qxx (rows i0,i0+1 <= cols j0,j0+1 < n)
                const double q00 = g00 + cxxi[i0 + n * j0], q01 = g01 + cxxi[i0 + n * (j0 + 1)],
                             q10 = g10 + cxxi[i0 + 1 + n * j0], q11 = g11 + cxxi[i0 + 1 + n * (j0 + 1)];
                Vs[i0 + LD * j0] = q00; Vs[i0 + LD * (j0 + 1)] = q01; Vs[i0 + 1 + LD * (j0 + 1)] = q11;
                if (ib != jb) { Vs[i0 + 1 + LD * j0] = q10; Vs[j0 + LD * i0] = q00; Vs[j0 + 1 + LD * i0] = q01; Vs[j0 + LD * (i0 + 1)] = q10; Vs[j0 + 1 + LD * (i0 + 1)] = q11; }
                else Vs[i0 + 1 + LD * j0] = q01;                    // diagonal block: (i0+1,i0) mirrors (i0,i0+1)
            } else if (!urow) {
                // rows x, cols u: G[i, n+a] = Qux[a, i] by symmetry — not needed (taken from the u-row blocks' transposes below)
                // here the block is (x rows i0.., u cols j0..): it IS the transpose of a Qux block; store it as such
                const int a0_ = j0 - n;
                Xs[a0_ + m * i0] = g00 + cxui[i0 + n * a0_];           Xs[a0_ + 1 + m * i0] = g01 + cxui[i0 + n * (a0_ + 1)];
                Xs[a0_ + m * (i0 + 1)] = g10 + cxui[i0 + 1 + n * a0_]; Xs[a0_ + 1 + m * (i0 + 1)] = g11 + cxui[i0 + 1 + n * (a0_ + 1)];
            } else {                                               // u rows, u cols: Quu block (upper block-triangle)
                const int a0_ = i0 - n, b0_ = j0 - n;
                const double u00 = g00 + cuui[a0_ + m * b0_], u01 = g01 + cuui[a0_ + m * (b0_ + 1)],
                             u10 = g10 + cuui[a0_ + 1 + m * b0_], u11 = g11 + cuui[a0_ + 1 + m * (b0_ + 1)];
                const double l00 = (regType == 2) ? lam * s00 : (a0_ == b0_ ? lam : 0.0), l01 = (regType == 2) ? lam * s01 : 0.0,
                             l10 = (regType == 2) ? lam * s10 : 0.0, l11 = (regType == 2) ? lam * s11 : (a0_ == b0_ ? lam : 0.0);
                Quus[a0_ + m * b0_] = u00; Quus[a0_ + m * (b0_ + 1)] = u01; Quus[a0_ + 1 + m * b0_] = u10; Quus[a0_ + 1 + m * (b0_ + 1)] = u11;
                QuuFs[a0_ + m * b0_] = u00 + l00; QuuFs[a0_ + m * (b0_ + 1)] = u01 + l01;
                QuuFs[a0_ + 1 + m * b0_] = u10 + l10; QuuFs[a0_ + 1 + m * (b0_ + 1)] = u11 + l11;
                if (ib != jb) {                                    // mirror the off-diagonal block
                    Quus[b0_ + m * a0_] = u00; Quus[b0_ + 1 + m * a0_] = u01; Quus[b0_ + m * (a0_ + 1)] = u10; Quus[b0_ + 1 + m * (a0_ + 1)] = u11;
                    QuuFs[b0_ + m * a0_] = u00 + l00; QuuFs[b0_ + 1 + m * a0_] = u01 + l01;
                    QuuFs[b0_ + m * (a0_ + 1)] = u10 + l10; QuuFs[b0_ + 1 + m * (a0_ + 1)] = u11 + l11;
                }
            }
        }
        __syncthreads();
        // Qux_reg = Qux + λ·F_u'F_x for regType 2 (backward_pass.jl:205-206); one dot product per entry
        for (int e = tid; e < m * n; e += NT) {
            double v = Xs[e];
            if (regType == 2) {
                const int q = e % m, j = e / m;
                const double *fc = Fs + (n + q) * LD, *fj = Fs + j * LD;
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += fc[l] * fj[l];
                v += lam * s;
            }
            Xrs[e] = v;
        }
        __syncthreads();

        // ================= P3: gains (backward_pass.jl:30-62) =================================================
        double H[MMAX * MMAX], R[MMAX * MMAX], kk[MMAX], ri[MMAX];
        unsigned clamped = 0u;
#pragma unroll
        for (int c2 = 0; c2 < MMAX; ++c2)
#pragma unroll
            for (int r2 = 0; r2 < MMAX; ++r2) H[r2 + MMAX * c2] = (r2 < m && c2 < m) ? QuuFs[r2 + m * c2] : 0.0;
        int fail;
        if (!LIMS || nolims) {
            fail = chol_masked_ri<MMAX>(m, H, 0u, R, ri);
#pragma unroll
            for (int q = 0; q < MMAX; ++q) kk[q] = (q < m) ? Qs[n + q] : 0.0;
            chol_solve_ri<MMAX>(m, R, ri, kk);
#pragma unroll
            for (int q = 0; q < MMAX; ++q) kk[q] = -kk[q];
        } else {
            double g[MMAX], lo[MMAX], up[MMAX], x0[MMAX];
#pragma unroll
            for (int q = 0; q < MMAX; ++q) {
                const double uq = (q < m) ? ug[(size_t)m * i + q] : 0.0;
                g[q] = (q < m) ? Qs[n + q] : 0.0;
                lo[q] = limlo[q] - uq; up[q] = limhi[q] - uq;
                x0[q] = (q < m) ? ks[q] : 0.0;
            }
            int iters;
            const int result = boxqp_dev_ri<MMAX>(m, H, g, lo, up, x0, qpo, kk, R, ri, clamped, iters);
            fail = (result < 1);
        }
        if (fail) {                                              // block-uniform: diverge = i
            diverge = i + 1;
            for (int e = tid; e < m * m; e += NT) Quug[mm * i + e] = Quus[e];
            break;
        }
        __syncthreads();                                         // everyone has read ks (warm start) before it is rewritten
        if (tid < n) {                                           // K_i column tid, Y = Quu·K + 2·Qux
            double col[MMAX];
#pragma unroll
            for (int q = 0; q < MMAX; ++q) col[q] = (q < m && !((clamped >> q) & 1u)) ? Xrs[q + m * tid] : 0.0;
            chol_solve_ri<MMAX>(m, R, ri, col);
#pragma unroll
            for (int q = 0; q < MMAX; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : -col[q];
#pragma unroll
            for (int q = 0; q < MMAX; ++q) {
                if (q < m) {
                    double t = 2.0 * Xs[q + m * tid];
#pragma unroll
                    for (int q2 = 0; q2 < MMAX; ++q2)
                        if (q2 < m) t += Quus[q + m * q2] * col[q2];
                    Ks[q + m * tid] = col[q];
                    Ys[q + m * tid] = t;
                    Kg[nm * i + q + (size_t)m * tid] = col[q];   // (:76)
                }
            }
        } else if (tid == n) {                                   // k_i, Quu·k, dV (:64-68)
            double kQu = 0.0, kQuuk = 0.0;
#pragma unroll
            for (int q = 0; q < MMAX; ++q) {
                if (q < m) {
                    double t = 0.0;
#pragma unroll
                    for (int q2 = 0; q2 < MMAX; ++q2)
                        if (q2 < m) t += Quus[q + m * q2] * kk[q2];
                    Quuks[q] = t; ks[q] = kk[q];
                    kg[(size_t)m * i + q] = kk[q];
                    kQu += kk[q] * Qs[n + q]; kQuuk += kk[q] * t;
                }
            }
            dV0 += kQu; dV1 += 0.5 * kQuuk;
        } else if (tid >= 128 && tid < 128 + m * m) {
            Quug[mm * i + (tid - 128)] = Quus[tid - 128];
        }
        __syncthreads();

        // ================= P4: Vxx_i = sym(Qxx) + ½(S+S')  (:69-72), Vx_i ======================================
        for (int t = tid; t < n * (n + 1) / 2; t += NT) {
            int jj = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (jj * (jj + 1) / 2 > t) --jj;
            while ((jj + 1) * (jj + 2) / 2 <= t) ++jj;
            const int ii = t - jj * (jj + 1) / 2;
            double s = 0.0;
            for (int q = 0; q < m; ++q) s += Ks[q + m * ii] * Ys[q + m * jj] + Ys[q + m * ii] * Ks[q + m * jj];
            const double v = 0.5 * (Vs[ii + LD * jj] + Vs[jj + LD * ii]) + 0.5 * s;
            Ws[ii + LD * jj] = v; Ws[jj + LD * ii] = v;            // staged in W (dead after P2): Vs is still being read
        }
        for (int j = tid; j < n; j += NT) {                      // Vx_i (:69) — written to vs after the barrier below
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int q = 0; q < m; ++q) {
                s1 += Ks[q + m * j] * Quuks[q];
                s2 += Ks[q + m * j] * Qs[n + q];
                s3 += Xs[q + m * j] * ks[q];
            }
            Qs[j] = ((Qs[j] + s1) + s2) + s3;
        }
        __syncthreads();
        {   // dense [n,n] <-> padded LDS index without a run-time division per element
            int rr = tid_r, cc = tid_c;
            for (int e = tid; e < n * n; e += NT) {
                const int o_ = rr + LD * cc;
                const double v = Ws[o_]; Vs[o_] = v; Vxxg[nn * i + e] = v;
                rr += NT; while (rr >= n) { rr -= n; ++cc; }
            }
        }
        for (int j = tid; j < n; j += NT) { const double v = Qs[j]; vs[j] = v; Vxg[(size_t)n * i + j] = v; }
        if (FXTV && i > 0) {
            int rr = tid_r, cc = tid_c;
#pragma unroll
            for (int r = 0; r < RF; ++r) {
                const int e = tid + NT * r;
                if (e < n * p) Fs[rr + LD * cc] = pfF[r];
                rr += NT; while (rr >= n) { rr -= n; ++cc; }
            }
        }
        __syncthreads();
    }
    if (diverge) {   // outputs earlier in time than the failing step are zero (backward_pass.jl:226-229)
        const size_t ie = (size_t)diverge;
        for (size_t e = tid; e < nm * ie; e += NT) Kg[e] = 0.0;
        for (size_t e = tid; e < (size_t)m * ie; e += NT) kg[e] = 0.0;
        for (size_t e = tid; e < (size_t)n * ie; e += NT) Vxg[e] = 0.0;
        for (size_t e = tid; e < nn * ie; e += NT) Vxxg[e] = 0.0;
        for (size_t e = tid; e < mm * (ie - 1); e += NT) Quug[e] = 0.0;
    }
    if (tid == n) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; }
    if (tid == 0) a.diverge[b] = diverge;
}

}   // namespace

// returns 1 if this shape is not handled here, 0 launched, <0 error
int ddp_launch_back_pass_big(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge)
{
    if (d->n > 64 || d->m > MMAX || (d->n & 1) || (d->m & 1) || d->n < 2) return 1;
    BPBArgs a;
    a.n = d->n; a.m = d->m; a.N = d->N; a.B = d->B;
    a.fx_tv = d->fx_tv; a.fx_batched = d->fx_batched; a.cost_tv = d->cost_tv; a.cost_batched = d->cost_batched;
    a.regType = d->regType; a.has_lims = d->has_lims;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.lims = lims;
    a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    const BigLds L(d->n, d->m);
    const size_t shmem = (size_t)L.total * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    DDP_CHECK(shmem <= 160 * 1024, "back_pass: n=%d m=%d needs %zu bytes of LDS (> 160 KiB)", d->n, d->m, shmem);
    hipLaunchKernelGGL(back_pass_big_kernel, dim3(d->B), dim3(NT), shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}
