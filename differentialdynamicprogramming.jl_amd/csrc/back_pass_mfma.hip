// back_pass_mfma.hip — backward pass for the BASELINE config-4 shape n = 64, m = 8 with the two products of the
// Q-function expansion on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).  Same arithmetic and failure semantics as
// back_pass.hip / back_pass_big.hip (src/backward_pass.jl:179-215 + :28-79).
//
// One 256-thread work-group (4 waves = the 4 SIMDs of a CU) per trajectory; LDS holds (leading dimension 80
// doubles, so that the four 16-lane groups of an MFMA operand read fall on disjoint bank ranges):
//     Vs [64 x 64]   Vxx_{i+1}, column-major                       (A operand of W = Vxx·F)
//     FT [80 x 64]   F' = [fx fu]' (column index of F fastest)     (B operand of W = Vxx·F, A operand of G = F'W)
//     WT [80 x 64]   W' (column index of W fastest)                (B operand of G = F'W)
//   P1  wave w: W[16w..16w+15, :] = 5 column tiles x 16 k-steps = 80 MFMAs;   q = [cx;cu] + F'Vx on the VALU
//   P2  the 15 upper tiles of G = F'W (72 x 72 padded to 80) over the 4 waves, 16 MFMAs each = 60 per wave;
//       Qxx (+cxx) goes to Vs in place (Vxx_{i+1} is dead after P1), Qux / Quu to small arrays
//   P3  gains: every thread factorises the 8x8 QuuF redundantly (or boxQP), thread c < 64 solves column c of K
//   P4  Vxx_i = sym(Qxx) + ½(S+S') element-wise on the VALU (rank-8 update), coalesced stores.
// Measured (profiles/microbench/mfma_f64_bench.hip): 30 ns per MFMA per wave with ONE wave per SIMD (66-70 TF/s of the
// 78.6 TF/s peak) — unlike the fp64 VALU, the matrix pipe does not need several waves to fill.
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

struct BPMArgs {
    int N, B;
    int fx_tv, fx_batched, cost_tv, cost_batched, regType, has_lims;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
};

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NT = 256, n = 64, m = 8, p = 72, PP = 80, LD = 80, LDS_ = 65;   // LDS_: stride of the symmetric staging copy
constexpr int oVs = 0, oFT = oVs + n * LD, oWT = oFT + n * LD, ovs = oWT + n * LD, oQs = ovs + n, oXs = oQs + PP,
              oXrs = oXs + m * n, oQuus = oXrs + m * n, oQuuFs = oQuus + m * m, oKs = oQuuFs + m * m, oYs = oKs + m * n,
              oks = oYs + m * n, oQuuks = oks + m, oTot = oQuuks + m;

__global__ __launch_bounds__(NT) void back_pass_mfma_kernel(BPMArgs a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.active && a.active[b] == 0) return;
    const int N = a.N, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Vs = lds + oVs, *FT = lds + oFT, *WT = lds + oWT, *vs = lds + ovs, *Qs = lds + oQs, *Xs = lds + oXs, *Xrs = lds + oXrs,
           *Quus = lds + oQuus, *QuuFs = lds + oQuuFs, *Ks = lds + oKs, *Ys = lds + oYs, *ks = lds + oks, *Quuks = lds + oQuuks;

    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
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
    double limlo[m], limhi[m];
    if (LIMS) {
        nolims = a.lims[0] > a.lims[m];                             // backward_pass.jl:31
#pragma unroll
        for (int q = 0; q < m; ++q) { limlo[q] = a.lims[q]; limhi[q] = a.lims[q + m]; }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35

    // ---- terminal step (backward_pass.jl:197-199)
    for (int e = tid; e < n * n; e += NT) {
        const double v = cxx[(CTV ? nn * (N - 1) : 0) + e];
        Vs[(e & 63) + LD * (e >> 6)] = v;
        Vxxg[nn * (N - 1) + e] = v;
    }
    if (tid < n) { const double v = cx[(size_t)n * (N - 1) + tid]; vs[tid] = v; Vxg[(size_t)n * (N - 1) + tid] = v; }
    if (tid < m * m) Quug[mm * (N - 1) + tid] = cuu[(CTV ? mm * (N - 1) : 0) + tid];
    for (int e = tid; e < m * n; e += NT) Kg[nm * (N - 1) + e] = 0.0;
    if (tid < m) { kg[(size_t)m * (N - 1) + tid] = 0.0; ks[tid] = 0.0; }
    for (int e = tid; e < (PP - p) * n; e += NT) FT[p + (e % (PP - p)) + LD * (e / (PP - p))] = 0.0;   // zero padding columns 72..79
    if (N < 2) {
        if (tid == 0) { a.dV[2 * b] = 0.0; a.dV[2 * b + 1] = 0.0; a.diverge[b] = 0; }
        return;
    }
    constexpr int RF = n * p / NT;                    // 18 elements of F per thread
    auto load_F = [&](int i, double (&r)[RF]) {
#pragma unroll
        for (int q = 0; q < RF; ++q) {
            const int e = tid + NT * q;
            r[q] = (e < n * n) ? fx[nn * (FXTV ? i : 0) + e] : fu[nm * (FXTV ? i : 0) + (e - n * n)];
        }
    };
    auto store_F = [&](const double (&r)[RF]) {       // F[k, c] (k fastest in memory) -> FT[c + LD*k]
#pragma unroll
        for (int q = 0; q < RF; ++q) {
            const int e = tid + NT * q;
            FT[(e >> 6) + LD * (e & 63)] = r[q];
        }
    };
    double pfF[RF];
    load_F(N - 2, pfF);
    store_F(pfF);
    __syncthreads();

    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    for (int i = N - 2; i >= 0; --i) {
        const double *cxxi = cxx + (CTV ? nn * i : 0), *cxui = cxu + (CTV ? nm * i : 0), *cuui = cuu + (CTV ? mm * i : 0);
        if (FXTV && i > 0) load_F(i - 1, pfF);          // next step's Jacobian lands while this step computes

        // ================= P1: W = Vxx·F on the matrix cores; Qs = [cx;cu] + F'Vx ============================
        {
            d4 acc[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
            const double *ap = Vs + 16 * wv + l15 + LD * l4;          // A[i][k] = Vxx[16w+i, k]
            const double *bp = FT + l15 + LD * l4;                    // B[k][j] = F[k, 16c+j]
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const double av = ap[LD * 4 * kk];
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[16 * c + LD * 4 * kk], acc[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 5; ++c) {                             // D[row = l4 + 4r][col = l15] -> WT[col + LD*row]
                double *wp = WT + 16 * c + l15 + LD * (16 * wv + l4);
                wp[0] = acc[c].x; wp[LD * 4] = acc[c].y; wp[LD * 8] = acc[c].z; wp[LD * 12] = acc[c].w;
            }
        }
        if (tid < p) {
            double s = 0.0;
#pragma unroll 8
            for (int kq = 0; kq < n; ++kq) s += FT[tid + LD * kq] * vs[kq];
            Qs[tid] = (tid < n ? cx[(size_t)n * i + tid] : cu[(size_t)m * i + (tid - n)]) + s;      // (:203-204)
        }
        __syncthreads();

        // ================= P2: upper tiles of G = F'W on the matrix cores ====================================
        for (int t = wv; t < 15; t += 4) {
            // tile list (ti <= tj) over 5x5: column-major upper triangle
            int tj = 0, base = 0;
            while (base + tj + 1 <= t) { base += tj + 1; ++tj; }
            const int ti = t - base;
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
            const double *ap = FT + 16 * ti + l15 + LD * l4;          // A[i][k] = F[k, 16ti+i]
            const double *bp = WT + 16 * tj + l15 + LD * l4;          // B[k][j] = W[k, 16tj+j]
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc =__builtin_amdgcn_mfma_f64_16x16x4f64(ap[LD * 4 * kk], bp[LD * 4 * kk], acc, 0, 0, 0);
            const double gv[4] = {acc.x, acc.y, acc.z, acc.w};
            const int gj = 16 * tj + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 16 * ti + l4 + 4 * r;
                const double g = gv[r];
                if (tj < 4) {                                         // Qxx[gi, gj]  (:210) — stored at (gj, gi): lanes contiguous
                    Vs[gj + LD * gi] = g + cxxi[gi + n * gj];
                } else if (ti < 4) {                                  // G[x row gi, u col] = Qux[a, gi] - cxu[gi, a]   (:208)
                    const int aq = gj - n;
                    if (aq < m) Xs[aq + m * gi] = g + cxui[gi + n * aq];
                } else {                                              // Quu block (:209)
                    const int aq = gi - n, bq = gj - n;
                    if (aq < m && bq < m) Quus[aq + m * bq] = g + cuui[aq + m * bq];
                }
            }
        }
        __syncthreads();
        // regularisation (:205-207): regType 1 -> QuuF = Quu + λI, Qux_reg = Qux; regType 2 -> + λ·F_u'F
        for (int e = tid; e < m * n + m * m; e += NT) {
            const bool isx = e < m * n;
            const int q = isx ? (e & 7) : ((e - m * n) & 7), j = isx ? (e >> 3) : n + ((e - m * n) >> 3);
            double add = 0.0;
            if (regType == 2) {
                double s = 0.0;
#pragma unroll 8
                for (int kq = 0; kq < n; ++kq) s += FT[n + q + LD * kq] * FT[j + LD * kq];
                add = lam * s;
            } else if (!isx && q == j - n) add = lam;
            if (isx) Xrs[e] = Xs[e] + add;
            else QuuFs[e - m * n] = Quus[e - m * n] + add;
        }
        __syncthreads();

        // ================= P3: gains (backward_pass.jl:30-62) =================================================
        double H[m * m], R[m * m], kk[m];
        unsigned clamped = 0u;
#pragma unroll
        for (int e = 0; e < m * m; ++e) H[e] = QuuFs[e];
        int fail;
        double ri[m];
        const bool use_ri = !LIMS || nolims;                     // division-free factor on the unconstrained path
        if (use_ri) {
            fail = ddp_chol_rinv<m>(H, R, ri);                   // cholesky(Hermitian(QuuF))  (:35)
#pragma unroll
            for (int q = 0; q < m; ++q) kk[q] = Qs[n + q];
            ddp_rsolve_neg<m>(R, ri, kk);                        // k_i = -(R\Qu)  (:41)
        } else {
            double g[m], lo[m], up[m], x0[m];
#pragma unroll
            for (int q = 0; q < m; ++q) {
                const double uq = ug[(size_t)m * i + q];
                g[q] = Qs[n + q]; lo[q] = limlo[q] - uq; up[q] = limhi[q] - uq; x0[q] = ks[q];
            }
            int iters;
            const int result = boxqp_dev<m>(m, H, g, lo, up, x0, qpo, kk, R, clamped, iters);
            fail = (result < 1);
        }
        if (fail) {                                              // block-uniform: diverge = i
            diverge = i + 1;
            if (tid < m * m) Quug[mm * i + tid] = Quus[tid];
            break;
        }
        __syncthreads();                                         // everyone has read ks (warm start) before it is rewritten
        if (tid < n) {                                           // K_i column tid, Y = Quu·K + 2·Qux
            double col[m];
#pragma unroll
            for (int q = 0; q < m; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : Xrs[q + m * tid];
            if (use_ri) ddp_rsolve_neg<m>(R, ri, col);
            else {
                chol_solve<m>(m, R, col);
#pragma unroll
                for (int q = 0; q < m; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : -col[q];
            }
#pragma unroll
            for (int q = 0; q < m; ++q) {
                double t = 2.0 * Xs[q + m * tid];
#pragma unroll
                for (int q2 = 0; q2 < m; ++q2) t += Quus[q + m * q2] * col[q2];
                Ks[q + m * tid] = col[q];
                Ys[q + m * tid] = t;
                Kg[nm * i + q + (size_t)m * tid] = col[q];       // (:76)
            }
        } else if (tid == n) {                                   // k_i, Quu·k, dV (:64-68)
            double kQu = 0.0, kQuuk = 0.0;
#pragma unroll
            for (int q = 0; q < m; ++q) {
                double t = 0.0;
#pragma unroll
                for (int q2 = 0; q2 < m; ++q2) t += Quus[q + m * q2] * kk[q2];
                Quuks[q] = t; ks[q] = kk[q];
                kg[(size_t)m * i + q] = kk[q];
                kQu += kk[q] * Qs[n + q]; kQuuk += kk[q] * t;
            }
            dV0 += kQu; dV1 += 0.5 * kQuuk;
        } else if (tid >= 128 && tid < 128 + m * m) {
            Quug[mm * i + (tid - 128)] = Quus[tid - 128];
        }
        __syncthreads();

        // ================= P4: Vxx_i = sym(Qxx) + ½(S+S')  (:69-72), Vx_i ======================================
        // Qxx[i,j] (i <= j) sits at Vs[j + LD*i]; inside a diagonal tile both (i,j) and (j,i) were computed and
        // are averaged like the reference does, across tiles only the upper one exists (used for both).
        for (int t = tid; t < n * (n + 1) / 2; t += NT) {
            int cb = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);      // packed upper triangle (ca <= cb) ...
            while (cb * (cb + 1) / 2 > t) --cb;
            while ((cb + 1) * (cb + 2) / 2 <= t) ++cb;
            const int ca = t - cb * (cb + 1) / 2;
            const int ii = n - 1 - cb, jj = n - 1 - ca;          // ... mirrored: ii <= jj and jj runs contiguously over the lanes
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < m; ++q) s += Ks[q + m * ii] * Ys[q + m * jj] + Ys[q + m * ii] * Ks[q + m * jj];
            const double qa = Vs[jj + LD * ii];
            const double qb = ((ii >> 4) == (jj >> 4)) ? Vs[ii + LD * jj] : qa;
            const double v = 0.5 * (qa + qb) + 0.5 * s;
            WT[ii + LDS_ * jj] = v; WT[jj + LDS_ * ii] = v;      // staged in WT (dead after P2), odd stride: both writes conflict-free
        }
        if (tid < n) {                                           // Vx_i (:69)
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int q = 0; q < m; ++q) {
                s1 += Ks[q + m * tid] * Quuks[q];
                s2 += Ks[q + m * tid] * Qs[n + q];
                s3 += Xs[q + m * tid] * ks[q];
            }
            Qs[tid] = ((Qs[tid] + s1) + s2) + s3;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += NT) {
            const double v = WT[(e & 63) + LDS_ * (e >> 6)];
            Vs[(e & 63) + LD * (e >> 6)] = v; Vxxg[nn * i + e] = v;
        }
        if (tid < n) { const double v = Qs[tid]; vs[tid] = v; Vxg[(size_t)n * i + tid] = v; }
        if (FXTV && i > 0) store_F(pfF);
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
int ddp_launch_back_pass_mfma(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                              const double *cxx, const double *cxu, const double *cuu, const double *fx,
                              const double *fu, const double *lambda, const double *lims, const double *u,
                              const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                              double *Vxx, double *dV, int32_t *diverge)
{
    if (d->n != n || d->m != m) return 1;
    BPMArgs a;
    a.N = d->N; a.B = d->B;
    a.fx_tv = d->fx_tv; a.fx_batched = d->fx_batched; a.cost_tv = d->cost_tv; a.cost_batched = d->cost_batched;
    a.regType = d->regType; a.has_lims = d->has_lims;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.lims = lims;
    a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    const size_t shmem = (size_t)oTot * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(back_pass_mfma_kernel, dim3(d->B), dim3(NT), shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}
