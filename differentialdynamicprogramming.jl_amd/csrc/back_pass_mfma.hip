// back_pass_mfma.hip — backward pass for the BASELINE config-4 shape n = 64, m = 8 with every product of the
// Riccati step on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).  Same arithmetic and failure semantics as
// back_pass.hip / back_pass_big.hip (src/backward_pass.jl:179-215 + :28-79).
//
// One 256-thread work-group (4 waves = the 4 SIMDs of a CU) per trajectory.  LDS:
//     Vs [64 x 64]  Vxx_{i+1}, symmetric, leading dimension LD = 80   (A operand of W = Vxx·F)
//     Fs [64 x 80]  F = [fx fu 0], state index fastest, leading dimension LDK = 66 — the coalesced global
//                   layout goes to LDS untransposed (conflict-free writes) and LDK ≡ 2 (mod 32) makes the
//                   k-major operand reads (B of W = Vxx·F, A of G = F'W) conflict-free too
//     WT [80 x 64]  W' (column index of W fastest, LD = 80); column 72 carries Vx_{i+1}, so G[:,72] = F'Vx
// Per step, 4 barriers:
//   P1   wave w: W[16w..16w+15, :] = 5 column tiles x 16 k-steps = 80 MFMAs, operands software-pipelined one
//        k-step ahead; Vxx_{i+1} streams to global from Vs in the same phase
//   P2a  wave w: tile (w,4) of G = F'W (Qux', Qx) + a quarter of the k-range of tile (4,4) (Quu, Qu) = 20 MFMAs
//   P3 | P2b  wave 0 reduces Quu and computes the gains (every lane factorises QuuF, lane c solves column c of K;
//        or the boxQP) WHILE waves 1-3 compute the 10 upper Qxx tiles (+cxx) into Vs (Vxx_{i+1} is dead after P1)
//   P4   Vxx_i = Qxx + ½(K'Y + Y'K): the rank-16 update [K;Y]'·½[Y;K] as 4 more MFMAs per upper tile with the
//        Qxx tile as the C operand; diagonal tiles are symmetrised through LDS, the others mirrored.
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
constexpr int NT = 256, n = 64, m = 8, p = 72, PP = 80, LD = 80, LDK = 66;
constexpr int oVs = 0, oFs = oVs + n * LD, oWT = oFs + PP * LDK, ovs = oWT + n * LD, oQs = ovs + n, oXs = oQs + PP,
              oXrs = oXs + m * n, oQuus = oXrs + m * n, oRadd = oQuus + m * m, oKs = oRadd + m * m, oYs = oKs + m * n,
              oks = oYs + m * n, oQuuks = oks + m, oPq = oQuuks + m, oFlag = oPq + 4 * 2 * 64, oTot = oFlag + 2;

__device__ __forceinline__ d4 mf(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }

template <bool LIMS>
__global__ __launch_bounds__(NT) void back_pass_mfma_kernel(BPMArgs a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.active && a.active[b] == 0) return;
    const int N = a.N, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Vs = lds + oVs, *Fs = lds + oFs, *WT = lds + oWT, *vs = lds + ovs, *Qs = lds + oQs, *Xs = lds + oXs, *Xrs = lds + oXrs,
           *Quus = lds + oQuus, *Radd = lds + oRadd, *Ks = lds + oKs, *Ys = lds + oYs, *ks = lds + oks, *Quuks = lds + oQuuks,
           *Pq = lds + oPq, *flag = lds + oFlag;

    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const bool FXTV = a.fx_tv, CTV = a.cost_tv;
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

    // ---- terminal step (backward_pass.jl:197-199); Vxx_{N-1} itself is streamed out by the first step below
    for (int e = tid; e < n * n; e += NT) Vs[(e & 63) + LD * (e >> 6)] = cxx[(CTV ? nn * (N - 1) : 0) + e];
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
    constexpr int RF = n * p / NT;                    // 18 elements of F per thread
    auto load_F = [&](int i, double (&r)[RF]) {
#pragma unroll
        for (int q = 0; q < RF; ++q) {
            const int e = tid + NT * q;
            r[q] = (e < n * n) ? fx[nn * (FXTV ? i : 0) + e] : fu[nm * (FXTV ? i : 0) + (e - n * n)];
        }
    };
    auto store_F = [&](const double (&r)[RF]) {       // F[k, c] (k fastest in memory) -> Fs[k + LDK*c]
#pragma unroll
        for (int q = 0; q < RF; ++q) {
            const int e = tid + NT * q;
            Fs[(e & 63) + LDK * (e >> 6)] = r[q];
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

        // ================= P1: W = Vxx·F on the matrix cores; column 72 of W := Vx ===========================
        {
            d4 acc[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
            const double *ap = Vs + 16 * wv + l15 + LD * l4;          // A[i][k] = Vxx[16w+i, k]
            const double *bp = Fs + l4 + LDK * l15;                   // B[k][j] = F[k, 16c+j]
            double a0 = ap[0], b0[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) b0[c] = bp[LDK * 16 * c];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                double a1 = 0.0, b1[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                if (kk < 15) {
                    a1 = ap[LD * 4 * (kk + 1)];
#pragma unroll
                    for (int c = 0; c < 5; ++c) b1[c] = bp[LDK * 16 * c + 4 * (kk + 1)];
                }
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[c] = mf(a0, b0[c], acc[c]);
                a0 = a1;
#pragma unroll
                for (int c = 0; c < 5; ++c) b0[c] = b1[c];
            }
            const bool vcol = (l15 == 8);
#pragma unroll
            for (int c = 0; c < 5; ++c) {                             // D[row = l4 + 4r][col = l15] -> WT[col + LD*row]
                double *wp = WT + 16 * c + l15 + LD * (16 * wv + l4);
                double w0 = acc[c].x, w1 = acc[c].y, w2 = acc[c].z, w3 = acc[c].w;
                if (c == 4 && vcol) { const double *vp = vs + 16 * wv + l4; w0 = vp[0]; w1 = vp[4]; w2 = vp[8]; w3 = vp[12]; }
                wp[0] = w0; wp[LD * 4] = w1; wp[LD * 8] = w2; wp[LD * 12] = w3;
            }
        }
        for (int e = tid; e < n * n; e += NT) Vxxg[nn * (i + 1) + e] = Vs[(e & 63) + LD * (e >> 6)];   // (:72) of step i+1
        __syncthreads();

        // ================= P2a: the u/Vx columns of G = F'W: Qux' (:208), Qx (:203), partial Quu/Qu ===========
        {
            double pre[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 16 * wv + l4 + 4 * r;
                pre[r] = (l15 < m) ? cxui[gi + n * l15] : (l15 == m ? cx[(size_t)n * i + gi] : 0.0);
            }
            d4 acc0 = d4{pre[0], pre[1], pre[2], pre[3]}, acc1 = d4{0.0, 0.0, 0.0, 0.0}, accq = d4{0.0, 0.0, 0.0, 0.0};
            const double *ap = Fs + l4 + LDK * (16 * wv + l15);       // A[i][k] = F[k, 16w+i]
            const double *aq = Fs + l4 + LDK * (n + l15) + 16 * wv;   // A[i][k] = F[k, 64+i], k-range [16w, 16w+16)
            const double *bp = WT + n + l15 + LD * l4;                // B[k][j] = W[k, 64+j]
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                acc0 = mf(ap[4 * kk], bp[LD * 4 * kk], acc0);
                acc1 = mf(ap[4 * kk + 4], bp[LD * 4 * kk + LD * 4], acc1);
            }
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) accq = mf(aq[4 * k2], bp[LD * 4 * (4 * wv + k2)], accq);
            const d4 g = acc0 + acc1;
            const double gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 16 * wv + l4 + 4 * r;
                if (l15 < m) Xs[l15 + m * gi] = gv[r];
                else if (l15 == m) Qs[gi] = gv[r];
            }
            Pq[(2 * wv) * 64 + lane] = accq.x;                        // rows l4, l4+4 of the (4,4) tile: the 8 u rows
            Pq[(2 * wv + 1) * 64 + lane] = accq.y;
        }
        __syncthreads();
        if (regType == 2) {     // (:205-207): QuuF = Quu + λ·fu'fu, Qux_reg = Qux + λ·fu'fx
            for (int e = tid; e < m * n + m * m; e += NT) {
                const bool isx = e < m * n;
                const int q = isx ? (e & 7) : ((e - m * n) & 7), j = isx ? (e >> 3) : n + ((e - m * n) >> 3);
                double s = 0.0;
#pragma unroll 8
                for (int kq = 0; kq < n; ++kq) s += Fs[kq + LDK * (n + q)] * Fs[kq + LDK * j];
                if (isx) Xrs[e] = Xs[e] + lam * s;
                else Radd[e - m * n] = lam * s;
            }
            __syncthreads();
        }
        const double *Xr = (regType == 2) ? Xrs : Xs;

        if (wv == 0) {
            // ================= P3 (wave 0): Quu/Qu reduction, gains (backward_pass.jl:30-68) ===================
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int aq = l4 + 4 * r;
                const double s = ((Pq[r * 64 + lane] + Pq[(2 + r) * 64 + lane]) + Pq[(4 + r) * 64 + lane]) + Pq[(6 + r) * 64 + lane];
                if (l15 < m) Quus[aq + m * l15] = s + cuui[aq + m * l15];          // (:209)
                else if (l15 == m) Qs[n + aq] = s + cu[(size_t)m * i + aq];       // (:204)
            }
            wave_sync();
            double H[m * m], R[m * m], kk[m];
            unsigned clamped = 0u;
#pragma unroll
            for (int e = 0; e < m * m; ++e) H[e] = Quus[e];
            if (regType == 2) {
#pragma unroll
                for (int e = 0; e < m * m; ++e) H[e] += Radd[e];
            } else {
#pragma unroll
                for (int q = 0; q < m; ++q) H[q + m * q] += lam;
            }
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
            if (lane == 0) flag[0] = fail ? 1.0 : 0.0;
            if (fail) {
                Quug[mm * i + lane] = Quus[lane];
            } else {
                {                                                    // K_i column `lane`, Y = Quu·K + 2·Qux
                    double col[m];
#pragma unroll
                    for (int q = 0; q < m; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : Xr[q + m * lane];
                    if (use_ri) ddp_rsolve_neg<m>(R, ri, col);
                    else {
                        chol_solve<m>(m, R, col);
#pragma unroll
                        for (int q = 0; q < m; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : -col[q];
                    }
#pragma unroll
                    for (int q = 0; q < m; ++q) {
                        double t = 2.0 * Xs[q + m * lane];
#pragma unroll
                        for (int q2 = 0; q2 < m; ++q2) t += Quus[q + m * q2] * col[q2];
                        Ks[q + m * lane] = col[q];
                        Ys[q + m * lane] = t;
                        Kg[nm * i + q + (size_t)m * lane] = col[q];  // (:76)
                    }
                }
                Quug[mm * i + lane] = Quus[lane];
                {                                                    // k_i, Quu·k, dV (:64-68)
                    const int q = lane & 7;
                    double t = 0.0;
#pragma unroll
                    for (int q2 = 0; q2 < m; ++q2) t += Quus[q + m * q2] * kk[q2];
                    if (lane < m) Quuks[q] = t;
                    wave_sync();
                    if (lane == 0) {
                        double kQu = 0.0, kQuuk = 0.0;
#pragma unroll
                        for (int q2 = 0; q2 < m; ++q2) {
                            ks[q2] = kk[q2];
                            kg[(size_t)m * i + q2] = kk[q2];
                            kQu += kk[q2] * Qs[n + q2]; kQuuk += kk[q2] * Quuks[q2];
                        }
                        dV0 += kQu; dV1 += 0.5 * kQuuk;
                    }
                }
            }
        } else {
            // ================= P2b (waves 1-3): the 10 upper Qxx tiles of G = F'W, + cxx, into Vs (:210) =======
            for (int t = wv - 1; t < 10; t += 3) {
                const int tj = (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0, ti = t - tj * (tj + 1) / 2;
                const int gj = 16 * tj + l15, gi0 = 16 * ti + l4;
                d4 acc0 = d4{cxxi[gi0 + n * gj], cxxi[gi0 + 4 + n * gj], cxxi[gi0 + 8 + n * gj], cxxi[gi0 + 12 + n * gj]};
                d4 acc1 = d4{0.0, 0.0, 0.0, 0.0};
                const double *ap = Fs + l4 + LDK * (16 * ti + l15);   // A[i][k] = F[k, 16ti+i]
                const double *bp = WT + gj + LD * l4;                 // B[k][j] = W[k, 16tj+j]
#pragma unroll
                for (int kk = 0; kk < 16; kk += 2) {
                    acc0 = mf(ap[4 * kk], bp[LD * 4 * kk], acc0);
                    acc1 = mf(ap[4 * kk + 4], bp[LD * 4 * kk + LD * 4], acc1);
                }
                const d4 g = acc0 + acc1;
                double *qp = Vs + gj + LD * gi0;                      // Qxx[gi, gj] stored at (gj, gi): lanes contiguous
                qp[0] = g.x; qp[LD * 4] = g.y; qp[LD * 8] = g.z; qp[LD * 12] = g.w;
            }
        }
        __syncthreads();
        if (flag[0] != 0.0) { diverge = i + 1; break; }              // block-uniform
        if (FXTV && i > 0) store_F(pfF);                             // Fs is dead from here on

        // ================= P4: Vxx_i = Qxx + ½(K'Y + Y'K), symmetrised (:69-72); Vx_i ==========================
        for (int t = wv; t < 10; t += 4) {
            const int tj = (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0, ti = t - tj * (tj + 1) / 2;
            const int gj = 16 * tj + l15, gi0 = 16 * ti + l4;
            double *qp = Vs + gj + LD * gi0;
            d4 acc = d4{qp[0], qp[LD * 4], qp[LD * 8], qp[LD * 12]};
            const int ia = l4 + m * (16 * ti + l15), ib = l4 + m * gj;
            acc = mf(Ks[ia], 0.5 * Ys[ib], acc);
            acc = mf(Ks[ia + 4], 0.5 * Ys[ib + 4], acc);
            acc = mf(Ys[ia], 0.5 * Ks[ib], acc);
            acc = mf(Ys[ia + 4], 0.5 * Ks[ib + 4], acc);
            double *mp = Vs + gi0 + LD * gj;                          // mirror position (gi, gj)
            if (ti == tj) {                                           // both halves exist: average them like the reference
                qp[0] = acc.x; qp[LD * 4] = acc.y; qp[LD * 8] = acc.z; qp[LD * 12] = acc.w;
                wave_sync();
                const double u0 = mp[0], u1 = mp[4], u2 = mp[8], u3 = mp[12];
                wave_sync();
                qp[0] = 0.5 * (acc.x + u0); qp[LD * 4] = 0.5 * (acc.y + u1); qp[LD * 8] = 0.5 * (acc.z + u2); qp[LD * 12] = 0.5 * (acc.w + u3);
            } else {
                qp[0] = acc.x; qp[LD * 4] = acc.y; qp[LD * 8] = acc.z; qp[LD * 12] = acc.w;
                mp[0] = acc.x; mp[4] = acc.y; mp[8] = acc.z; mp[12] = acc.w;
            }
        }
        if (wv == 3) {                                               // Vx_i (:69)
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int q = 0; q < m; ++q) {
                s1 += Ks[q + m * lane] * Quuks[q];
                s2 += Ks[q + m * lane] * Qs[n + q];
                s3 += Xs[q + m * lane] * ks[q];
            }
            const double v = ((Qs[lane] + s1) + s2) + s3;
            vs[lane] = v; Vxg[(size_t)n * i + lane] = v;
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
    } else {
        for (int e = tid; e < n * n; e += NT) Vxxg[e] = Vs[(e & 63) + LD * (e >> 6)];
    }
    if (tid == 0) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
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
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (d->has_lims) hipLaunchKernelGGL(back_pass_mfma_kernel<true>, dim3(d->B), dim3(NT), shmem, h->stream, a);
    else hipLaunchKernelGGL(back_pass_mfma_kernel<false>, dim3(d->B), dim3(NT), shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}
