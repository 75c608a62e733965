// back_pass_mfma_kernel.h — backward pass for the BASELINE config-4 shape n = 64, m = 8 with every product of the
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
//        Qxx tile as the C operand; the upper triangle is mirrored (exactly symmetric Vxx).
// Measured (profiles/microbench/mfma_f64_bench.hip): 30 ns per MFMA per wave with ONE wave per SIMD (66-70 TF/s of the
// 78.6 TF/s peak) — unlike the fp64 VALU, the matrix pipe does not need several waves to fill.
// Included by back_pass_mfma.hip (no control limits; built with -amdgpu-mfma-vgpr-form) and back_pass_mfma_lims.hip
// (boxQP variant; its register pressure crashes the compiler's AGPR rewrite pass under that flag, so it is a
// separate translation unit built without it).
#pragma once
#include "ddp_internal.h"
#include "boxqp_dev.h"

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
constexpr int NT = 256, n = 64, m = 8, p = 72, PP = 80, LD = 80, LDK = 66, LDV = 65, KS = 10;
constexpr int oVs = 0, oFs = oVs + n * LDV, oWT = oFs + PP * LDK, ovs = oWT + n * LD, oQs = ovs + n, oXs = oQs + PP,
              oXrs = oXs + m * n, oQuus = oXrs + m * n, oRadd = oQuus + m * m, oKs = oRadd + m * m, oYs = oKs + KS * n,
              oks = oYs + KS * n, oQuuks = oks + m, oPq = oQuuks + m, oFlag = oPq + 4 * 2 * 64, oTot = oFlag + 2;

#ifdef DDP_MFPROF     // per-phase cycle counts (s_memtime) of block 0, printed per wave: profiling builds only
#define MFP_DECL long long mfp_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mfp_t = __builtin_amdgcn_s_memtime()
#define MFP(k) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_amdgcn_s_memtime(); mfp_[k] += t_ - mfp_t; mfp_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define MFP_PRINT do { if (b == 0 && lane == 0) printf("MFPROF wave %d steps %d: p1 %lld bar %lld p2a %lld bar %lld p3|p2b %lld bar %lld p4 %lld bar %lld | p1: gemm %lld wst %lld; p4: stF %lld tiles %lld; p3: red+H %lld chol+k %lld Ksolve %lld\n", wv, N - 1, \
    mfp_[0] / (N - 1), mfp_[1] / (N - 1), mfp_[2] / (N - 1), mfp_[3] / (N - 1), mfp_[4] / (N - 1), mfp_[5] / (N - 1), mfp_[6] / (N - 1), mfp_[7] / (N - 1), mfp_[8] / (N - 1), mfp_[9] / (N - 1), mfp_[10] / (N - 1), mfp_[11] / (N - 1), mfp_[12] / (N - 1), mfp_[13] / (N - 1), mfp_[14] / (N - 1)); } while (0)
#else
#define MFP_DECL
#define MFP(k)
#define MFP_PRINT
#endif

__device__ __forceinline__ d4 mf(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }

// Two independent NK-step MFMA chains; the four operands of a k-step are fetched from LDS PF steps ahead of their use
// (the scheduler otherwise issues each ds_read right before its MFMA and exposes the LDS latency on every step).
template <int NK, int PF, int SA, int SB>
__device__ __forceinline__ void mfma_chain2(const double *aA, const double *bA, const double *aB, const double *bB, d4 &cA, d4 &cB)
{
    double r[PF + 1][4];
#pragma unroll
    for (int j = 0; j < PF; ++j) { r[j][0] = aA[SA * j]; r[j][1] = bA[SB * j]; r[j][2] = aB[SA * j]; r[j][3] = bB[SB * j]; }
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        if (kk + PF < NK) {
            const int j = (kk + PF) % (PF + 1);
            r[j][0] = aA[SA * (kk + PF)]; r[j][1] = bA[SB * (kk + PF)]; r[j][2] = aB[SA * (kk + PF)]; r[j][3] = bB[SB * (kk + PF)];
        }
        const int c = kk % (PF + 1);
        cA = mf(r[c][0], r[c][1], cA);
        cB = mf(r[c][2], r[c][3], cB);
    }
}

template <bool LIMS>
__global__ __launch_bounds__(NT) void back_pass_mfma_kernel(BPMArgs a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.active && a.active[b] == 0) return;
    const int N = a.N, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
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
    auto tile_of = [](int t, int &ti, int &tj) { tj = (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0; ti = t - tj * (tj + 1) / 2; };
    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    double cxxr[4][4], pre2a[4], preq[2];             // cost-Hessian operands of this thread's tiles (reloaded per step only if CTV)
    MFP_DECL;
    for (int i = N - 2; i >= 0; --i) {
        const double *cxxi = cxx + (CTV ? nn * i : 0), *cxui = cxu + (CTV ? nm * i : 0), *cuui = cuu + (CTV ? mm * i : 0);
        const bool ldF = FXTV && i > 0;                 // next step's Jacobian is fetched under this step's first product
        if (CTV || i == N - 2) {                        // so do the cost terms: they become the C operands of the G tiles
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 16 * wv + l4 + 4 * r;
                pre2a[r] = (l15 < m) ? cxui[gi + n * l15] : 0.0;
            }
            if (wv == 0) {
#pragma unroll
                for (int r = 0; r < 2; ++r) preq[r] = (l15 < m) ? cuui[l4 + 4 * r + m * l15] : 0.0;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int ti, tj;
                    tile_of(min(wv - 1 + 3 * u, 9), ti, tj);
                    const double *cp = cxxi + 16 * ti + l4 + n * (16 * tj + l15);
                    cxxr[u][0] = cp[0]; cxxr[u][1] = cp[4]; cxxr[u][2] = cp[8]; cxxr[u][3] = cp[12];
                }
            }
        }
        double gx[4] = {0.0, 0.0, 0.0, 0.0}, gu[2] = {0.0, 0.0};   // gradient entries riding in column 72: cx (rows of this wave), cu
        // ================= P1: W = Vxx·F on the matrix cores; column 72 of W := Vx ===========================
        {
            d4 acc[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
            const double *ap = Vs + 16 * wv + l15 + LDV * l4;          // A[i][k] = Vxx[16w+i, k]
            const double *bp = Fs + l4 + LDK * l15;                   // B[k][j] = F[k, 16c+j]
            double a0 = ap[0], b0[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) b0[c] = bp[LDK * 16 * c];
            const double *vout = Vs + lane + LDV * wv;                // Vxx_{i+1} streams out under the MFMAs (:72 of step i+1)
            double *gout = Vxxg + nn * (i + 1) + tid;
            double vprev = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                if (kk > 0) gout[NT * (kk - 1)] = vprev;
                vprev = vout[LDV * 4 * kk];
                // global loads ride in the MFMA shadow too, one per k-step: the address unit takes ~16 cycles per wave
                // instruction and all four waves share it, so a burst at the top of the step costs ~2k cycles
                if (kk < 4) { if (l15 == m) gx[kk] = cx[(size_t)n * i + 16 * wv + l4 + 4 * kk]; }        // needed first (P2a)
                else if (kk < 6) { if (wv == 0 && l15 == m) gu[kk - 4] = cu[(size_t)m * i + l4 + 4 * (kk - 4)]; }
                else if (kk < 6 + RF) { if (ldF) pfF[kk - 6] = load_F1(i - 1, kk - 6); }
                double a1 = 0.0, b1[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                if (kk < 15) {
                    a1 = ap[LDV * 4 * (kk + 1)];
#pragma unroll
                    for (int c = 0; c < 5; ++c) b1[c] = bp[LDK * 16 * c + 4 * (kk + 1)];
                }
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[c] = mf(a0, b0[c], acc[c]);
                a0 = a1;
#pragma unroll
                for (int c = 0; c < 5; ++c) b0[c] = b1[c];
            }
            MFP(8);
            gout[NT * 15] = vprev;
            const bool vcol = (l15 == 8);
#pragma unroll
            for (int c = 0; c < 5; ++c) {                             // D[row = l4 + 4r][col = l15] -> WT[col + LD*row]
                double *wp = WT + 16 * c + l15 + LD * (16 * wv + l4);
                double w0 = acc[c].x, w1 = acc[c].y, w2 = acc[c].z, w3 = acc[c].w;
                if (c == 4 && vcol) { const double *vp = vs + 16 * wv + l4; w0 = vp[0]; w1 = vp[4]; w2 = vp[8]; w3 = vp[12]; }
                wp[0] = w0; wp[LD * 4] = w1; wp[LD * 8] = w2; wp[LD * 12] = w3;
            }
        }
        MFP(9);
        MFP(0);
        __syncthreads();
        MFP(1);

        // ================= P2a: the u/Vx columns of G = F'W: Qux' (:208), Qx (:203), partial Quu/Qu ===========
        {
            d4 acc0 = d4{pre2a[0] + gx[0], pre2a[1] + gx[1], pre2a[2] + gx[2], pre2a[3] + gx[3]}, acc1 = d4{0.0, 0.0, 0.0, 0.0}, accq = d4{0.0, 0.0, 0.0, 0.0};
            const double *ap = Fs + l4 + LDK * (16 * wv + l15);       // A[i][k] = F[k, 16w+i]
            const double *aq = Fs + l4 + LDK * (n + l15) + 16 * wv;   // A[i][k] = F[k, 64+i], k-range [16w, 16w+16)
            const double *bp = WT + n + l15 + LD * l4;                // B[k][j] = W[k, 64+j]
            double qa[4], qb[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) { qa[k2] = aq[4 * k2]; qb[k2] = bp[LD * 4 * (4 * wv + k2)]; }
            mfma_chain2<8, 2, 8, LD * 8>(ap, bp, ap + 4, bp + LD * 4, acc0, acc1);    // even / odd k-steps
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) accq = mf(qa[k2], qb[k2], accq);
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
        MFP(2);
        __syncthreads();
        MFP(3);
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
                if (l15 < m) Quus[aq + m * l15] = s + preq[r];                     // (:209)
                else if (l15 == m) Qs[n + aq] = s + gu[r];                        // (:204)
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
            MFP(12);
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
                const int result = boxqp_dev_ri<m>(m, H, g, lo, up, x0, qpo, kk, R, ri, clamped, iters);
                fail = (result < 1);
            }
            MFP(13);
            if (lane == 0) flag[0] = fail ? 1.0 : 0.0;
            if (fail) {
                Quug[mm * i + lane] = Quus[lane];
            } else {
                // every LDS read first, every write last: the compiler cannot prove the K/Y writes do not alias Quus
                double col[m], x2[m], qu[m];
#pragma unroll
                for (int q = 0; q < m; ++q) { x2[q] = Xs[q + m * lane]; col[q] = ((clamped >> q) & 1u) ? 0.0 : Xr[q + m * lane]; qu[q] = Qs[n + q]; }
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
        } else {
            // ================= P2b (waves 1-3): the 10 upper Qxx tiles of G = F'W, + cxx, into Vs (:210) =======
#pragma unroll
            for (int u = 0; u < 4; u += 2) {                          // two tiles at a time: two independent MFMA chains
                const int tA = wv - 1 + 3 * u, tB = tA + 3;           // tA <= 8 always, tB may run past the last tile
                const bool vB = tB < 10;
                int tiA, tjA, tiB, tjB;
                tile_of(tA, tiA, tjA);
                tile_of(vB ? tB : tA, tiB, tjB);
                d4 accA = d4{cxxr[u][0], cxxr[u][1], cxxr[u][2], cxxr[u][3]};
                d4 accB = d4{cxxr[u + 1][0], cxxr[u + 1][1], cxxr[u + 1][2], cxxr[u + 1][3]};
                const double *apA = Fs + l4 + LDK * (16 * tiA + l15), *apB = Fs + l4 + LDK * (16 * tiB + l15);   // A[i][k] = F[k, 16ti+i]
                const double *bpA = WT + 16 * tjA + l15 + LD * l4, *bpB = WT + 16 * tjB + l15 + LD * l4;         // B[k][j] = W[k, 16tj+j]
                mfma_chain2<16, 2, 4, LD * 4>(apA, bpA, apB, bpB, accA, accB);
                double *qA = Vs + 16 * tjA + l15 + LDV * (16 * tiA + l4);   // Qxx[gi, gj] stored at (gj, gi): lanes contiguous
                qA[0] = accA.x; qA[LDV * 4] = accA.y; qA[LDV * 8] = accA.z; qA[LDV * 12] = accA.w;
                if (vB) {
                    double *qB = Vs + 16 * tjB + l15 + LDV * (16 * tiB + l4);
                    qB[0] = accB.x; qB[LDV * 4] = accB.y; qB[LDV * 8] = accB.z; qB[LDV * 12] = accB.w;
                }
            }
        }
        MFP(4);
        __syncthreads();
        MFP(5);
        if (flag[0] != 0.0) { diverge = i + 1; break; }              // block-uniform
        if (FXTV && i > 0) store_F(pfF);                             // Fs is dead from here on
        MFP(10);

        // ================= P4: Vxx_i = Qxx + ½(K'Y + Y'K), symmetrised (:69-72); Vx_i ==========================
        {
            d4 acc[3];
            double kA[3][2], yA[3][2], kB[3][2], yB[3][2], *qp[3], *mp[3];
            bool diag[3], valid[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {                             // all operands of this wave's (up to) 3 tiles first ...
                const int t = wv + 4 * u;
                valid[u] = t < 10;
                int ti, tj;
                tile_of(valid[u] ? t : 9, ti, tj);
                diag[u] = ti == tj;
                const int gj = 16 * tj + l15, gi0 = 16 * ti + l4;
                qp[u] = Vs + gj + LDV * gi0;
                mp[u] = Vs + gi0 + LDV * gj;                          // mirror position (gi, gj)
                acc[u] = d4{qp[u][0], qp[u][LDV * 4], qp[u][LDV * 8], qp[u][LDV * 12]};
                const int ia = l4 + KS * (16 * ti + l15), ib = l4 + KS * gj;
                kA[u][0] = Ks[ia]; kA[u][1] = Ks[ia + 4]; yA[u][0] = Ys[ia]; yA[u][1] = Ys[ia + 4];
                kB[u][0] = 0.5 * Ks[ib]; kB[u][1] = 0.5 * Ks[ib + 4]; yB[u][0] = 0.5 * Ys[ib]; yB[u][1] = 0.5 * Ys[ib + 4];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) acc[u] = mf(kA[u][0], yB[u][0], acc[u]);      // ... then three interleaved chains
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
        MFP(11);
        if (wv == 3) {                                               // Vx_i (:69)
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

template <bool LIMS>
static int ddp_bpm_launch(ddp_handle h, const BPMArgs &a)
{
    const size_t shmem = (size_t)oTot * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mfma_kernel<LIMS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(back_pass_mfma_kernel<LIMS>, dim3(a.B), dim3(NT), shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}

int ddp_bpm_launch_lims(ddp_handle h, const BPMArgs &a);      // back_pass_mfma_lims.hip
