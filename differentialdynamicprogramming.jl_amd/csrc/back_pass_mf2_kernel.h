// back_pass_mf2_kernel.h — backward pass for LARGE STATES, 32 < n <= 64, m <= 8 (src/backward_pass.jl:162-215 + :28-79; BASELINE config 4
// is n = 64, m = 8), every product of the Riccati step on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), sizes at RUN TIME.
//
// Round 6 re-cut of back_pass_mfma_kernel.h (n = 64, m = 8 only; 32 < n < 64 went through padded COPIES of every operand and result):
//   * n, m are run-time values.  The state is padded to NP = 16 NT (NT = 3 or 4 tiles, a template parameter) and the controls to 8 inside
//     the LDS only: rows / columns past n of Vxx, F = [fx fu] and the cost terms are exact zeros there, the padded controls get an
//     identity block of Quu (their gains are exact zeros), products skip the k-steps past NP.  Nothing is padded in global memory.
//   * W = Vxx·fx never goes through the LDS.  Wave c owns COLUMN TILE c of W (NT x 4 NT products, the B operand shared by the NT row
//     tiles) and keeps it in its accumulators; in the accumulator layout a tile of W IS the B operand of four k-steps of F'W (k permuted
//     the same way on the A side), so the tiles P[ti][c] = cxx + fx'W of that column are computed straight from the registers with ONE
//     LDS read per product (the round-5 kernel: W through a 41 KB LDS image, two reads per product).  Vxx is symmetric, so a wave may
//     compute tile (ti, c) for ANY ti — upper or lower — and mirror it: the ten (six) tiles are spread over the column owners.
//   * phases:  A  every wave w < NT: rows 16w.. of W for the column tiles {u | Vx} and 0 (A operand shared); the u tile stays in the
//                 registers as the B operand of this wave's k-slice of G[:, u | Vx] = F'W_u; column tile 0 goes to a 8 KB LDS image
//              -- barrier --
//              B  wave 0: sums the partial tiles, gains (Cholesky or box-QP), Vx_i, then tile (0,0) from the image  ||  wave c >= 1: its
//                 column of W, its two tiles cxx + fx'W from the registers, tile (c, 0) from the image; in the shadow of these products
//                 FIRST the next Jacobian (loads), THEN Vxx_{i+1} out (stores)
//              -- barrier (gains known, everybody is done reading Vxx_{i+1} and F) --
//              C  the rank-16 update ½(K'Y + Y'K) on the same accumulators (½K, ½Y halved once by the gain wave), tiles mirrored into
//                 Vxx; K_i out through its LDS image (2 x 256 consecutive doubles); the next Jacobian goes to the LDS
//              -- barrier --
//     three barriers per step (four before); the result tiles wait in registers for the second barrier, so Vxx needs no second buffer.
//   * what the phase profile (-DDDP_MF2PROF, profiles/r06_mf2_phases.txt) taught on the way from 9.2 to 7.5 ms at C4:
//       - a VECTOR instruction between two fp64 products does not hide in the product's shadow, it adds its time (35 ticks per product
//         with one address add per load): every piece of side traffic is scalar base + constant 32-bit lane offset (loads, stores) or
//         per-wave base + immediate (LDS), the lane offset re-introduced inside the block so that instruction selection folds it;
//       - a select right behind a load is a wait for HBM: raw values, selects at the use (gradients, cost terms: 2 300 ticks per step);
//       - address arithmetic hoisted out of the time loop took 600 scalar registers (v_writelane / v_readlane spills inside the
//         chains): `n` and the wave number are made opaque once per step;
//       - one copy of the product code for the waves 1..3 (run-time tile slots): per-wave instantiation was 64 KB of loop body;
//       - K_i leaves through the LDS (the gain wave stored it with 8 strided 8-byte stores per lane), the Jacobian's LDS write is one
//         instruction per column.
// Same arithmetic and failure semantics as the other backward kernels; included by back_pass_mf2.hip (no limits, built with
// -amdgpu-mfma-vgpr-form) and back_pass_mf2_lims.hip (box-QP instantiations).
#pragma once
#include "ddp_internal.h"
#include "boxqp_dev.h"
#include <type_traits>
#include "boxqp_rows.h"

struct BPM2Args {
    int n, m, N, B;
    int fx_tv, fx_batched, cost_tv, cost_batched, regType, has_lims;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
    double *sink;                 // 4 KB of device memory that lanes without a result may write (stores without exec-mask branches)
};

namespace mf2 {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NTH = 256, NX = 64, MX = 8, LDV = 66, LDK = 66, KSD = 10, PTS = 36, FC = 80;
// LDS image (doubles)
constexpr int oVs = 0, oFs = oVs + NX * LDV, oW0 = oFs + FC * LDK, ovs = oW0 + NX * 16, oQs = ovs + NX, oXs = oQs + FC, oXadd = oXs + MX * NX,
              oQuus = oXadd + MX * NX, oRadd = oQuus + MX * MX, oKs = oRadd + MX * MX, oYs = oKs + KSD * NX, oKh = oYs + KSD * NX, oYh = oKh + KSD * NX, oks = oYh + KSD * NX,
              oQuuks = oks + MX, oPT = oQuuks + MX, oFlag = oPT + 4 * 5 * 4 * PTS, oTot = oFlag + 2;

#define MF2_SB __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ d4 mf(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }
using bqr::sfor;
template <int I> using ic = std::integral_constant<int, I>;

#ifdef DDP_MF2PROF     // per-phase cycle counts (s_memtime) of work-group 0, printed per wave: profiling builds only
#define MFP_DECL long long mfp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mfp_t = __builtin_amdgcn_s_memtime()
#define MFP(k) do { MF2_SB; const long long t_ = __builtin_amdgcn_s_memtime(); mfp_[k] += t_ - mfp_t; mfp_t = t_; MF2_SB; } while (0)
#define MFP_PRINT do { if (b == 0 && lane == 0) printf("MF2PROF wave %d steps %d: top %lld A1 %lld A2 %lld Atail %lld bar %lld | B: col/gains %lld own/img0 %lld img %lld bar %lld | C %lld fstore %lld bar %lld\n", wv, N - 1, \
    mfp_[9] / (N - 1), mfp_[0] / (N - 1), mfp_[1] / (N - 1), mfp_[10] / (N - 1), mfp_[2] / (N - 1), mfp_[3] / (N - 1), mfp_[4] / (N - 1), mfp_[5] / (N - 1), mfp_[6] / (N - 1), mfp_[11] / (N - 1), mfp_[7] / (N - 1), mfp_[8] / (N - 1)); } while (0)
#else
#define MFP_DECL
#define MFP(k)
#define MFP_PRINT
#endif

// NCH product chains that share one operand (the A operand if SHA, else B), operands fetched PF k-steps ahead; `side(ic<s>)` is called
// once behind every product (s = S0, S0 + 1, ...): the piece of global / LDS traffic that issues in that product's 64-cycle shadow.
template <int NK, int NCH, int S0, bool SHA, class FS, class FO, class SIDE>
__device__ __forceinline__ void chains(FS sh, FO op, d4 *c, SIDE side)
{
    constexpr int PF = 2;
    double rs[PF + 1], ro[PF + 1][NCH];
    sfor<0, PF>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NK) { rs[j] = sh(ic<j>{}); sfor<0, NCH>([&](auto uc) { ro[j][decltype(uc)::value] = op(uc, ic<j>{}); }); }
    });
    sfor<0, NK>([&](auto kc) {
        constexpr int kk = decltype(kc)::value, q = kk % (PF + 1), j = (kk + PF) % (PF + 1);
        sfor<0, NCH>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            MF2_SB;
            c[u] = SHA ? mf(rs[q], ro[q][u], c[u]) : mf(ro[q][u], rs[q], c[u]);
            MF2_SB;
            if constexpr (kk + PF < NK) {
                if constexpr (u == 0) rs[j] = sh(ic<kk + PF>{});
                ro[j][u] = op(uc, ic<kk + PF>{});
            }
            side(ic<S0 + kk * NCH + u>{});
        });
    });
    MF2_SB;
}

// Which tiles of Vxx_i a wave computes — RUN-TIME data, so that the waves 1..3 share one copy of the code (the first form of this kernel
// instantiated every chain per wave: 64 KB of loop body, the size of the instruction cache).  Four accumulator slots per wave:
//   slots 0, 1  "own": tile (ti, c) of the wave's own column c, B operand = its W column in registers
//   slots 2, 3  "image": tile (ti, 0), B operand from the LDS image of W[:, 0..15]
// Vxx is symmetric: a wave may compute tile (ti, c) for any ti and mirror it.
//   NT = 4: wave c = 1..3: (c, c), (c % 3 + 1, c) | (c, 0);  (0, 0): wave 0 behind the gains — with limits (the QP is the long pole) wave 3
//   NT = 3: wave 1: (1,1) (2,1); wave 2: (2,2) (0,2); wave 3: (0,0) (1,0) from the image; wave 0: the gains only
struct Slots { int ti[4], tc[4]; bool valid[4]; };
template <int NT, bool LIMS> __device__ __forceinline__ Slots make_slots(int w)
{
    Slots s;
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.ti[u] = 0; s.tc[u] = 0; s.valid[u] = false; }
    if (NT == 4) {
        if (w == 0) { s.valid[2] = !LIMS; }                                          // (0, 0)
        else {
            s.ti[0] = w; s.tc[0] = w; s.ti[1] = w % 3 + 1; s.tc[1] = w; s.valid[0] = s.valid[1] = true;
            s.ti[2] = w; s.valid[2] = true;                                          // (w, 0)
            s.valid[3] = LIMS && w == 3;                                             // (0, 0)
        }
    } else {
        if (w == 1) { s.ti[0] = 1; s.tc[0] = 1; s.ti[1] = 2; s.tc[1] = 1; s.valid[0] = s.valid[1] = true; }
        if (w == 2) { s.ti[0] = 2; s.tc[0] = 2; s.ti[1] = 0; s.tc[1] = 2; s.valid[0] = s.valid[1] = true; }
        if (w == 3) { s.ti[2] = 0; s.ti[3] = 1; s.valid[2] = s.valid[3] = true; }
    }
    return s;
}
template <int NT, bool LIMS> constexpr bool has_slot3() { return NT == 3 || LIMS; }

template <int NT, bool LIMS, bool CTV, bool PAIR>
__global__ __launch_bounds__(NTH) void back_pass_mf2_kernel(BPM2Args a)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.active && a.active[b] == 0) return;
    constexpr int KS = 4 * NT;                         // k-steps of a product over the (padded) state index
    constexpr int m8 = MX;
    const int n = a.n, m = a.m, N = a.N;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Vs = lds + oVs, *Fs = lds + oFs, *W0 = lds + oW0, *vs = lds + ovs, *Qs = lds + oQs, *Xs = lds + oXs, *Xadd = lds + oXadd,
           *Quus = lds + oQuus, *Radd = lds + oRadd, *Ks = lds + oKs, *Ys = lds + oYs, *Kh = lds + oKh, *Yh = lds + oYh, *ks = lds + oks, *Quuks = lds + oQuuks,
           *PT = lds + oPT, *flag = lds + oFlag;

    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const int tvF = a.fx_tv ? 1 : 0;
    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    const double *ug = LIMS ? a.u + (size_t)m * N * b : nullptr;
    const double *fx = a.fx + (a.fx_batched ? nn * (tvF ? N : 1) * b : 0);
    const double *fu = a.fu + (a.fx_batched ? nm * (tvF ? N : 1) * b : 0);
    const double *cxx = a.cxx + (a.cost_batched ? nn * (CTV ? N : 1) * b : 0);
    const double *cxu = a.cxu + (a.cost_batched ? nm * (CTV ? N : 1) * b : 0);
    const double *cuu = a.cuu + (a.cost_batched ? mm * (CTV ? N : 1) * b : 0);
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + mm * N * b,
           *Vxg = a.Vx + (size_t)n * N * b, *Vxxg = a.Vxx + nn * N * b;
    double *sink = a.sink + lane;                      // where lanes without a result store
    const double lam = a.lambda[b];
    const int regType = a.regType;
    bool nolims = true;
    if (LIMS) nolims = a.lims[0] > a.lims[m];          // backward_pass.jl:31
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35
    // the QP runs with one coordinate per lane (boxqp_rows.h): this lane's bounds, and u[:, i] one step ahead of its use.  The padded
    // coordinates (m..7) are free inside [-1, 1] around 0: with their identity block of Quu they stay at 0.
    const int qcoord = (l15 < m8) ? l15 : 0;
    const bool qreal = qcoord < m;
    double qlo = -1.0, qhi = 1.0, qu_next = 0.0;
    if (LIMS) {
        if (qreal) { qlo = nolims ? -HUGE_VAL : a.lims[qcoord]; qhi = nolims ? HUGE_VAL : a.lims[qcoord + m]; }
        if (N >= 2 && qreal) qu_next = ug[(size_t)m * (N - 2) + qcoord];
    }

    // ---- the LDS image starts as zeros: the padding never changes
    for (int e = tid; e < oTot; e += NTH) lds[e] = 0.0;
    __syncthreads();
    // ---- terminal step (backward_pass.jl:197-199); Vxx_{N-1} itself is streamed out by the first step below
    {
        const double *cT = cxx + (CTV ? nn * (N - 1) : 0);
        for (int e = tid; e < n * n; e += NTH) { const int r = e % n, c = e / n; Vs[r + LDV * c] = cT[e]; }
        if (tid < n) { const double v = cx[(size_t)n * (N - 1) + tid]; vs[tid] = v; Vxg[(size_t)n * (N - 1) + tid] = v; }
        if (tid < m * m) Quug[mm * (N - 1) + tid] = cuu[(CTV ? mm * (N - 1) : 0) + tid];
        for (int e = tid; e < m * n; e += NTH) Kg[nm * (N - 1) + e] = 0.0;
        if (tid < m) kg[(size_t)m * (N - 1) + tid] = 0.0;
        if (N < 2) {
            for (int e = tid; e < n * n; e += NTH) Vxxg[nn * (N - 1) + e] = cT[e];
            if (tid == 0) { a.dV[2 * b] = 0.0; a.dV[2 * b + 1] = 0.0; a.diverge[b] = 0; }
            return;
        }
    }
    // ---- the Jacobian [fx fu] of a step comes in through the waves 1..3 (wave 0 is the serial gain wave: nothing of this may live in its
    // registers): fx column c = 3 q + wave - 1 (q < NFX), fu column c = 3 q' + wave - 1 (q' < 3), lane = row.  Every load has a wave-uniform
    // base and a 32-bit lane offset; rows past n and columns past n / m are CLAMPED to the last one (a valid address, the value is
    // dropped) and keep the zero of the padding in the LDS.
    // PAIR (n even): 16 bytes per lane — lane (p, h) = (lane & 31, lane >> 5) holds rows 2p, 2p + 1 of column 2 j + h, so ONE instruction
    // moves the column pair j = 3 q + wave - 1 (1 KB): half the vector-memory instructions and half the LDS writes of the 8-byte form
    // (~40 ticks each inside the product chains, where the address unit is shared by four waves).  fu (m may be odd) keeps 8-byte columns.
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int NFX = PAIR ? (8 * NT + 2) / 3 : (16 * NT + 2) / 3, NFL = NFX + 3;
    const int rowc = lane < n ? lane : n - 1;
    const bool rowv = lane < n;
    const unsigned row8 = 8u * (unsigned)rowc;
    const int pp_ = lane & 31, hh_ = lane >> 5;
    const bool rowvP = 2 * pp_ < n;
    const unsigned row16 = 8u * (unsigned)((2 * pp_ < n ? 2 * pp_ : n - 2) + n * hh_);
    const int fw = wv > 0 ? wv - 1 : 0;
    // (the column part of the address is wave-uniform and goes into the scalar base, the lane part is a constant 32-bit offset: NO vector
    // instruction per load — between two fp64 products a vector instruction does not hide, it adds its time: 35 ticks per product)
    auto f_load8 = [&](const char *base, int nv, int c, int lim, unsigned r8) -> double {                   // 8-byte form: one column
        const char *colb = base + (size_t)(unsigned)(8 * nv * (c < lim ? c : lim - 1));                    // (a column past n / m: the last one, dropped below)
        return *(const double *)(colb + r8);
    };
    auto f_loadP = [&](const char *fxb, int nv, int q, unsigned r16, int fw) -> d2 {                        // 16-byte form: fx column pair
        const int j = 3 * q + fw, hv = nv >> 1;
        const char *colb = fxb + (size_t)(unsigned)(16 * nv * (j < hv ? j : hv - 1));
        return *(const d2 *)(colb + r16);
    };
    // One ds_write per column (pair), address = a per-wave base + an immediate (the first form: compare, branch, multiply, add per column
    // — 48 ticks each).  A column past n / m is written as zeros (a wave-uniform select on the data; it was loaded from a clamped
    // address); rows past n (lanes) are not written: they keep the zero of the initial image.  Only the last fx column (pair) of a wave
    // can lie past the image: it goes to the zero padding behind fu.
    double *const fdst = Fs + lane + LDK * fw;
    double *const fdstP = Fs + 2 * pp_ + LDK * (2 * fw + hh_);
    double pfF[PAIR ? 3 : NFL];                        // 8-byte pieces: [fx columns |] fu columns
    d2 pfP[PAIR ? NFX : 1];                            // 16-byte pieces: fx column pairs
    auto f_store = [&](int nvs, int fws) {
        if constexpr (PAIR) {
            if (rowvP) {
#pragma unroll
                for (int q = 0; q < NFX; ++q) {
                    const int j = 3 * q + fws;
                    const d2 v = 2 * j < nvs ? pfP[q] : d2{0.0, 0.0};
                    if (q < NFX - 1) *(d2 *)(fdstP + LDK * 6 * q) = v;
                    else { double *p_ = j < 8 * NT ? fdstP + LDK * 6 * q : Fs + 2 * pp_ + LDK * (FC - 2 + hh_); *(d2 *)p_ = v; }
                }
            }
            if (rowv) {
#pragma unroll
                for (int q = 0; q < 3; ++q) fdst[LDK * (NX + 3 * q)] = 3 * q + fws < m ? pfF[q] : 0.0;
            }
        } else {
            if (rowv) {
#pragma unroll
                for (int q = 0; q < NFL; ++q) {
                    const int c = 3 * (q < NFX ? q : q - NFX) + fws;
                    const double v = (q < NFX ? c < nvs : c < m) ? pfF[q] : 0.0;
                    if (q < NFX - 1) fdst[LDK * 3 * q] = v;
                    else if (q == NFX - 1) { double *p_ = c < 16 * NT ? fdst + LDK * 3 * q : Fs + lane + LDK * (FC - 1); *p_ = v; }
                    else fdst[LDK * (NX + 3 * (q - NFX))] = v;
                }
            }
        }
    };
    // piece s of the Jacobian of a wave (s < NFL): fx first, then the three fu columns
    auto f_piece = [&](auto sc, const char *fxb, const char *fub, int nv, unsigned r8, unsigned r16, int fwl) {
        constexpr int s_ = decltype(sc)::value;
        if constexpr (s_ >= NFX) pfF[PAIR ? s_ - NFX : s_] = f_load8(fub, nv, 3 * (s_ - NFX) + fwl, m, r8);
        else if constexpr (PAIR) pfP[s_] = f_loadP(fxb, nv, s_, r16, fwl);
        else pfF[s_] = f_load8(fxb, nv, 3 * s_ + fwl, nv, r8);
    };
    if (wv > 0) {
        const char *fxb = (const char *)(fx + nn * (tvF ? N - 2 : 0)), *fub = (const char *)(fu + nm * (tvF ? N - 2 : 0));
        sfor<0, NFL>([&](auto sc) { f_piece(sc, fxb, fub, n, row8, row16, fw); });
        f_store(n, fw);
    }
    __syncthreads();
    // K_i leaves through the LDS image, 256 consecutive doubles per store: element e = q + m j of K[m, n] (two per thread)
    int kls[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { const int e = tid + NTH * t, ec = e < m * n ? e : m * n - 1; kls[t] = ec % m + KSD * (ec / m); }

    const Slots sl = make_slots<NT, LIMS>(wv);
    constexpr int NSL = has_slot3<NT, LIMS>() ? 4 : 3;
    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    // cost-Hessian operands of this thread, RAW (reloaded per step only if CTV): the `valid ? v : 0` selects sit at the point of use — a
    // select right behind a load is a wait for HBM at the top of every step (2 300 of 21 000 ticks per step in the first form).
    // (Tried: fetching the cxx tiles inside phase B, in front of the products of the W column, to free their 24-32 registers in the gain
    // wave's code — the register allocator parks fresh loads in the accumulator file at once, i.e. waits for them: +1 500 ticks per step.)
    double cxxr[NSL][4], cxur[m8], preq[2];
    // tile (ti, tc) of slot u: C operand cxx[16 ti + l4 + 4 r, 16 tc + l15]
    auto cxx_ok = [&](int u, int r) { return sl.valid[u] && 16 * sl.ti[u] + l4 + 4 * r < n && 16 * sl.tc[u] + l15 < n; };
    auto load_cost_w = [&](int i) {
        const double *cxxi = cxx + (CTV ? nn * i : 0), *cxui = cxu + (CTV ? nm * i : 0), *cuui = cuu + (CTV ? mm * i : 0);
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int col = 16 * sl.tc[u] + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * sl.ti[u] + l4 + 4 * r;
                const bool ok = cxx_ok(u, r);
                cxxr[u][r] = cxxi[(ok ? row : 0) + (size_t)n * (ok ? col : 0)];
            }
        }
        if (wv == 0) {                                 // wave 0 adds cxu (column `lane`), cuu to the reduced partial tiles
#pragma unroll
            for (int q = 0; q < m8; ++q) cxur[q] = cxui[rowc + (size_t)n * (q < m ? q : 0)];
#pragma unroll
            for (int r = 0; r < 2; ++r) { const int aq = l4 + 4 * r; preq[r] = cuui[(aq < m && l15 < m) ? aq + m * l15 : 0]; }
        }
    };
    auto cxx_init = [&](int u) { return d4{cxx_ok(u, 0) ? cxxr[u][0] : 0.0, cxx_ok(u, 1) ? cxxr[u][1] : 0.0, cxx_ok(u, 2) ? cxxr[u][2] : 0.0, cxx_ok(u, 3) ? cxxr[u][3] : 0.0}; };
    if (!CTV) load_cost_w(0);
    // gradient entries cx[lane], cu of a step (wave 0 adds them to the reduced partial tiles), RAW, requested by the gain wave right
    // behind their use for the step after: nothing is (re-)initialised at the top of the loop — a `= 0.0` there overwrites a register
    // that a load of the previous iteration may still target on the paths that never read it, and the compiler answered with
    // s_waitcnt vmcnt(0) at the loop header: every wave waited for all its stores of the last step (500 ticks per step)
    double gxc = 0.0, gu[2] = {0.0, 0.0};
    auto load_grad = [&](int i) {
        gxc = *(const double *)((const char *)(cx + (size_t)n * i) + row8);
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int aq = l4 + 4 * r; gu[r] = cu[(size_t)m * i + (aq < m ? aq : 0)]; }
    };
    if (wv == 0) load_grad(N - 2);
    MFP_DECL;

    for (int i = N - 2; i >= 0; --i) {
        if (CTV) load_cost_w(i);
        const int inext = i > 0 ? i - 1 : 0;             // the next Jacobian (the last step reads its own again: no branch around the loads)
        int nv = n;                                      // the address arithmetic of a step stays inside the step: hoisted out of the loop it
        asm volatile("" : "+s"(nv));                     // took 600 scalar registers (spilled through v_writelane / v_readlane)
        char *goutb = (char *)(Vxxg + (size_t)(nv * nv) * (i + 1));
        const char *fxb = (const char *)(fx + (size_t)(nv * nv) * (tvF ? inext : 0)), *fub = (const char *)(fu + (size_t)(nv * m) * (tvF ? inext : 0));
        // side traffic, one piece per product of phase B (waves 1..3): FIRST the next Jacobian (NFL loads), THEN Vxx_{i+1} out — column
        // c = 3 q + wave - 1: an LDS read and (one product later) its store; rows / columns past n repeat the last one (the same value to
        // the same address: no mask, no second base).  Loads in front of the stores: the in-order vmcnt wait of the Jacobian in phase C
        // then never waits for a store to be acknowledged (with the stores in phase A it cost 2 500 ticks per step).
        constexpr int NVX = PAIR ? (8 * NT + 2) / 3 : (16 * NT + 2) / 3, VD = 3, NSIDE = 2 * NVX + VD;
        // (the LDS read of column (pair) j sits in slot 2 j, its store VD slots later: one product is not enough for the LDS round trip;
        // LDS address = a per-wave base + an immediate, global address = a scalar base per column + the lane offset: no vector
        // instruction; a column past n reads the zero padding and stores it into the sink)
        double vq[2] = {0.0, 0.0};
        d2 vqP[2] = {d2{0.0, 0.0}, d2{0.0, 0.0}};
        const double *vsrc = Vs + rowc + LDV * fw, *vsrcP = Vs + (2 * pp_ < n ? 2 * pp_ : n - 2) + LDV * (2 * fw + hh_);    // (rows past n repeat the last pair: the same value to the same address)
        // (the 32-bit lane offsets, re-introduced in the block of the products: instruction selection only folds `scalar base + zero-
        // extended vector offset` into ONE load when it sees the extension in the same block — hoisted out of the loop as a 64-bit
        // value it became a 64-bit vector add per load)
        unsigned r8 = row8, r16 = row16;
        asm volatile("" : "+v"(r8), "+v"(r16));
        int fwv = fw;                                    // (as nv: the 3 q + wave column numbers are not kept in 25 + 22 scalar registers)
        asm volatile("" : "+s"(fwv));
        const char *sinkb = (const char *)a.sink;
        auto sideA = [&](auto sc) {                                  // phase A (every wave: wave 0 repeats wave 1's loads and never reads them)
            if constexpr (decltype(sc)::value < NFL) f_piece(sc, fxb, fub, nv, r8, r16, fwv);
        };
        auto sideB = [&](auto sc) {
            constexpr int t = decltype(sc)::value;
            if constexpr (t < NSIDE) {
                if constexpr ((t & 1) == 0 && (t >> 1) < NVX) {
                    if constexpr (PAIR) vqP[(t >> 1) & 1] = *(const d2 *)(vsrcP + LDV * 6 * (t >> 1));
                    else vq[(t >> 1) & 1] = vsrc[LDV * 3 * (t >> 1)];
                }
                if constexpr (t >= VD && ((t - VD) & 1) == 0 && ((t - VD) >> 1) < NVX) {
                    constexpr int j = (t - VD) >> 1;
                    const int c = (PAIR ? 2 : 1) * (3 * j + fwv);
                    char *colb = c < nv ? goutb + (size_t)(unsigned)(8 * nv * c) : (char *)sinkb;
#if !(defined(MF2_EXP) && (MF2_EXP & 2))       // timing experiment 2: no Vxx stores (results invalid)
                    if constexpr (PAIR) *(d2 *)(colb + r16) = vqP[j & 1];
                    else *(double *)(colb + r8) = vq[j & 1];
#endif
                }
            }
        };
        auto sideB2 = [&](auto sc) { sideB(ic<decltype(sc)::value + KS * NT>{}); };        // continued behind the products of the own tiles
        auto no_side = [](auto) {};

        MFP(9);
        // ================= phase A: W[16w.., {u|Vx, 0}] = Vxx·F; partial G[:, u|Vx] from the registers ==========
        if (wv < NT) {
            d4 acc2[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};       // [0] the u | Vx tile, [1] column tile 0
            const double *ap = Vs + 16 * wv + l15 + LDV * l4;          // A[i][k] = Vxx[16w+i, k]
            const double *bp = Fs + l4 + LDK * l15;                   // B[k][j] = F[k, 16c+j]
            chains<KS, 2, 0, true>([&](auto kc) { return ap[LDV * 4 * decltype(kc)::value]; },
                                   [&](auto uc, auto kc) { return bp[(decltype(uc)::value == 0 ? LDK * NX : 0) + 4 * decltype(kc)::value]; }, acc2, sideA);
            static_assert(2 * KS >= NFL, "the Jacobian loads fit behind the products of phase A");
            MFP(0);
            {                                                         // D[row = l4 + 4r][col = l15] -> W0[col + 16 row]
                double *wp = W0 + l15 + 16 * (16 * wv + l4);
                wp[0] = acc2[1].x; wp[16 * 4] = acc2[1].y; wp[16 * 8] = acc2[1].z; wp[16 * 12] = acc2[1].w;
            }
            // The u|Vx tile of W as B operand: k-step r uses k = 16w + l4 + 4r, which is the accumulator register r of this lane
            double bu[4] = {acc2[0].x, acc2[0].y, acc2[0].z, acc2[0].w};
            if (l15 == m8) { const double *vp = vs + 16 * wv + l4; bu[0] = vp[0]; bu[1] = vp[4]; bu[2] = vp[8]; bu[3] = vp[12]; }   // column 72 := Vx_{i+1}
            d4 pg[NT + 1];
#pragma unroll
            for (int ti = 0; ti <= NT; ++ti) pg[ti] = d4{0.0, 0.0, 0.0, 0.0};
            const double *fp = Fs + 16 * wv + l4 + LDK * l15;         // A[i][k] = F[16w + l4 + 4r, 16ti + i]; the u tile is column tile 4
            chains<4, NT + 1, 2 * KS, false>([&](auto rc) { return bu[decltype(rc)::value]; },
                                             [&](auto tc, auto rc) { return fp[LDK * 16 * (decltype(tc)::value == NT ? 4 : decltype(tc)::value) + 4 * decltype(rc)::value]; }, pg, no_side);
            MFP(1);
            if (l15 <= m8) {                                           // columns u | Vx of the partial tiles, compact; the u tile is tile 4
                double *pp = PT + (wv * 5 * 4) * PTS + l4 * 9 + l15;
#pragma unroll
                for (int ti = 0; ti <= NT; ++ti) {
                    const int tq = ti == NT ? 4 : ti;
                    pp[(tq * 4 + 0) * PTS] = pg[ti].x; pp[(tq * 4 + 1) * PTS] = pg[ti].y;
                    if (ti < NT) { pp[(tq * 4 + 2) * PTS] = pg[ti].z; pp[(tq * 4 + 3) * PTS] = pg[ti].w; }     // u tile: rows 0..7 only
                }
            }
        } else {
            sfor<0, NFL>([&](auto sc) { sideA(sc); });              // NT = 3: wave 3 has no row tile, only its share of the Jacobian
        }
        if (regType == 2) {     // (:205-207): QuuF = Quu + λ·fu'fu, Qux_reg = Qux + λ·fu'fx — the λ terms only, added in phase B
            for (int e = tid; e < m8 * NX + m8 * m8; e += NTH) {
                const bool isx = e < m8 * NX;
                const int q = isx ? (e & 7) : ((e - m8 * NX) & 7), j = isx ? (e >> 3) : NX + ((e - m8 * NX) >> 3);
                double s = 0.0;
#pragma unroll 8
                for (int kq = 0; kq < 16 * NT; ++kq) s += Fs[kq + LDK * (NX + q)] * Fs[kq + LDK * j];
                if (isx) Xadd[e] = lam * s;
                else Radd[e - m8 * NX] = lam * s;
            }
        }
        MFP(10);
        __syncthreads();
        MFP(2);

        d4 acc3[NSL];                                                // this wave's tiles of Vxx_i
        // a tile of column 0, B operand from the LDS image: a chain of its own, two reads per product
        auto image_tile = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (sl.valid[u]) {                                       // (wave-uniform)
                acc3[u] = cxx_init(u);
                const double *apt = Fs + l4 + LDK * (16 * sl.ti[u] + l15);   // A[i][k] = F[k, 16ti+i]
                const double *bpt = W0 + l15 + 16 * l4;                // B[k][j] = W[k, j]
                chains<KS, 1, 0, false>([&](auto kc) { return bpt[64 * decltype(kc)::value]; },
                                        [&](auto, auto kc) { return apt[4 * decltype(kc)::value]; }, acc3 + u, [](auto) {});
            }
        };
        auto phaseC = [&]() __attribute__((always_inline)) {
        // ================= phase C: + ½(K'Y + Y'K) on the accumulators, tiles mirrored into Vxx (:69-72, :210) ==========
        {
            double kA[NSL][2], yA[NSL][2], kB[NSL][2], yB[NSL][2], *qp[NSL], *mp[NSL];
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const int gj = 16 * sl.tc[u] + l15, gi0 = 16 * sl.ti[u] + l4;
                qp[u] = Vs + gj + LDV * gi0;                          // Vxx[gi, gj] stored at (gj, gi): lanes contiguous
                mp[u] = Vs + gi0 + LDV * gj;                          // mirror position (gi, gj)
                const int ia = l4 + KSD * (16 * sl.ti[u] + l15), ib = l4 + KSD * gj;
                kA[u][0] = Ks[ia]; kA[u][1] = Ks[ia + 4]; yA[u][0] = Ys[ia]; yA[u][1] = Ys[ia + 4];
                kB[u][0] = Kh[ib]; kB[u][1] = Kh[ib + 4]; yB[u][0] = Yh[ib]; yB[u][1] = Yh[ib + 4];
            }
            // the four products of a tile depend on each other: run the tiles of the wave side by side (an unused slot multiplies
            // whatever its registers hold and stores nothing)
#pragma unroll
            for (int u = 0; u < NSL; ++u) acc3[u] = mf(kA[u][0], yB[u][0], acc3[u]);
#pragma unroll
            for (int u = 0; u < NSL; ++u) acc3[u] = mf(kA[u][1], yB[u][1], acc3[u]);
#pragma unroll
            for (int u = 0; u < NSL; ++u) acc3[u] = mf(yA[u][0], kB[u][0], acc3[u]);
#pragma unroll
            for (int u = 0; u < NSL; ++u) acc3[u] = mf(yA[u][1], kB[u][1], acc3[u]);
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                if (!sl.valid[u]) continue;
                // slot 0 is (c, c), slot 1 never diagonal, slot 3 is (0, 0) for NT = 4 and (1, 0) for NT = 3; slot 2: (0, 0) or (w, 0)
                const bool diag = u == 0 ? true : (u == 1 ? false : (u == 3 ? NT == 4 : sl.ti[u] == sl.tc[u]));
                // Off-diagonal tiles exist once and are mirrored.  A diagonal tile holds both (i,j) and (j,i), equal up to rounding:
                // its upper triangle is mirrored in the same way, so the result is exactly symmetric without an exchange
                // (the reference averages the two halves, (:71-72); the difference is of the order of the rounding error of G).
                const double av[4] = {acc3[u].x, acc3[u].y, acc3[u].z, acc3[u].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!diag || l4 + 4 * r <= l15) { qp[u][LDV * 4 * r] = av[r]; mp[u][4 * r] = av[r]; }
                }
            }
        }
        {                                                            // K_i[m, n]: 2 x 256 consecutive doubles (entries past m n repeat the last)
            double *kgi = Kg + (size_t)(nv * m) * i;
            const int e1 = tid + NTH < m * nv ? tid + NTH : m * nv - 1, e0 = tid < m * nv ? tid : m * nv - 1;
#if !(defined(MF2_EXP) && (MF2_EXP & 4))       // timing experiment 4: no K stores
            kgi[e0] = Ks[kls[0]]; kgi[e1] = Ks[kls[1]];
#endif
        }
        };
        if (wv == 0) {
            // ================= phase B, wave 0: reduce the partial tiles, gains (backward_pass.jl:30-68) =========
            __builtin_amdgcn_s_setprio(3);
            double x2[m8], xr[m8];
            {
                // row `lane` of G[:, u|Vx]: tile lane/16, register (lane%16)/4, tile row lane%4 -> offset 9*lane
                const double *pp = PT + 9 * lane;
                double s[m8 + 1];
#pragma unroll
                for (int q = 0; q <= m8; ++q) s[q] = ((pp[q] + pp[q + 20 * PTS]) + pp[q + 40 * PTS]) + pp[q + 60 * PTS];
#pragma unroll
                for (int q = 0; q < m8; ++q) {
                    x2[q] = s[q] + ((rowv && q < m) ? cxur[q] : 0.0);                // Qux[q, lane]  (:208)
                    xr[q] = (regType == 2) ? x2[q] + Xadd[q + m8 * lane] : x2[q];
                }
                Qs[lane] = s[m8] + (rowv ? gxc : 0.0);                            // Qx[lane]  (:203)
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {                                         // the u tile, rows l4 + 4r < 8
                const int aq = l4 + 4 * r;
                if (l15 <= m8) {
                    const double *pp = PT + (16 + r) * PTS + l4 * 9 + l15;
                    const double s = ((pp[0] + pp[20 * PTS]) + pp[40 * PTS]) + pp[60 * PTS];
                    if (l15 < m8) Quus[aq + m8 * l15] = s + ((aq < m && l15 < m) ? preq[r] : (aq == l15 ? 1.0 : 0.0));     // (:209); the padded controls: an identity block
                    else Qs[NX + aq] = s + (aq < m ? gu[r] : 0.0);                   // (:204)
                }
            }
            load_grad(inext);                                        // the gradients of the NEXT step, a whole step ahead of their use
            wave_sync();
            double H[m8 * m8], R[m8 * m8], kk[m8];
            unsigned clamped = 0u;
            int fail;
            double ri[m8];
            constexpr bool use_ri = !LIMS;                           // division-free factor on the unconstrained path; a kernel compiled for
                                                                     // limits takes the QP also for `lims[1,1] > lims[1,2]` (bounds at ±Inf)
            double qu[m8];
#pragma unroll
            for (int q = 0; q < m8; ++q) qu[q] = Qs[NX + q];
            if constexpr (use_ri) {
#pragma unroll
                for (int e = 0; e < m8 * m8; ++e) H[e] = Quus[e];
                if (regType == 2) {
#pragma unroll
                    for (int e = 0; e < m8 * m8; ++e) H[e] += Radd[e];
                } else {
#pragma unroll
                    for (int q = 0; q < m8; ++q) H[q + m8 * q] += lam;
                }
                fail = ddp_chol_rinv<m8>(H, R, ri);                  // cholesky(Hermitian(QuuF))  (:35)
#pragma unroll
                for (int q = 0; q < m8; ++q) kk[q] = qu[q];
                ddp_rsolve_neg<m8>(R, ri, kk);                       // k_i = -(R\Qu)  (:41)
            } else {
                // boxQP with one coordinate per lane (boxqp_rows.h): lane l15 < 8 of every 16-lane row holds row and column l15 of QuuF
                bqr::Rows<m8> qr;
                const bool qin = l15 < m8;
                const int qi = qcoord;
#pragma unroll
                for (int j = 0; j < m8; ++j) {
                    double hr = Quus[qi + m8 * j], hc = Quus[j + m8 * qi];
                    if (regType == 2) { hr += Radd[qi + m8 * j]; hc += Radd[j + m8 * qi]; }
                    else if (j == qi) { hr += lam; hc += lam; }
                    qr.Hrow[j] = qin ? hr : 0.0; qr.Hcol[j] = qin ? hc : 0.0;
                }
                const double uq = qu_next;                           // u[qi, i], requested a step ago
                { const double w_ = ug[(size_t)m * inext + (qreal ? qi : 0)]; qu_next = qreal ? w_ : 0.0; }
                const double gq = qin ? Qs[NX + qi] : 0.0, loq = qin ? qlo - uq : 0.0, upq = qin ? qhi - uq : 0.0;   // (:45-46)
                const double x0q = qin ? ks[qi] : 0.0;               // warm start k[:, min(i+1, N-1)] (:49)
                double xq;
                int iters;
                const int result = bqr::boxqp_rows<m8>(qr, gq, loq, upq, x0q, qpo, l15, xq, clamped, iters);
                fail = (result < 1);                                 // (:53)
                // every lane solves a column of K with the factor, and needs all of k
                asm volatile("s_nop 1" : "+v"(xq));
                sfor<0, m8>([&](auto qc) { constexpr int q = decltype(qc)::value; kk[q] = bqr::bcast<q>(xq); });
                sfor<0, m8>([&](auto cc) {
                    constexpr int c2 = decltype(cc)::value;
                    ri[c2] = qr.ri[c2];
                    sfor<0, m8>([&](auto kc) {
                        constexpr int k2 = decltype(kc)::value;
                        if constexpr (k2 < c2) R[k2 + m8 * c2] = bqr::bcast<c2>(qr.Rcol[k2]); else R[k2 + m8 * c2] = 0.0;     // R[k2][c2] lives in lane c2
                    });
                });
            }
            if (lane == 0) flag[0] = fail ? 1.0 : 0.0;
            // Quu[r, c] (r = lane & 7, c = lane >> 3) of the REAL controls goes out; the padded rows / columns do not exist in global memory
            double *quup = ((lane & 7) < m && (lane >> 3) < m) ? Quug + mm * i + (lane & 7) + m * (lane >> 3) : sink;
            if (fail) {
                *quup = Quus[lane];
            } else {
                // every LDS read first, every write last: the compiler cannot prove the K/Y writes do not alias Quus
                double col[m8];
#pragma unroll
                for (int q = 0; q < m8; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : xr[q];
                // Unconstrained regType 1: (Quu + λI)·K = -Qux and (Quu + λI)·k = -Qu hold to the backward error of the solve,
                // so Quu·K and Quu·k need no product with Quu (the other cases take it from LDS again: H is dead, R holds the factor)
                const bool by_residual = use_ri && regType != 2;
                if (!by_residual) {
#pragma unroll
                    for (int e = 0; e < m8 * m8; ++e) H[e] = Quus[e];
                }
                const double quu_l = Quus[lane];
                if (use_ri) ddp_rsolve_neg<m8>(R, ri, col);          // K_i column `lane`
                else {
                    chol_solve_ri<m8>(m8, R, ri, col);
#pragma unroll
                    for (int q = 0; q < m8; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : -col[q];
                }
                double y[m8], quuk[m8], kQu = 0.0, kQuuk = 0.0;
#pragma unroll
                for (int q = 0; q < m8; ++q) {                       // Y = Quu·K + 2·Qux;  Quu·k, dV (:64-68)
                    if (by_residual) {
                        y[q] = fma(-lam, col[q], x2[q]);                 // Quu·K = -Qux - λK
                        quuk[q] = -fma(lam, kk[q], qu[q]);               // Quu·k = -Qu - λk
                    } else {
                        double t = 2.0 * x2[q], t2 = 0.0;
#pragma unroll
                        for (int q2 = 0; q2 < m8; ++q2) {
                            const double hq = H[(q < q2 ? q : q2) + m8 * (q < q2 ? q2 : q)];  // upper triangle, like the factorisation
                            t += hq * col[q2]; t2 += hq * kk[q2];
                        }
                        y[q] = t; quuk[q] = t2;
                    }
                }
#pragma unroll
                for (int q = 0; q < m8; ++q) { kQu += kk[q] * qu[q]; kQuuk += kk[q] * quuk[q]; }
                dV0 += kQu; dV1 += 0.5 * kQuuk;                      // (every lane; lane 0 reports)
#pragma unroll
                for (int q = 0; q < m8; ++q) {
                    Xs[q + m8 * lane] = x2[q];
                    Ks[q + KSD * lane] = col[q];                      // (:76) K_i goes out from this image in phase C
                    Ys[q + KSD * lane] = y[q];
                    Kh[q + KSD * lane] = 0.5 * col[q];                // the B operands of the rank-16 update ½(K'Y + Y'K): halved here, once, instead
                    Yh[q + KSD * lane] = 0.5 * y[q];                  // of by every lane of every tile in phase C (vector instructions between products)
                }
                *quup = quu_l;
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < m8; ++q) { Quuks[q] = quuk[q]; ks[q] = kk[q]; }
                }
                wave_sync();
                { const int qc = lane < m ? lane : m - 1; kg[(size_t)m * i + qc] = ks[qc]; }       // k_i (lanes past m repeat the last entry)
                {                                                    // Vx_i (:69)
                    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                    for (int q = 0; q < m8; ++q) { s1 += col[q] * quuk[q]; s2 += col[q] * qu[q]; s3 += x2[q] * kk[q]; }
                    const double v = ((Qs[lane] + s1) + s2) + s3;
                    vs[lane] = v;
                    *(double *)((char *)(Vxg + (size_t)nv * i) + row8) = vs[rowc];      // (lanes past n repeat the last entry)
                }
            }
            __builtin_amdgcn_s_setprio(0);
            MFP(3);
            if constexpr (NT == 4 && !LIMS) image_tile(ic<2>{});     // (0, 0)
            MFP(4);
            // The gain wave has its OWN copy of the second barrier and of phase C: the Jacobian registers of the other waves (requested in
            // phase A, written to the LDS in phase C) are then dead on every path through this wave's code — its 8 x 8 solve is where the
            // kernel runs out of registers.
            MFP(5);
            __syncthreads();
            MFP(6);
            if (flag[0] != 0.0) { diverge = i + 1; }                 // block-uniform
            else {
                phaseC();
                MFP(11);
                MFP(7);
                __syncthreads();
                MFP(8);
            }
        } else {
            // ================= phase B, wave c = 1..3: column tile c of W = Vxx·F in the registers, then its tiles cxx + fx'W ==========
            if (wv < NT) {
                d4 Wc[NT];
#pragma unroll
                for (int rb = 0; rb < NT; ++rb) Wc[rb] = d4{0.0, 0.0, 0.0, 0.0};
                const double *ap = Vs + l15 + LDV * l4;                   // A[i][k] = Vxx[16rb+i, k]
                const double *bp = Fs + l4 + LDK * (16 * wv + l15);       // B[k][j] = F[k, 16c+j]
                chains<KS, NT, 0, false>([&](auto kc) { return bp[4 * decltype(kc)::value]; },
                                         [&](auto rc, auto kc) { return ap[16 * decltype(rc)::value + LDV * 4 * decltype(kc)::value]; }, Wc, sideB);
                static_assert(KS * NT >= NFL, "the Jacobian loads fit behind the products of a W column");
                MFP(3);
                // the two own tiles (ti, c) from the registers: k = 16 rb + l4 + 4 r is accumulator register r of W[rb][c] in this lane
                const double *fpt[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc3[u] = cxx_init(u);
                    fpt[u] = Fs + l4 + LDK * (16 * sl.ti[u] + l15);     // A[i][k] = F[k, 16ti+i]
                }
                chains<KS, 2, 0, false>([&](auto kc) { constexpr int k_ = decltype(kc)::value; const d4 w_ = Wc[k_ >> 2];
                                                      return (k_ & 3) == 0 ? w_.x : ((k_ & 3) == 1 ? w_.y : ((k_ & 3) == 2 ? w_.z : w_.w)); },
                                        [&](auto uc, auto kc) { constexpr int k_ = decltype(kc)::value; return fpt[decltype(uc)::value][16 * (k_ >> 2) + 4 * (k_ & 3)]; },
                                        acc3, sideB2);
                sfor<KS * NT + 2 * KS, NSIDE>([&](auto sc) { sideB(sc); });      // (whatever did not fit behind a product)
            } else {
                sfor<0, NSIDE>([&](auto sc) { sideB(sc); });         // NT = 3: wave 3 has no column, only its share of the traffic
            }
            MFP(4);
            image_tile(ic<2>{});
            if constexpr (has_slot3<NT, LIMS>()) image_tile(ic<3>{});
            MFP(5);
            __syncthreads();
            MFP(6);
            if (flag[0] != 0.0) { diverge = i + 1; }                 // block-uniform
            else {
                phaseC();
                MFP(11);
                f_store(nv, fwv);                                    // everybody is past its reads of F (second barrier)
                MFP(7);
                __syncthreads();
                MFP(8);
            }
        }
        if (diverge) break;
    }
    MFP_PRINT;
    if (diverge) {   // outputs earlier in time than the failing step are zero (backward_pass.jl:226-229)
        const size_t ie = (size_t)diverge;
        for (size_t e = tid; e < nm * ie; e += NTH) Kg[e] = 0.0;
        for (size_t e = tid; e < (size_t)m * ie; e += NTH) kg[e] = 0.0;
        for (size_t e = tid; e < (size_t)n * ie; e += NTH) Vxg[e] = 0.0;
        for (size_t e = tid; e < nn * ie; e += NTH) Vxxg[e] = 0.0;
        for (size_t e = tid; e < mm * (ie - 1); e += NTH) Quug[e] = 0.0;
    } else {
        for (int e = tid; e < n * n; e += NTH) { const int r = e % n, c = e / n; Vxxg[e] = Vs[r + LDV * c]; }
    }
    if (tid == 0) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
}

template <int NT, bool LIMS, bool CTV, bool PAIR>
static int launch_k(ddp_handle h, const BPM2Args &a)
{
    const size_t shmem = (size_t)oTot * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        DDP_HIP(hipFuncSetAttribute((const void *)back_pass_mf2_kernel<NT, LIMS, CTV, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((back_pass_mf2_kernel<NT, LIMS, CTV, PAIR>), dim3(a.B), dim3(NTH), shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}

// time-invariant cost: its terms are loaded once, before the loop; even n: 16-byte pieces of the Jacobian and of Vxx
template <int NT, bool LIMS>
static int launch(ddp_handle h, const BPM2Args &a)
{
    if (a.n & 1) return a.cost_tv ? launch_k<NT, LIMS, true, false>(h, a) : launch_k<NT, LIMS, false, false>(h, a);
    return a.cost_tv ? launch_k<NT, LIMS, true, true>(h, a) : launch_k<NT, LIMS, false, true>(h, a);
}

}   // namespace mf2

int ddp_bpm2_launch_lims(ddp_handle h, const BPM2Args &a, int nt);      // back_pass_mf2_lims.hip
