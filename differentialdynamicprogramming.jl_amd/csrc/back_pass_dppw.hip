// back_pass_dppw.hip — the 16-lane-row backward pass of back_pass_dpp.hip (one DPP row per trajectory, four per wave; same arithmetic:
// src/backward_pass.jl:217-252 + :28-42,:64-76) for MACHINE-FILLING batches of the time-invariant, unlimited case, with the result
// traffic taken off the chain so that TWO chain waves fit a SIMD:
//
//   chain wave   the recursion of its four trajectories; the results of a step go into an LDS record per trajectory
//                (Vxx 100 | Vx 10 | K 20 | k 2 | Quu 4 doubles, the order they have in memory) — no global store, no address arithmetic
//                for results; ½(V + V') through the record's own Vxx area (column written, row read, symmetric column written back:
//                LDS operations of a wave execute in order), so what the writer finds is exactly symmetric.
//   writer wave  (one per two chain waves) copies finished groups of 4 steps x 4 trajectories LDS -> registers, releases the buffer at
//                once, then stores the 16-byte pieces, contiguous across its lanes (20 store instructions per group, scalar array base +
//                a 32-bit lane offset: no per-item address arithmetic).
//
// Why (profiles, B = 32 768): back_pass_dpp takes 12.8 ms per pass; without its stores 8.9 ms, and its ~30 8-byte store instructions per
// wave-step (~60 issue cycles each) cannot be hidden by a second wave.  A chain wave WITHOUT them is bound by the issue rate of a lone
// wave (514 instructions per step, one every ~7 cycles: the first version of this kernel — 4 chain + 4 writer waves, one chain per SIMD
// — took the same 12.7 ms), so the lever is a second chain wave per SIMD: 8 chain + 4 writer waves per work-group, three waves per SIMD
// (<= 168 registers), ONE record buffer of 4 steps per chain wave (the writer frees it as soon as its LDS reads have returned, a whole
// chain step before the chain needs it again) = 139 KB of LDS, one work-group per CU: 10.7 ms.  What the rest came from:
//   - groups aligned to multiples of 4 of the ABSOLUTE step index, so that the bursts of Vxx (3 200 B), K (640 B), Quu (128 B) are whole
//     128-byte lines (a line completed by the NEXT group, ~13 us later, has left the L2 half-written by then): 10.0 ms;
//   - Vx (320 B per group) and k (64 B) of the upper group of an aligned pair wait in the writer's registers for the lower one,
//     non-temporal stores: 9.8 ms;
//   - the chain's accumulators start from the cost Hessians / from Qxx (no clears, no final adds), broadcasts without an "old" operand:
//     421 -> 383 fp64 vector instructions per step, 9.45 ms (= 0.51 of 8 TB/s algorithmic).
// Removal experiments (DDP_DPPW_EXP, profiles/ab_dppw_exp.sh): no result stores 7.67 ms (the vector-issue floor of two chain waves per
// SIMD: 2.05 ns per fp64 instruction and SIMD, profiles/microbench/dpp_fma_bench2.hip), only Vxx stored 8.75 ms; stores aimed at an
// L2-resident region cost nothing, i.e. the remaining 1.8 ms is HBM write back-pressure, not instruction issue; staggering the chain
// waves in time changes nothing.
#include <stdlib.h>
#include <type_traits>
#include "ddp_internal.h"

namespace {

struct BPWArgs {
    int N, B, regType;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
    double *sink;               // >= 64 x 16 B of device memory for the writer lanes without an item
};

typedef double d2 __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ void fmac_bc(double &acc, double src0, double src1)
{   // acc += src0[lane L of this 16-lane row] * src1
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src0), "v"(src1), "n"(L));
}
template <int L>
__device__ __forceinline__ double row_bcast(double x)
{   // every lane of a row has a source lane: no "old" value to prepare (the builtin costs a v_mov of it per use)
    double r;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(x), "n"(L));
    return r;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// freshly written DPP sources: the hazard (VALU write -> DPP read, 2 wait states) is not tracked into inline asm
template <int NN>
__device__ __forceinline__ void dpp_fence(double (&v)[NN])
{
#pragma unroll
    for (int i = 0; i < NN; ++i) asm volatile("" : "+v"(v[i]));
    asm volatile("s_nop 1" ::: "memory");
}

constexpr int NCW = 8;                              // chain waves per work-group
constexpr int NWW = 4;                              // writer waves per work-group (waves NCW ..): writer w serves chain waves 2w, 2w + 1
constexpr int CPW = NCW / NWW;                      // chain waves per writer
constexpr int GW = 4;                               // steps per group = per record buffer (one buffer per chain wave)
constexpr int FIN = 1 << 30, DONE_ALL = 1 << 29;

typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int lds_load_flag(const int *p) { return *(const volatile lds_int *)p; }
__device__ __forceinline__ void lds_store_flag(int *p, int v) { asm volatile("" ::: "memory"); *(volatile lds_int *)p = v; asm volatile("" ::: "memory"); }

// EXP (removal experiments, profiles/ab_dppw_exp.sh): 1 = the writer reads the records but stores nothing, 2 = only Vxx leaves
template <int NS, int MS, int EXP>
__global__ __launch_bounds__(DDP_WAVE * (NCW + NWW)) void back_pass_dppw_kernel(BPWArgs a)
{
    constexpr int n = NS, m = MS, p = n + m, G = 16, GPW = DDP_WAVE / G;
    constexpr int nn = n * n, nm = n * m, mm = m * m;
    constexpr int REC = nn + n + nm + m + mm, R_VX = nn, R_K = nn + n, R_KV = R_K + nm, R_QUU = R_KV + m;     // one step of one trajectory
    constexpr int SLOT = GPW * REC, BUF = GW * SLOT;            // doubles per step / per chain wave
    static_assert(p + 1 <= G && n % 2 == 0 && m % 2 == 0, "16-byte pieces of every result array, n + m + 1 lanes per row");
    static_assert(sizeof(double) * NCW * BUF <= 150 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) double recs[NCW][GW][GPW][REC];
    __shared__ int flags[NCW][4];                                // per chain wave: [0] groups finished (| FIN), [1] groups read by the writer
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / DDP_WAVE), lane = threadIdx.x % DDP_WAVE;     // wave: an SGPR, so that what follows from it is scalar
    const bool writer = wave >= NCW;
    if (threadIdx.x < NCW * 4) flags[threadIdx.x / 4][threadIdx.x % 4] = 0;
    __syncthreads();
    const int N = a.N;

    if (writer) {
        // ================================================ the writer ====================================================
        if (N < 2) return;
        const int c0 = (wave - NCW) * CPW;                       // its chain waves: c0, c0 + 1
        // item = one 16-byte piece; per array the order (trajectory, step ascending, piece): consecutive lanes -> consecutive global
        // addresses.  A pass (one instruction over 64 items) never mixes arrays, so its global address is a scalar base + a lane offset;
        // the lanes past the end of an array repeat its last item (same bytes to the same address).
        constexpr int PC[5] = {nn / 2, n / 2, nm / 2, m / 2, mm / 2};             // pieces per step: 50 5 10 1 2
        constexpr int PER[5] = {nn, n, nm, m, mm};                                // doubles per step
        constexpr int ROFF[5] = {0, R_VX, R_K, R_KV, R_QUU};
        constexpr int NPA[5] = {(GPW * GW * PC[0] + 63) / 64, (GPW * GW * PC[1] + 63) / 64, (GPW * GW * PC[2] + 63) / 64,
                                (GPW * GW * PC[3] + 63) / 64, (GPW * GW * PC[4] + 63) / 64};
        constexpr int Q0[6] = {0, NPA[0], NPA[0] + NPA[1], NPA[0] + NPA[1] + NPA[2], NPA[0] + NPA[1] + NPA[2] + NPA[3],
                               NPA[0] + NPA[1] + NPA[2] + NPA[3] + NPA[4]};
        constexpr int NP = Q0[5];                                                 // 13 + 2 + 3 + 1 + 1 = 20 passes per group
        unsigned loff[NP], voff[NP];                             // byte offset inside a chain wave's buffer; byte offset from the array's group base
        static_for<0, 5>([&](auto ac) {
            constexpr int A = decltype(ac)::value;
            static_for<0, NPA[A]>([&](auto qc) {
                constexpr int qa = decltype(qc)::value, q = Q0[A] + qa;
                constexpr int items = GPW * GW * PC[A];
                const int idx = qa * DDP_WAVE + lane, id = idx < items ? idx : items - 1;
                const int t = id / (GW * PC[A]), r = id % (GW * PC[A]), so = r / PC[A], pc = r % PC[A];
                const int slot = GW - 1 - so;                    // the chain walks a group downwards in time: slot 0 = highest step
                loff[q] = 8u * (unsigned)(c0 * BUF + (slot * GPW + t) * REC + ROFF[A] + 2 * pc);
                voff[q] = 8u * (unsigned)(((size_t)t * N + so) * PER[A] + 2 * pc);
            });
        });
        // groups are aligned to multiples of GW in the ABSOLUTE step index: with 800 / 160 / 32 bytes per step the bursts of Vxx, K, Quu are
        // then whole 128-byte lines (N a multiple of 4).  Group 0 holds the (N-2) % GW + 1 highest steps in its LAST slots.
        const int ilo0 = (N - 2) / GW * GW, top0 = N - 2 - ilo0;  // lowest step of group 0, its highest step offset
        double *const gb[5] = {a.Vxx, a.Vx, a.K, a.k, a.Quu};
        int g[CPW];
        bool fin[CPW];
        int onm[CPW];                                            // bit t: trajectory t of the chain wave exists and is active
        // Vx (80 bytes per step) and k (16) fill whole lines only over EIGHT steps: the upper group of an aligned pair waits in registers
        // for the lower one, both leave back to back and meet in the L2 (a line completed 13 us later has been evicted half-written)
        constexpr int NH = NPA[1] + NPA[3];
        d2 hv[CPW][NH];
        bool held[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const long tb0 = ((long)blockIdx.x * NCW + c0 + c) * GPW;
            g[c] = 0;
            held[c] = false;
            fin[c] = tb0 >= a.B;
            onm[c] = 0;
            for (int t = 0; t < GPW; ++t) {
                const long tb = tb0 + t;
                if (tb < a.B && !(a.active && a.active[tb] == 0)) onm[c] |= 1 << t;
            }
            onm[c] = __builtin_amdgcn_readfirstlane(onm[c]);
        }
        for (;;) {
            bool all = true, progress = false;
            static_for<0, CPW>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (fin[c]) return;
                all = false;
                int *ready = &flags[c0 + c][0], *done = &flags[c0 + c][1];
                const int r = __builtin_amdgcn_readfirstlane(lds_load_flag(ready));
                const int ng = r & (FIN - 1);
                if (g[c] >= ng) {
                    if (r & FIN) {
                        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this chain wave's results have left
                        lds_store_flag(done, DONE_ALL);
                        fin[c] = true;
                    }
                    return;
                }
                progress = true;
                asm volatile("" ::: "memory");
                const char *rb = (const char *)&recs[0][0][0][0] + c * (BUF * 8);     // loff holds the offset of chain wave c0
                d2 v[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) v[q] = *(const d2 *)(rb + loff[q]);
                const int ilo = ilo0 - g[c] * GW;                // lowest step of the group
                const int top = g[c] == 0 ? top0 : GW - 1;
                const long tb0 = ((long)blockIdx.x * NCW + c0 + c) * GPW;
                ++g[c];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lds_store_flag(done, g[c]);                      // the buffer is free again; the stores follow
                const bool plain = onm[c] == (1 << GPW) - 1 && top == GW - 1;
                auto stores = [&](auto plain_c) {
                    static_for<0, 5>([&](auto ac) {
                        constexpr int A = decltype(ac)::value;
                        const char *sb = (const char *)(gb[A] + (size_t)PER[A] * ((size_t)N * (size_t)tb0)) + (long)PER[A] * 8 * (long)ilo;
                        static_for<0, NPA[A]>([&](auto qc) {
                            constexpr int qa = decltype(qc)::value, q = Q0[A] + qa;
                            if constexpr (EXP == 1 || (EXP == 2 && A >= 1)) return;
                            const unsigned vo = voff[q];         // (asm operands do not capture)
                            const d2 vv = v[q];
                            const char *sbq = sb;
                            if constexpr (decltype(plain_c)::value && (A == 1 || A == 3)) {
                                constexpr int hq = A == 1 ? qa : NPA[1] + qa;
                                if (ilo & GW) { hv[c][hq] = vv; if (A == 3) held[c] = true; return; }      // upper half of an aligned pair of groups
                                if (held[c]) {
                                    const d2 hh = hv[c][hq];
                                    const char *sbh = sbq + GW * PER[A] * 8;
                                    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(vo), "v"(hh), "s"(sbh) : "memory");
                                    if (A == 3) held[c] = false;
                                }
                            }
                            if constexpr (decltype(plain_c)::value) {
                                // nt: the results are not read again by this kernel (9.80 against 10.0 ms per pass at B = 32 768)
                                asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(vo), "v"(vv), "s"(sbq) : "memory");
                            } else {                              // a missing / finished trajectory, or the short first group
                                constexpr int items = GPW * GW * PC[A];
                                const int idx = qa * DDP_WAVE + lane, id = idx < items ? idx : items - 1;
                                const int t = id / (GW * PC[A]), so = (id % (GW * PC[A])) / PC[A];
                                if (((onm[c] >> t) & 1) && so <= top)
                                    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vo), "v"(vv), "s"(sbq) : "memory");
                            }
                        });
                    });
                };
                if (plain) stores(std::true_type{});
                else stores(std::false_type{});
            });
            if (all) break;
            if (!progress) __builtin_amdgcn_s_sleep(4);
        }
        return;
    }

    const int cw = wave;
    const long tb0 = ((long)blockIdx.x * NCW + cw) * GPW;       // first trajectory of this chain wave
    if (tb0 >= a.B) return;
    int *ready = &flags[cw][0], *done = &flags[cw][1];

    // ==================================================== the chain =====================================================
    __builtin_amdgcn_s_setprio(3);
    const int grp = lane / G, j = lane % G;
    long tb = tb0 + grp;
    const bool valid = tb < a.B;
    if (!valid) tb = a.B - 1;                                   // all lanes stay alive (DPP reads every lane)
    const size_t b = (size_t)tb;
    const bool act = valid && !(a.active && a.active[b] == 0);
    const bool inx = j < n, inu = j >= n && j < p, ink = j == p;   // column roles: x-columns, u-columns, spare lane (k)
    const int jx = inx ? j : 0, ja = inu ? j - n : 0;
    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    double *Kg = a.K + (size_t)nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + (size_t)mm * N * b, *Vxg = a.Vx + (size_t)n * N * b,
           *Vxxg = a.Vxx + (size_t)nn * N * b;
    const double lam = a.lambda[b];
    const bool reg2 = a.regType == 2;
    // column j of F = [fx fu] and of the cost Hessians (zero for the idle lanes)
    double Fcol[n], cxxcol[n], ccol[m];
    {
        const double zF = (j < p) ? 1.0 : 0.0, zx = inx ? 1.0 : 0.0, zu = (inx || inu) ? 1.0 : 0.0;
        const double *src = (j < n) ? a.fx + (size_t)n * jx : a.fu + (size_t)n * ja;
#pragma unroll
        for (int r = 0; r < n; ++r) { Fcol[r] = zF * src[r]; cxxcol[r] = zx * a.cxx[(size_t)n * jx + r]; }
        const double *s2 = inx ? a.cxu + jx : a.cuu + (size_t)m * ja;                                  // x-lanes cxu[j, :], u-lanes cuu[:, j-n]
#pragma unroll
        for (int q = 0; q < m; ++q) ccol[q] = zu * s2[inx ? (size_t)n * q : (size_t)q];
    }
    double Vcol[n], vj;
    {   // terminal step (backward_pass.jl:234-236): straight to global memory
        const size_t tl = (size_t)(N - 1);
#pragma unroll
        for (int r = 0; r < n; ++r) Vcol[r] = inx ? a.cxx[(size_t)n * jx + r] : 0.0;
        vj = inx ? cx[(size_t)n * tl + jx] : 0.0;
        if (act) {
            if (inx) {
#pragma unroll
                for (int r = 0; r < n; ++r) Vxxg[nn * tl + (size_t)n * j + r] = Vcol[r];
                Vxg[(size_t)n * tl + j] = vj;
#pragma unroll
                for (int q = 0; q < m; ++q) Kg[nm * tl + (size_t)m * j + q] = 0.0;
            }
            if (inu) {
#pragma unroll
                for (int q = 0; q < m; ++q) Quug[mm * tl + (size_t)m * ja + q] = a.cuu[q + (size_t)m * ja];
            }
            if (ink) {
#pragma unroll
                for (int q = 0; q < m; ++q) kg[(size_t)m * tl + q] = 0.0;
            }
        }
    }
    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    if (N >= 2 && !__any(act)) {                                // every trajectory of this wave has finished: the writer is released at once
        lds_store_flag(ready, FIN);
        return;
    }
    if (N >= 2) {
        double FuF[m];                                          // regType 2 adds λ·F_u'F to the u-rows (backward_pass.jl:245-247)
#pragma unroll
        for (int q = 0; q < m; ++q) FuF[q] = 0.0;
        dpp_fence(Fcol);
        if (reg2) {
            static_for<0, n>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                static_for<0, m>([&](auto qc) { constexpr int q = decltype(qc)::value; fmac_bc<n + q>(FuF[q], Fcol[k], Fcol[k]); });
            });
        }
        // this lane's record cells: column j of Vxx (x-lanes), K[:, j] and Vx[j] (x-lanes), Quu[:, j-n] (u-lanes), k (spare lane)
        double *rw = &recs[cw][0][grp][0];
        auto rc_load = [&](int i) { return inx ? cx[(size_t)n * i + jx] : cu[(size_t)m * i + ja]; };     // cx[j, i] | cu[j-n, i]
        // the gradient entry of step i is requested four steps ahead: with the result stores of a machine-filling batch in the memory
        // pipeline a load takes several microseconds
        auto rc_at = [&](int i) { return rc_load(i > 0 ? i : 0); };
        double rc = rc_load(N - 2), rc1 = rc_at(N - 3), rc2 = rc_at(N - 4), rc3 = rc_at(N - 5);
        dpp_fence(Vcol);
        asm volatile("s_nop 1" : "+v"(vj));
        int g = 0, slot = GW - 1 - (N - 2) % GW;                // groups end on multiples of GW of the step index (the writer's whole lines)
        for (int i = N - 2; i >= 0; --i) {
            const double rc4 = EXP == 3 ? 0.001 : rc_at(i - 4);
            double *rec = rw + slot * SLOT;
            // ================= P1: w = Vxx·F[:,j],  q = c + F[:,j]'Vx ==================================
            double w[n], qj = 0.0;
#pragma unroll
            for (int r = 0; r < n; ++r) w[r] = 0.0;
            static_for<0, n>([&](auto lc) {
                constexpr int l = decltype(lc)::value;
                static_for<0, n>([&](auto rcx) { constexpr int r = decltype(rcx)::value; fmac_bc<l>(w[r], Vcol[r], Fcol[l]); });
                fmac_bc<l>(qj, vj, Fcol[l]);
            });
            qj += rc;                                                    // Qx (x-lanes) / Qu (u-lanes)  (:240-241)
            // ================= P2: g = F'·w  (column j of G = F'VxxF) ===================================
            double gg[p];
#pragma unroll
            for (int r = 0; r < p; ++r) gg[r] = r < n ? cxxcol[r] : ccol[r - n];     // the accumulators start from the cost Hessian: Qxx[:, j] (:244), Qux / Quu (:242-243)
            dpp_fence(w);
            static_for<0, n>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                static_for<0, p>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(gg[ii], Fcol[k], w[k]); });
            });
            double gu[m], gr[m];                                         // x-lanes: Qux[:, j]; u-lanes: Quu[:, j-n]
#pragma unroll
            for (int q = 0; q < m; ++q) {
                gu[q] = gg[n + q];
                gr[q] = gu[q] + (reg2 ? lam * FuF[q] : ((j == n + q) ? lam : 0.0));     // Qux_reg / QuuF (:246-247)
            }
            // ================= P3: gains (:30-42): every lane of the row factorises QuuF ====================
            double Quu[m * m], H[m * m], R[m * m], Qu[m], kk[m], Kc[m], ri[m];
            dpp_fence(gu);
            dpp_fence(gr);
            asm volatile("" : "+v"(qj));                                 // (written before the fences' s_nop)
            static_for<0, m>([&](auto bc) {
                constexpr int bb = decltype(bc)::value;
                Qu[bb] = row_bcast<n + bb>(qj);
                static_for<0, m>([&](auto ac) {
                    constexpr int aa = decltype(ac)::value;
                    Quu[aa + m * bb] = row_bcast<n + bb>(gu[aa]);
                    H[aa + m * bb] = row_bcast<n + bb>(gr[aa]);
                });
            });
            int fail = 0;
#pragma unroll
            for (int c = 0; c < m; ++c) {                                // cholesky(Hermitian(QuuF)) (:35), reciprocal pivots
                double ajj = H[c + m * c];
#pragma unroll
                for (int k2 = 0; k2 < c; ++k2) ajj -= R[k2 + m * c] * R[k2 + m * c];
                if (!(ajj > 0.0) && fail == 0) fail = c + 1;
                ri[c] = ddp_rsqrt(ajj);
#pragma unroll
                for (int c2 = c + 1; c2 < m; ++c2) {
                    double s = H[c + m * c2];
#pragma unroll
                    for (int k2 = 0; k2 < c; ++k2) s -= R[k2 + m * c] * R[k2 + m * c2];
                    R[c + m * c2] = s * ri[c];
                }
            }
            auto rsolve = [&](double (&bv)[m]) {                         // bv <- -(R'R)\bv
#pragma unroll
                for (int c = 0; c < m; ++c) {
                    double s = bv[c];
#pragma unroll
                    for (int k2 = 0; k2 < c; ++k2) s -= R[k2 + m * c] * bv[k2];
                    bv[c] = s * ri[c];
                }
#pragma unroll
                for (int c = m - 1; c >= 0; --c) {
                    double s = bv[c];
#pragma unroll
                    for (int k2 = c + 1; k2 < m; ++k2) s -= R[c + m * k2] * bv[k2];
                    bv[c] = s * ri[c];
                }
#pragma unroll
                for (int c = 0; c < m; ++c) bv[c] = -bv[c];
            };
#pragma unroll
            for (int q = 0; q < m; ++q) { kk[q] = Qu[q]; Kc[q] = gr[q]; }
            rsolve(kk);                                                  // k_i = -(R\Qu)        (:41)
            rsolve(Kc);                                                  // K_i[:, j] = -(R\Qux_reg[:, j])  (:42)
            const bool alive = diverge == 0 && !fail;
            if (diverge == 0 && fail) diverge = i + 1;                   // (:37-38)
            double Y[m], Quuk[m];
#pragma unroll
            for (int q = 0; q < m; ++q) {
                double t = gu[q], s = 0.0;                               // T = Quu·K + Qux, Y = T + Qux
#pragma unroll
                for (int q2 = 0; q2 < m; ++q2) { t += Quu[q + m * q2] * Kc[q2]; s += Quu[q + m * q2] * kk[q2]; }
                Y[q] = t + gu[q];
                Quuk[q] = s;                                             // (:64)
            }
            if (alive) {                                                 // (:68)
#pragma unroll
                for (int q = 0; q < m; ++q) { dV0 += kk[q] * Qu[q]; dV1 += 0.5 * kk[q] * Quuk[q]; }
            }
            // ================= P4: value update (:69-72) ===================================================
            // column j of Qxx + ½(K'Y + Y'K), accumulated ON Qxx[:, j] with Y halved first (exact): no accumulators to clear, nothing to add
#pragma unroll
            for (int q = 0; q < m; ++q) Y[q] *= 0.5;
            dpp_fence(Kc);
            dpp_fence(Y);
            static_for<0, m>([&](auto ac) {
                constexpr int aa = decltype(ac)::value;
                static_for<0, n>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(gg[ii], Kc[aa], Y[aa]); });
                static_for<0, n>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(gg[ii], Y[aa], Kc[aa]); });
            });
            double vx = qj;                                              // Vx_i[j] (:69)
#pragma unroll
            for (int q = 0; q < m; ++q) vx += Kc[q] * (Quuk[q] + Qu[q]) + gu[q] * kk[q];
            // ---- the step's record (a diverged trajectory keeps writing; its range is zero-filled after the loop)
            double vnew[n];
            if (slot == 0 && g >= 1) {                                       // the buffer holds group g - 1 until the writer has read it
                while (lds_load_flag(done) < g) __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int r = 0; r < n; ++r) vnew[r] = gg[r];                 // Qxx + ½(S+S'), column j as computed (lanes >= n: never read by anyone)
            if (inx) {
#pragma unroll
                for (int r = 0; r < n; ++r) rec[n * j + r] = vnew[r];
#pragma unroll
                for (int q = 0; q < m; ++q) rec[R_K + m * j + q] = Kc[q];                                      // (:76)
                rec[R_VX + j] = vx;
            }
            if (inu) {
#pragma unroll
                for (int q = 0; q < m; ++q) rec[R_QUU + m * ja + q] = gu[q];
            }
            if (ink) {
#pragma unroll
                for (int q = 0; q < m; ++q) rec[R_KV + q] = kk[q];                                             // (:75)
            }
            wave_sync();
            // Vxx_i = ½(V + V') (:71-72): row j of the columns just written; the recursion continues with the symmetrised value
#pragma unroll
            for (int r = 0; r < n; ++r) Vcol[r] = 0.5 * (vnew[r] + rec[n * r + jx]);
            wave_sync();
            if (inx) {
#pragma unroll
                for (int r = 0; r < n; ++r) rec[n * j + r] = Vcol[r];
            }
            vj = vx;
            rc = rc1; rc1 = rc2; rc2 = rc3; rc3 = rc4;
            dpp_fence(Vcol);
            asm volatile("s_nop 1" : "+v"(vj));
            if (++slot == GW) {                                          // the group is complete (i is a multiple of GW)
                slot = 0;
                ++g;
                wave_sync();
                lds_store_flag(ready, g);
            }
        }
        lds_store_flag(ready, g | FIN);
        if (diverge) {                                                   // outputs earlier in time than a failing step are zero (:37-38 with :226-229)
            while (lds_load_flag(done) != DONE_ALL) __builtin_amdgcn_s_sleep(4);     // the writer's copies of the garbage steps have left
            __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
            if (act) {
                const size_t ie = (size_t)diverge;                       // = failing 0-based step + 1
                for (size_t e = j; e < nm * ie; e += G) Kg[e] = 0.0;
                for (size_t e = j; e < (size_t)m * ie; e += G) kg[e] = 0.0;
                for (size_t e = j; e < (size_t)n * ie; e += G) Vxg[e] = 0.0;
                for (size_t e = j; e < nn * ie; e += G) Vxxg[e] = 0.0;
                for (size_t e = j; e < mm * (ie - 1); e += G) Quug[e] = 0.0;
            }
        }
    }
    if (act && j == 0) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
}

}   // namespace

// returns 1 if this launch is not for this kernel (the caller goes on to back_pass_dpp), 0 launched, <0 error
int ddp_launch_back_pass_dppw(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                              const double *cxx, const double *cxu, const double *cuu, const double *fx,
                              const double *fu, const double *lambda, const int32_t *active, double *K,
                              double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge)
{
    if (d->N > 400000) return 1;                                 // 32-bit lane offsets inside a chain wave's four trajectories
    if (d->n != 10 || d->m != 2 || d->has_lims || d->fx_tv || d->cost_tv || d->fx_batched || d->cost_batched || !h->sink) return 1;
    const char *env = ddp_env(h, ENV_DPPW);                        // 0: never, 1: whenever the shape allows (A/B timing, tests)
    if (env && env[0] == '0') return 1;
    if (!(env && env[0] == '1') && d->B < 6144) return 1;        // measured cross-over (profiles/ab_fill_crossover.sh): 4 096: dpp 1.81, this 1.97 ms; 6 144: mx 2.84, this 2.21
    if ((((uintptr_t)K | (uintptr_t)k | (uintptr_t)Quu | (uintptr_t)Vx | (uintptr_t)Vxx) & 15) != 0) return 1;     // 16-byte pieces
    BPWArgs a;
    a.N = d->N; a.B = d->B; a.regType = d->regType;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge; a.sink = (double *)h->sink;
    const dim3 grid((unsigned)((d->B + NCW * 4 - 1) / (NCW * 4))), block(DDP_WAVE * (NCW + NWW));
    const char *exp_env = ddp_env(h, ENV_DPPW_EXP);
    const int exp = exp_env ? atoi(exp_env) : 0;
    if (exp == 1) hipLaunchKernelGGL((back_pass_dppw_kernel<10, 2, 1>), grid, block, 0, h->stream, a);
    else if (exp == 2) hipLaunchKernelGGL((back_pass_dppw_kernel<10, 2, 2>), grid, block, 0, h->stream, a);
    else if (exp == 3) hipLaunchKernelGGL((back_pass_dppw_kernel<10, 2, 3>), grid, block, 0, h->stream, a);
    else hipLaunchKernelGGL((back_pass_dppw_kernel<10, 2, 0>), grid, block, 0, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}
