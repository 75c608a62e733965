// back_pass_mxg.hip — the fp64-matrix-core TILE backward pass (back_pass_mx.hip) for a RANGE of shapes: any n <= 12, m <= 4 with
// n + m <= 15, no control limits; one wavefront per trajectory, every matrix of a time step in ONE 16x16 v_mfma_f64_16x16x4_f64 tile.
// Same arithmetic as src/backward_pass.jl:162-252 + :28-42, :64-76.
//
// Tile coordinates (NP = n rounded up to a multiple of 4, a template parameter): state j -> j, control a -> NP + a, vectors in column 15.
//   * padded state rows/columns (n .. NP-1) are exact zeros in every operand and stay exact zeros through the recursion;
//   * the controls fill ONE accumulator register: register NP/4 of G = F'W + H holds, in the 16-lane row a, the row [Qux | Quu | Qu] of
//     control a.  It goes through a 64-entry LDS image so that every lane has all the rows of ITS column; the MS x MS system
//     (MS = 4, or 3 at NP = 12) is factorised redundantly by every lane (upper Cholesky with reciprocal pivots: the pivots are the
//     positive-definiteness test of :35-38) and each lane solves its own column — K for the columns < n, k for column 15 (:41-42);
//     controls m .. MS-1 are an identity block of the system (K rows exactly zero);
//   * value update (:69-72) for any m <= 4:  D = G + K'Y + Qux'[0 | k],  Y = (Quu K + Qux) + Qux,  V = ½(D + D')  (the transpose
//     comes through the padded LDS tile as in back_pass_mx.hip); column 15 is not symmetrised and wants K'(Quu k + Qu) + Qux'k: the
//     first term is the product with Y[:,15] = Quu k + Qu, the second a product of the control rows of G with k in column 15.
// Per step: 2·NP/4 + 2 (+ NP/4 for regType 2) dependent matrix instructions, two LDS round trips, ~170 other instructions (the wave
// is issue-bound: a second wave on the SIMD does not hide anything, so selects are multiplications by lane constants, the stores
// share exec-mask switches, and what a 64-cycle matrix instruction can replace of > 11 vector instructions goes to the matrix pipe).
// Measured against the 16-lane-row kernel it replaces for small and medium batches: DESIGN.md §3.2.
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

#include "back_pass_mx_common.h"

constexpr int VG = 15;                              // tile column of the vectors
#ifndef MXG_EXP
#define MXG_EXP 0          // timing experiments (wrong results): 1 no result stores, 2 no operand refills
#endif

typedef double d2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) d2 *g2p;            // explicit global accesses (a FLAT access would wait for the LDS too)
typedef __attribute__((address_space(1))) d2 *g2w;
typedef __attribute__((address_space(1))) double *g1w;
constexpr int even_up(int x) { return (x + 1) & ~1; }

// COAL (the default): what moves in time comes and goes in CONTIGUOUS 16-byte pieces.  profiles/r05_mxg_exp.txt: with one 8-byte
// element per lane (the tile layout: 16 runs of 32 bytes per instruction) the step at n=12, m=3 costs 0.86 ms at B=2048 against 0.48
// without any memory operation and 0.53 with only the loads or only the stores — the memory system serves the fine-grained mix at
// ~3.8 TB/s.  Here a step's operand blocks (fx | fu | cx | cu | cxx | cxu | cuu, whichever are time-varying: each contiguous in
// memory) are fetched as 16-byte pieces PDC steps ahead, dropped into an LDS image one step before use (two images, by the parity of
// the unrolled slot), and the tile operands are LDS reads; the results of a step are written into an LDS record in memory order
// (Vxx | K | Vx | k | Quu) and leave as 16-byte pieces (+ one 8-byte piece per block of odd length).  A piece of an odd-length block
// reads 8 bytes into the NEXT time step of the same array (never the last one: the loop starts at N-2) and ignores them.
// A/B switches of the limited instantiations (profiles/build_variant.sh; defaults = what ships; same-box table in DESIGN section 9):
#ifndef MXG_LIMS_PDC
#define MXG_LIMS_PDC 0      // != 0: unroll / prefetch distance of the limited kernels (default: as without limits)
#endif
#ifndef MXG_U_IMAGE
#define MXG_U_IMAGE 1       // u_i rides in the operand image (0: a load of its own per step, behind the stores in the in-order counter)
#endif
#ifndef MXG_GAIN22
#define MXG_GAIN22 1        // m <= 2: gain columns from the 2 x 2 factor on scalars (0: the padded MS x MS solve)
#endif
template <int NP> struct MxgLds {                   // doubles
    static constexpr int MS = NP == 12 ? 3 : 4;
    static constexpr int HC = 0, ZERO = 16 * VG, CONSTS = 256;        // offset of the time-invariant H tile [row + 16 col] (its column VG is zero), its size
    static constexpr int IMG = 2 * even_up(NP * NP) + 2 * even_up(NP * MS) + even_up(NP) + 2 * even_up(MS) + even_up(MS * MS);     // (+ u_i with limits)
    static constexpr int BUF = CONSTS + IMG;
    static constexpr int RECD = even_up(NP * NP) + even_up(NP * MS) + even_up(NP) + even_up(MS) + even_up(MS * MS);
    static constexpr int REC = RECD + 64;           // + one dump cell per lane
};

// the stores of a step under three exec masks (registers 0 .. KS-2 of V | the last one | K, k, Quu): one switch per mask, the row
// registers 32 bytes apart through the instruction's offset field
template <int KS>
__device__ __forceinline__ void store_results(char *vst, const double (&S)[KS], unsigned long long full, unsigned long long last,
                                              char *kq, double kv, unsigned long long lanesK)
{
    if constexpr (KS == 1)
        asm volatile("s_mov_b64 exec, %2\n\tglobal_store_dwordx2 %0, %1, off\n\ts_mov_b64 exec, %5\n\tglobal_store_dwordx2 %3, %4, off\n\ts_mov_b64 exec, -1"
                     ::"v"(vst), "v"(S[0]), "s"(last), "v"(kq), "v"(kv), "s"(lanesK) : "memory");
    else if constexpr (KS == 2)
        asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx2 %0, %1, off\n\ts_mov_b64 exec, %4\n\tglobal_store_dwordx2 %0, %2, off offset:32\n\t"
                     "s_mov_b64 exec, %7\n\tglobal_store_dwordx2 %5, %6, off\n\ts_mov_b64 exec, -1"
                     ::"v"(vst), "v"(S[0]), "v"(S[1]), "s"(full), "s"(last), "v"(kq), "v"(kv), "s"(lanesK) : "memory");
    else
        asm volatile("s_mov_b64 exec, %4\n\tglobal_store_dwordx2 %0, %1, off\n\tglobal_store_dwordx2 %0, %2, off offset:32\n\t"
                     "s_mov_b64 exec, %5\n\tglobal_store_dwordx2 %0, %3, off offset:64\n\t"
                     "s_mov_b64 exec, %8\n\tglobal_store_dwordx2 %6, %7, off\n\ts_mov_b64 exec, -1"
                     ::"v"(vst), "v"(S[0]), "v"(S[1]), "v"(S[2]), "s"(full), "s"(last), "v"(kq), "v"(kv), "s"(lanesK) : "memory");
}

#ifdef DDP_MXGPROF     // per-phase cycle counts (s_memtime) of trajectory 0: profiling builds only (profiles/build_variant.sh)
#define MXP_DECL long long mxp_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mxp_t = __builtin_amdgcn_s_memtime()
#define MXP(k) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_amdgcn_s_memtime(); mxp_[k] += t_ - mxp_t; mxp_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define MXP_PRINT do { if (b == 0 && lane == 0) printf("MXGPROF steps %d: operands %lld | products + control rows %lld | row reads %lld | box-QP %lld | gain solve %lld | K, T, Y %lld | value update %lld | transpose %lld | record + stores %lld | refill %lld\n", N - 1, \
    mxp_[0] / (N - 1), mxp_[1] / (N - 1), mxp_[2] / (N - 1), mxp_[3] / (N - 1), mxp_[4] / (N - 1), mxp_[5] / (N - 1), mxp_[6] / (N - 1), mxp_[7] / (N - 1), mxp_[8] / (N - 1), mxp_[9] / (N - 1)); } while (0)
#else
#define MXP_DECL
#define MXP(k)
#define MXP_PRINT
#endif

template <int NP, bool FXTV, bool CTV, bool REG2, bool COAL, bool LIMS = false>
// With limits: two waves per SIMD (<= 256 registers).  The limited instantiations sit at 250-256 vector registers and the allocator took a few
// accumulator registers on top in SOME of them — one wave per SIMD, i.e. two rounds for any batch of 1 025 .. 2 048 trajectories
// (n = 12, m = 3, B = 2 048: 1.51 or 2.13 ms depending on which side of 256 a build fell).
__global__ __launch_bounds__(DDP_WAVE) __attribute__((amdgpu_waves_per_eu(LIMS ? 2 : 1))) void back_pass_mxg_kernel(BPXArgs a)
{
    constexpr int KS = NP / 4, UR = NP / 4, MS = NP == 12 ? 3 : 4;
    using L = MxgLds<NP>;
    // 16-byte pieces of a step's time-varying operand blocks / of its result blocks -> vector-memory instructions per step
    constexpr int LPC = (FXTV ? even_up(NP * NP) / 2 + even_up(NP * MS) / 2 : 0) + even_up(NP) / 2 + even_up(MS) / 2 +
                        (CTV ? even_up(NP * NP) / 2 + even_up(NP * MS) / 2 + even_up(MS * MS) / 2 : 0) + ((LIMS && MXG_U_IMAGE) ? even_up(MS) / 2 : 0);
    constexpr int NLI = (LPC + 63) / 64, NSI = (L::RECD / 2 + 63) / 64;
    // prefetch distance = slots of the unrolled loop (even)
    constexpr int PDC = (LIMS && MXG_LIMS_PDC != 0) ? MXG_LIMS_PDC : (COAL ? (NLI <= 2 ? 8 : 4) : PD);
    const int b = blockIdx.x, lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    if (a.active && a.active[b] == 0) return;
    const int N = a.N, nr = a.n, mr = a.m;
    const size_t nn = (size_t)nr * nr, nm = (size_t)nr * mr, mm = (size_t)mr * mr;

    __shared__ __attribute__((aligned(16))) double lds[TLD * 16 + 16];      // transpose tile + zero cells
    __shared__ __attribute__((aligned(16))) double zl[2][64];               // the control rows of G (and of the regularised G): [a][column]
    __shared__ __attribute__((aligned(16))) double img[COAL ? 2 * L::BUF : 2];   // two operand buffers: constants | image of a step
    __shared__ __attribute__((aligned(16))) double rec[COAL ? L::REC : 2];      // the result record of a step

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

    // ---- per-lane operand streams (tile coordinate -> state index, control index, or nothing)
    auto six = [&](int r) { return r < nr ? r : -1; };
    auto uix = [&](int r) { return (r >= NP && r < NP + mr) ? r - NP : -1; };
    const Stream zeroS = Stream{(const char *)mx_zero, 0u, nullptr};
    auto h_stream = [&](int row, int col) -> Stream {       // H = [cxx cxu; cxu' cuu] in tile coordinates; identity for the unused controls
        const int sr = six(row), sc = six(col), ur = uix(row), uc = uix(col);
        if (sr >= 0 && sc >= 0) return Stream{(const char *)(cxx + sr + nr * sc), CTV ? (unsigned)(nn * 8) : 0u, nullptr};
        if (sr >= 0 && uc >= 0) return Stream{(const char *)(cxu + sr + nr * uc), CTV ? (unsigned)(nm * 8) : 0u, nullptr};
        if (ur >= 0 && sc >= 0) return Stream{(const char *)(cxu + sc + nr * ur), CTV ? (unsigned)(nm * 8) : 0u, nullptr};
        if (ur >= 0 && uc >= 0) return Stream{(const char *)(cuu + ur + mr * uc), CTV ? (unsigned)(mm * 8) : 0u, nullptr};
        if (row == col && row >= NP + mr && row < NP + MS) return Stream{(const char *)mx_one, 0u, nullptr};
        return zeroS;
    };
    auto f_stream = [&](int row, int col) -> Stream {       // F = [fx fu] in tile coordinates, zero outside
        const int sr = six(row), sc = six(col), uc = uix(col);
        if (sr >= 0 && sc >= 0) return Stream{(const char *)(fx + sr + nr * sc), FXTV ? (unsigned)(nn * 8) : 0u, nullptr};
        if (sr >= 0 && uc >= 0) return Stream{(const char *)(fu + sr + nr * uc), FXTV ? (unsigned)(nm * 8) : 0u, nullptr};
        return zeroS;
    };
    auto e_stream = [&](int row) -> Stream {                // column VG of the C operand: [cx; cu] (:239-241)
        if (six(row) >= 0) return Stream{(const char *)(cx + row), (unsigned)(nr * 8), nullptr};
        if (uix(row) >= 0) return Stream{(const char *)(cu + (row - NP)), (unsigned)(mr * 8), nullptr};
        return zeroS;
    };
    Stream cS[4], fS[KS];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        cS[s] = l15 == VG ? e_stream(l4 + 4 * s) : h_stream(l4 + 4 * s, l15);
        if (s < KS) fS[s] = f_stream(l4 + 4 * s, l15);
    }

    // ---- loop-invariant lane constants
    const double maskV = l15 == VG ? 1.0 : 0.0, maskM = l15 == VG ? 0.0 : 1.0;
    const double rowm[4] = {l4 == 0 ? 1.0 : 0.0, l4 == 1 ? 1.0 : 0.0, l4 == 2 ? 1.0 : 0.0, (l4 == 3 && MS > 3) ? 1.0 : 0.0};
    const int wr = l4 + TLD * l15;                           // accumulator register s -> tile element (l4+4s, l15)
    const int rdT = l15 == VG ? TZERO : l15 + TLD * l4;      // its transpose (l15, l4+4s): + 4*TLD per register
    const int rdS = l15 == VG ? 0 : 4 * TLD;
    const bool v_col = l15 < nr || l15 == VG;
    const double vscl = l15 == VG ? 1.0 : 0.5;               // V = ½(D + D'); column VG: Vx = D[:,VG] (its transposed read is a zero cell)
    char *vst = l15 == VG ? (char *)(Vxg + (size_t)nr * (tl - 1) + l4) : (char *)(Vxxg + nn * (tl - 1) + l4 + nr * (l15 < nr ? l15 : 0));
    const unsigned vst_stride = l15 == VG ? (unsigned)(nr * 8) : (unsigned)(nn * 8);
    // rows 4s .. 4s+3 of the registers s < KS-1 are all states (NP = n rounded up to 4): one mask for them, one for the last register
    const unsigned long long lanesVf = __builtin_amdgcn_ballot_w64(v_col), lanesVl = __builtin_amdgcn_ballot_w64(v_col && l4 + 4 * (KS - 1) < nr);
    // K | k | Quu ride on one store: lane (a = l4, column): columns < n: K[a, col]; column VG: k[a]; control columns: Quu[a, col - NP]
    const bool quu_lane = uix(l15) >= 0;
    const unsigned long long lanesK = __builtin_amdgcn_ballot_w64(l4 < mr && (l15 < nr || l15 == VG || quu_lane));
    char *kq = l15 < nr ? (char *)(Kg + nm * (tl - 1) + l4 + mr * l15)
                        : (l15 == VG ? (char *)(kg + (size_t)mr * (tl - 1) + l4) : (char *)(Quug + mm * (tl - 1) + l4 + mr * (quu_lane ? l15 - NP : 0)));
    const unsigned kq_stride = l15 < nr ? (unsigned)(nm * 8) : (l15 == VG ? (unsigned)(mr * 8) : (unsigned)(mm * 8));

    // ---- register-resident operands: ring of PDC steps for what moves in time
    double F[KS];                                            // F_s: B of GEMM1, A of GEMM2
    double cr[COAL ? 1 : PDC][4], fr[(FXTV && !COAL) ? PDC : 1][KS];
    const int i0 = N - 2;
    if constexpr (!COAL) {
#pragma unroll
        for (int j = 0; j < PDC; ++j) {
            const int t = i0 - j > 0 ? i0 - j : 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                cr[j][s] = cS[s].at(t);
                if (FXTV && s < KS) fr[j][s] = fS[s].at(t);
            }
        }
        const int t = i0 - PDC > 0 ? i0 - PDC : 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) { cS[s].seek(t); if (s < KS) fS[s].seek(t); }
    }
    if (!FXTV) {
#pragma unroll
        for (int s = 0; s < KS; ++s) F[s] = fS[s].at(0);
    }
    // ---- COAL: piece tables.  lg/lgs/lim: my 16-byte piece of instruction j of a step's operand blocks (global pointer, bytes per
    // step, byte offset in an image); co/fo: where my tile operands sit in a buffer; wS/wK: where my results go in the record;
    // sp/sstr/srd: my 16-byte piece of instruction j of the record; tp/tstr/trd: the last element of a block of odd length
    const char *lg[NLI];
    unsigned lgs[NLI], lim[NLI], co[4], fo[KS], wS[KS], wK = 0, sstr[NSI], srd[NSI], tstr = 0, trd = 0, uo = 0;
    const double *ug = LIMS ? a.u + (size_t)mr * N * b : nullptr;
    char *sp[NSI], *tp = nullptr;
    bool has4 = false, hast = false;
    d2 ring[COAL ? PDC : 1][NLI];
    if constexpr (COAL) {
        // (with limits u_i rides in the image as well: a load of its own would sit in the SAME in-order counter as the operand ring and the
        // stores — the first limited build waited at the top of the box-QP for the stores and the refill of the step before, 2 000 ticks
        // of memory latency per step that looked like the QP's arithmetic in every profile)
        const double *bp[8] = {fx, fu, cx, cu, cxx, cxu, cuu, ug};
        const int bsz[8] = {(int)nn, (int)nm, nr, mr, (int)nn, (int)nm, (int)mm, mr};
        const bool bon[8] = {FXTV, FXTV, true, true, CTV, CTV, CTV, LIMS && MXG_U_IMAGE};
        int off[8], o = L::CONSTS;
#pragma unroll
        for (int q = 0; q < 8; ++q) { off[q] = o; if (bon[q]) o += even_up(bsz[q]); }
        uo = (unsigned)off[7] * 8u;
#pragma unroll
        for (int j = 0; j < NLI; ++j) {
            int pi = lane + 64 * j;
            bool found = false;
            lg[j] = (const char *)cx; lgs[j] = nr * 8u; lim[j] = (unsigned)off[2] * 8u;     // lanes past the end repeat piece 0 of cx
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (!bon[q]) continue;
                const int np = (bsz[q] + 1) / 2;
                if (!found && pi < np) { lg[j] = (const char *)bp[q] + 16 * pi; lgs[j] = (unsigned)bsz[q] * 8u; lim[j] = (unsigned)(off[q] + 2 * pi) * 8u; found = true; }
                if (!found) pi -= np;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = l4 + 4 * r, sr = six(row), sc = six(l15), ur = uix(row), uc = uix(l15);
            int e = L::HC + row + 16 * l15;                                                    // the constant tile (zero / identity / time-invariant H)
            if (l15 == VG) e = sr >= 0 ? off[2] + sr : (ur >= 0 ? off[3] + ur : L::ZERO);
            else if (CTV) {
                if (sr >= 0 && sc >= 0) e = off[4] + sr + nr * sc;
                else if (sr >= 0 && uc >= 0) e = off[5] + sr + nr * uc;
                else if (ur >= 0 && sc >= 0) e = off[5] + sc + nr * ur;
                else if (ur >= 0 && uc >= 0) e = off[6] + ur + mr * uc;
            }
            co[r] = (unsigned)e * 8u;
            if (r < KS) fo[r] = (unsigned)((sr >= 0 && sc >= 0) ? off[0] + sr + nr * sc : ((sr >= 0 && uc >= 0) ? off[1] + sr + nr * uc : L::ZERO)) * 8u;
            // constants of both buffers
            double hv = 0.0;
            if (l15 != VG) hv = CTV ? ((row == l15 && row >= NP + mr && row < NP + MS) ? 1.0 : 0.0) : h_stream(row, l15).at(0);
            img[L::HC + row + 16 * l15] = hv; img[L::BUF + L::HC + row + 16 * l15] = hv;
        }
        // the record: Vxx | K | Vx | k | Quu in memory order, every block at an even offset
        const int rsz[5] = {(int)nn, (int)nm, nr, mr, (int)mm};
        char *rp[5] = {(char *)(Vxxg + nn * (tl - 1)), (char *)(Kg + nm * (tl - 1)), (char *)(Vxg + (size_t)nr * (tl - 1)), (char *)(kg + (size_t)mr * (tl - 1)),
                       (char *)(Quug + mm * (tl - 1))};
        int ro[5], o2 = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) { ro[q] = o2; o2 += even_up(rsz[q]); }
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
            const int row = l4 + 4 * s2;
            wS[s2] = (unsigned)((l15 < nr && row < nr) ? ro[0] + row + nr * l15 : ((l15 == VG && row < nr) ? ro[2] + row : L::RECD + lane)) * 8u;
        }
        wK = (unsigned)(l4 < mr ? (l15 < nr ? ro[1] + l4 + mr * l15 : (l15 == VG ? ro[3] + l4 : (uix(l15) >= 0 ? ro[4] + l4 + mr * (l15 - NP) : L::RECD + lane))) : L::RECD + lane) * 8u;
        int first = -1, ntail = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) if (first < 0 && rsz[q] >= 2) first = q;
        has4 = first >= 0;
#pragma unroll
        for (int j = 0; j < NSI; ++j) {
            int pi = lane + 64 * j;
            bool found = false;
            const int f0 = first >= 0 ? first : 0;
            sp[j] = rp[f0]; sstr[j] = (unsigned)rsz[f0] * 8u; srd[j] = (unsigned)ro[f0] * 8u;                 // lanes past the end repeat the first piece
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int np = rsz[q] / 2;
                if (!found && pi < np) { sp[j] = rp[q] + 16 * pi; sstr[j] = (unsigned)rsz[q] * 8u; srd[j] = (unsigned)(ro[q] + 2 * pi) * 8u; found = true; }
                if (!found) pi -= np;
            }
        }
#pragma unroll
        for (int q = 4; q >= 0; --q)
            if (rsz[q] & 1) { ++ntail; }
        hast = ntail > 0;
        {
            int tq = -1, seen = 0, firstodd = -1;
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (rsz[q] & 1) { if (firstodd < 0) firstodd = q; if (seen == lane) tq = q; ++seen; }
            if (tq < 0) tq = firstodd >= 0 ? firstodd : 0;                                                     // lanes past the end repeat the first one
            tp = rp[tq] + (size_t)(rsz[tq] - 1) * 8; tstr = (unsigned)rsz[tq] * 8u; trd = (unsigned)(ro[tq] + rsz[tq] - 1) * 8u;
        }
        // operand pieces of the first PDC steps; the image of step N-2 goes into buffer 0
#pragma unroll
        for (int s2 = 0; s2 < PDC; ++s2) {
            const int t = i0 - s2 > 0 ? i0 - s2 : 0;
#pragma unroll
            for (int j = 0; j < NLI; ++j) ring[s2][j] = *(g2p)(lg[j] + (size_t)lgs[j] * (unsigned)t);
        }
#pragma unroll
        for (int j = 0; j < NLI; ++j) {
            lg[j] += (size_t)lgs[j] * (unsigned)(i0 - PDC > 0 ? i0 - PDC : 0);
            *(d2 *)((char *)img + lim[j]) = ring[0][j];
        }
    }

    // value function of the terminal step in tile layout: S = Vxx, column VG: Vx
    double S[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int row = l4 + 4 * s;
        S[s] = (l15 < nr && row < nr) ? cxx[(CTV ? nn * tl : 0) + row + nr * l15] : ((l15 == VG && row < nr) ? cx[(size_t)nr * tl + row] : 0.0);
    }
    const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
    wave_sync();
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): all set-up loads have landed

    // ---- control limits (backward_pass.jl:43-62): the box-QP on the MS x MS system — the same solve on the same (LDS-broadcast) data in every
    // lane, so its branches are wave-uniform (boxqp_dev.h); controls past m: gradient 1 on [0, 0], clamped in every iteration.  u_i comes
    // with the operand image of its step.
    bool nolims = true;
    double limlo[MS], limhi[MS], ucur[MS], kprev[MS];
#pragma unroll
    for (int c2 = 0; c2 < MS; ++c2) { limlo[c2] = 0.0; limhi[c2] = 0.0; ucur[c2] = 0.0; kprev[c2] = 0.0; }
    if constexpr (LIMS) {
        nolims = a.lims[0] > a.lims[mr];                    // backward_pass.jl:31
#pragma unroll
        for (int c2 = 0; c2 < MS; ++c2)
            if (c2 < mr) { limlo[c2] = a.lims[c2]; limhi[c2] = a.lims[c2 + mr]; ucur[c2] = ug[(size_t)mr * i0 + c2]; }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35
    double dVa = 0.0, dVp = 0.0;                    // Σ k'Qu (lanes of column VG) and the per-row parts of Σ k'(Quu k + Qu)
    int diverge = 0;
    MXP_DECL;
    auto reg = [](const d4 &v, int r) __attribute__((always_inline)) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); };
    // One time step; no exits inside (a diverged trajectory steps through garbage until the loop around the step looks at `diverge`;
    // its outputs below the failing step are zero-filled after the loop).
    auto step = [&](const int i, auto slot_c) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        const bool okp = diverge == 0;
        d4 c;
        if constexpr (COAL) {
            // my tile operands from the image of this step (buffer = parity of the slot), then the image of step i-1 into the other one
            constexpr int cur = slot & 1, nslot = (slot + 1) % PDC;
            const char *bufc = (const char *)img + cur * L::BUF * 8;
            c = d4{*(const double *)(bufc + co[0]), *(const double *)(bufc + co[1]), *(const double *)(bufc + co[2]), *(const double *)(bufc + co[3])};
            if (FXTV) {
#pragma unroll
                for (int s = 0; s < KS; ++s) F[s] = *(const double *)(bufc + fo[s]);
            }
            char *bufn = (char *)img + (cur ^ 1) * L::BUF * 8;
#pragma unroll
            for (int j = 0; j < NLI; ++j) *(d2 *)(bufn + lim[j]) = ring[nslot][j];
        } else {
            c = d4{cr[slot][0], cr[slot][1], cr[slot][2], cr[slot][3]};
            if (FXTV) {
#pragma unroll
                for (int s = 0; s < KS; ++s) F[s] = fr[slot][s];
            }
        }
        MXP(0);
        // ================= GEMM1: W = Vxx·F; column VG := Vx (F[:,VG] = 0, S[:,VG] = Vx) ============================
        d4 w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[0], F[0], zero4, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < KS; ++s) w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[s], F[s], w, 0, 0, 0);
        double W[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) W[s] = fma(S[s], maskV, reg(w, s));
        // ================= GEMM2: G = F'W + H, column VG: [cx;cu] + F'Vx  (:203-210, :239-247) =====================
        d4 g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], W[0], c, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < KS; ++s) g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[s], W[s], g, 0, 0, 0);
        // lane (a, col): G[control a][col] = Qux | Quu | Qu  (through the vector ALU where a row broadcast reads it: regType 2)
        const double Z = (REG2 || LIMS) ? reg(g, UR) + 0.0 : reg(g, UR);
        zl[0][lane] = Z;
        if (REG2) {                                        // control rows of F'(W + λF) + H: Qux_reg, QuuF (:205-207)
            d4 gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], fma(lam, F[0], W[0]), c, 0, 0, 0);
#pragma unroll
            for (int s = 1; s < KS; ++s) gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[s], fma(lam, F[s], W[s]), gr, 0, 0, 0);
            zl[1][lane] = reg(gr, UR) + 0.0;
        }
        wave_sync();
        MXP(1);
        // ================= gains (backward_pass.jl:30-42): every lane factorises QuuF, solves its own column ===========
        const double *zs = zl[REG2 ? 1 : 0];
        double Hq[MS * MS], R[MS * MS], ri[MS], q[MS];
#pragma unroll
        for (int c2 = 0; c2 < MS; ++c2) {
            q[c2] = zs[16 * c2 + l15];
#pragma unroll
            for (int c1 = 0; c1 <= c2; ++c1) Hq[c1 + MS * c2] = zs[16 * c1 + NP + c2] + ((!REG2 && c1 == c2) ? lam : 0.0);
        }
        double Qu[MS];                                     // unregularised rows at my column (dV; column VG: Qu)
#pragma unroll
        for (int c2 = 0; c2 < MS; ++c2) Qu[c2] = REG2 ? zl[0][16 * c2 + l15] : q[c2];
        int fail;
#ifdef DDP_MXGPROF
        asm volatile("" : "+v"(q[0]), "+v"(Hq[0]), "+v"(Qu[MS - 1]));
#endif
        MXP(2);
        if (!LIMS || nolims) {
            fail = ddp_chol_rinv<MS>(Hq, R, ri);
            ddp_rsolve_neg<MS>(R, ri, q);                  // q <- -(QuuF)\q: K[:, col] (:42), column VG: k_i (:41)
        } else {
            if (MXG_U_IMAGE) {   // u_i: from the image of this step
                const double *ui = (const double *)((const char *)img + (slot & 1) * L::BUF * 8 + uo);
#pragma unroll
                for (int c2 = 0; c2 < MS; ++c2)
                    if (c2 < mr) ucur[c2] = ui[c2];
            }
            bool tail = true;                               // the padded MS x MS gain solve below
            double kk[MS];
            unsigned clamped = 0u;
            int iters, result;
            if (mr <= 2) {
                // (uniform) the usual sizes as a 2 x 2 problem on scalars
                const double h01 = Hq[MS];
                double H2[4] = {Hq[0], h01, h01, Hq[1 + MS]}, g2[2] = {zl[0][VG], mr > 1 ? zl[0][16 + VG] : 1.0},        // Qu: the gradient column of the unregularised rows
                       lo2[2] = {limlo[0] - ucur[0], mr > 1 ? limlo[1] - ucur[1] : 0.0}, up2[2] = {limhi[0] - ucur[0], mr > 1 ? limhi[1] - ucur[1] : 0.0},      // (:45-46)
                       x02[2] = {kprev[0], kprev[1]}, k2[2], R2[4], ri2[2];
                // (:49), warm start k[:, min(i+1, N-1)]
                result = boxqp_dev2(H2, g2, lo2, up2, x02, qpo, k2, R2, ri2, clamped, iters);     // straight-line (boxqp_dev.h)
#ifdef DDP_MXGPROF
                asm volatile("" : "+v"(k2[0]), "+v"(R2[0]));
#endif
                MXP(3);
                if (MXG_GAIN22) {                          // K[free, col] = -(R'R)\Qux_reg[free, col] with the 2 x 2 factor (:57-61)
                    tail = false;
                    kprev[0] = k2[0]; kprev[1] = k2[1];
                    const bool cl0 = (clamped & 1u) != 0, cl1 = (clamped & 2u) != 0;
                    double b0 = (cl0 ? 0.0 : q[0]) * ri2[0], t2 = cl1 ? 0.0 : q[1];                            // chol_solve_ri<2>
                    t2 -= R2[2] * b0;
                    double b1 = t2 * ri2[1];
                    b1 = b1 * ri2[1]; t2 = b0; t2 -= R2[2] * b1; b0 = t2 * ri2[0];
                    q[0] = l15 == VG ? k2[0] : (cl0 ? 0.0 : -b0);                                              // column VG: k_i from the QP (bounds included)
                    q[1] = l15 == VG ? k2[1] : (cl1 ? 0.0 : -b1);
#pragma unroll
                    for (int c2 = 2; c2 < MS; ++c2) q[c2] = 0.0;                                               // the controls past 2: clamped
                } else {
#pragma unroll
                    for (int e = 0; e < MS * MS; ++e) R[e] = 0.0;
#pragma unroll
                    for (int c2 = 0; c2 < MS; ++c2) { R[c2 + MS * c2] = 1.0; ri[c2] = 1.0; kk[c2] = 0.0; }
                    R[0] = R2[0]; R[MS] = R2[2]; R[1 + MS] = R2[3]; ri[0] = ri2[0]; ri[1] = ri2[1]; kk[0] = k2[0]; kk[1] = k2[1];
                    clamped |= ((1u << MS) - 1u) & ~3u;    // the controls past 2: clamped
                }
            } else {
                double Hf[MS * MS], gq[MS], lo[MS], up[MS];
#pragma unroll
                for (int c2 = 0; c2 < MS; ++c2) {
#pragma unroll
                    for (int c1 = 0; c1 <= c2; ++c1) { Hf[c1 + MS * c2] = Hq[c1 + MS * c2]; Hf[c2 + MS * c1] = Hq[c1 + MS * c2]; }
                    gq[c2] = c2 < mr ? zl[0][16 * c2 + VG] : 1.0;                  // Qu (the gradient column of the unregularised rows)
                    lo[c2] = c2 < mr ? limlo[c2] - ucur[c2] : 0.0; up[c2] = c2 < mr ? limhi[c2] - ucur[c2] : 0.0;      // (:45-46)
                }
                result = boxqp_dev_ri<MS>(MS, Hf, gq, lo, up, kprev, qpo, kk, R, ri, clamped, iters);
                MXP(3);
            }
            fail = result < 1;                             // (:53)
            if (tail) {
#pragma unroll
                for (int c2 = 0; c2 < MS; ++c2) { kprev[c2] = kk[c2]; q[c2] = ((clamped >> c2) & 1u) ? 0.0 : q[c2]; }
                chol_solve_ri<MS>(MS, R, ri, q);           // K[free, col] = -(R'R)\Qux_reg[free, col], clamped rows zero (:57-61)
#pragma unroll
                for (int c2 = 0; c2 < MS; ++c2) {
                    const double kq_ = ((clamped >> c2) & 1u) ? 0.0 : -q[c2];
                    q[c2] = l15 == VG ? kk[c2] : kq_;      // column VG: k_i from the QP (bounds included)
                }
            }
            if (!MXG_U_IMAGE) {
#pragma unroll
                for (int c2 = 0; c2 < MS; ++c2)            // u_{i-1} for the next step
                    if (c2 < mr) ucur[c2] = ug[(size_t)mr * (i > 0 ? i - 1 : 0) + c2];
            }
        }
#ifdef DDP_MXGPROF
        asm volatile("" : "+v"(q[0]), "+v"(q[1]));
#endif
        MXP(4);
        double Ksel = q[0] * rowm[0];                      // K[a = l4, col]: a sum with lane constants 1 / 0, no selects
#pragma unroll
        for (int c2 = 1; c2 < MS; ++c2) Ksel = fma(q[c2], rowm[c2], Ksel);
        // T_a = Quu[a,:]·K + Qux_a (:64) for my row a = l4
        double Tsel;
        if (!REG2 && (!LIMS || nolims)) {
            Tsel = -lam * Ksel;                            // regType 1: (Quu + λI) K = -Qux (not so for clamped rows or the k of a box-QP)
        } else {
            Tsel = Z;
            static_for<0, MS>([&](auto bc) __attribute__((always_inline)) {
                constexpr int bb = decltype(bc)::value;
                if (bb == 0) fmac_bcast<NP + bb, 0xf, true>(Tsel, Z, q[bb]); else fmac_bcast<NP + bb>(Tsel, Z, q[bb]);
            });
        }
        const double Ysel = fma(Z, maskM, Tsel);           // Y = T + Qux; column VG: Quu k + Qu
#ifdef DDP_MXGPROF
        { double y_ = Ysel; asm volatile("" : "+v"(y_)); }
#endif
        MXP(5);
        // ================= value update (:69-72): D = G + K'Y, column VG also + Qux'k ==================================
        d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(Ksel, Ysel, g, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f64_16x16x4f64(Z, Ksel * maskV, v, 0, 0, 0);     // A[i][a] = G[control a][i], B[a][VG] = k_a
        const bool badu = fail != 0;                       // the same in every lane
        if (__builtin_expect(badu || !okp, 0)) {
            asm volatile("" ::: "memory");
            if (okp) diverge = i + 1;                      // diverge = i (:37-38)
        } else {                                           // column VG: k'Qu and k_a (Quu k + Qu)_a  (:68)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int c2 = 0; c2 < MS; ++c2) dVa = fma(q[c2], Qu[c2], dVa);
            dVp = fma(Ksel, Tsel, dVp);
        }
#ifdef DDP_MXGPROF
        asm volatile("" : "+v"(v.x));
#endif
        MXP(6);
        // ---- ½(D + D') through the transpose tile (column VG: Vx, not symmetrised)
#pragma unroll
        for (int s = 0; s < KS; ++s) lds[wr + 4 * s] = reg(v, s);
        wave_sync();
#pragma unroll
        for (int s = 0; s < KS; ++s) S[s] = vscl * (reg(v, s) + lds[rdT + s * rdS]);
        // Stores are unconditional: a diverged trajectory writes garbage into time steps that are zero-filled after the loop
        const double kv = quu_lane ? Z : Ksel;             // K | k | Quu (:75-76)
#ifdef DDP_MXGPROF
        asm volatile("" : "+v"(S[0]));
#endif
        MXP(7);
        if constexpr (COAL) {
#pragma unroll
            for (int s = 0; s < KS; ++s) *(double *)((char *)rec + wS[s]) = S[s];
            *(double *)((char *)rec + wK) = kv;
            wave_sync();
            if (!(MXG_EXP & 1)) {
                if (has4) {
                    d2 pc[NSI];
#pragma unroll
                    for (int j = 0; j < NSI; ++j) pc[j] = *(const d2 *)((const char *)rec + srd[j]);
#pragma unroll
                    for (int j = 0; j < NSI; ++j) { *(g2w)sp[j] = pc[j]; sp[j] -= sstr[j]; }
                }
                if (hast) {
                    const double tv = *(const double *)((const char *)rec + trd);
                    *(g1w)tp = tv; tp -= tstr;
                }
            }
            wave_sync();                                   // the tile, the image and the record are free again
            MXP(8);
            if (!(MXG_EXP & 2)) {   // refill the ring slot with the step PDC ahead (clamped: always a valid load)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < NLI; ++j) ring[slot][j] = *(g2p)lg[j];
                if (i - PDC > 0) {
#pragma unroll
                    for (int j = 0; j < NLI; ++j) lg[j] -= lgs[j];
                }
            }
        } else {
            if (!(MXG_EXP & 1)) store_results<KS>(vst, S, lanesVf, lanesVl, kq, kv, lanesK);      // Vxx | Vx, K | k | Quu
            vst -= vst_stride;
            kq -= kq_stride;
            wave_sync();                                   // the tile and the image are free again
            if (!(MXG_EXP & 2)) {   // refill the ring slot with the step PDC ahead (clamped: always a valid load)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    cr[slot][s] = cS[s].next();
                    if (FXTV && s < KS) fr[slot][s] = fS[s].next();
                }
                if (i - PDC > 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        cS[s].back();
                        if (FXTV && s < KS) fS[s].back();
                    }
                }
            }
        }
    };
    int i = i0;
    while (i >= PDC - 1 && diverge == 0) {
        static_for<0, PDC>([&](auto sc) __attribute__((always_inline)) { step(i - decltype(sc)::value, sc); });
        i -= PDC;
    }
    static_for<0, PDC - 1>([&](auto sc) __attribute__((always_inline)) {    // the last (N-1) mod PDC steps
        if (i >= 0 && diverge == 0) { step(i, sc); --i; }
    });

    MXP_PRINT;
#ifdef DDP_QP2_STATS
    if (b == 0 && lane == 0) printf("QP2STATS (all launches so far) first-iteration exit 6: %llu | finished straight-line: %llu (iterations %llu; result 0: %llu 2: %llu 4: %llu 5: %llu 6: %llu) | generic loop: %llu\n",
        ddp_qp2_stats[10], ddp_qp2_stats[1], ddp_qp2_stats[2], ddp_qp2_stats[3], ddp_qp2_stats[5], ddp_qp2_stats[7], ddp_qp2_stats[8], ddp_qp2_stats[9], ddp_qp2_stats[0]);
#endif
    if (diverge) {   // outputs earlier in time than the failing step are zero (backward_pass.jl:226-229)
        const size_t ie = (size_t)diverge;          // = i + 1
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): the garbage of the steps after the failure has landed
        for (size_t e = lane; e < nm * ie; e += DDP_WAVE) Kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)mr * ie; e += DDP_WAVE) kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)nr * ie; e += DDP_WAVE) Vxg[e] = 0.0;
        for (size_t e = lane; e < nn * ie; e += DDP_WAVE) Vxxg[e] = 0.0;
        for (size_t e = lane; e < mm * (ie - 1); e += DDP_WAVE) Quug[e] = 0.0;
    }
    {   // dV (:68): [Σ k'Qu, ½ Σ k'Quu k];  k'Quu k = k'(Quu k + Qu) - k'Qu, the control rows live in lanes VG, 16+VG, 32+VG, 48+VG
        double kT = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int plo = __builtin_amdgcn_readlane(__double2loint(dVp), 16 * r + VG), phi = __builtin_amdgcn_readlane(__double2hiint(dVp), 16 * r + VG);
            if (r < MS) kT += __hiloint2double(phi, plo);
        }
        if (lane == VG) { a.dV[2 * b] = dVa; a.dV[2 * b + 1] = 0.5 * (kT - dVa); }
    }
    if (lane == 0) a.diverge[b] = diverge;
}

template <int NP, bool REG2, bool COAL>
int launch_mxg(ddp_handle h, const ddp_bp_desc *d, const BPXArgs &a)
{
    const dim3 grid(d->B), block(DDP_WAVE);
    const int key = (d->fx_tv ? 2 : 0) | (d->cost_tv ? 1 : 0);
    if constexpr (COAL) {
        if (d->has_lims) {
            switch (key) {
            case 0: hipLaunchKernelGGL((back_pass_mxg_kernel<NP, false, false, REG2, true, true>), grid, block, 0, h->stream, a); break;
            case 1: hipLaunchKernelGGL((back_pass_mxg_kernel<NP, false, true, REG2, true, true>), grid, block, 0, h->stream, a); break;
            case 2: hipLaunchKernelGGL((back_pass_mxg_kernel<NP, true, false, REG2, true, true>), grid, block, 0, h->stream, a); break;
            case 3: hipLaunchKernelGGL((back_pass_mxg_kernel<NP, true, true, REG2, true, true>), grid, block, 0, h->stream, a); break;
            }
            DDP_HIP(hipGetLastError());
            return 0;
        }
    }
    switch (key) {
    case 0: hipLaunchKernelGGL((back_pass_mxg_kernel<NP, false, false, REG2, COAL>), grid, block, 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((back_pass_mxg_kernel<NP, false, true, REG2, COAL>), grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((back_pass_mxg_kernel<NP, true, false, REG2, COAL>), grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL((back_pass_mxg_kernel<NP, true, true, REG2, COAL>), grid, block, 0, h->stream, a); break;
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// returns 1 if this shape is not handled here, 0 launched, <0 error
int ddp_launch_back_pass_mxg(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u, const int32_t *active, double *K,
                             double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge)
{
    if (d->m > 4 || d->n > 12 || d->n + d->m > 15) return 1;
    if (d->has_lims && !(lims && u)) return 1;
    const int np = d->n <= 4 ? 4 : (d->n <= 8 ? 8 : 12);
    if (np == 12 && d->m > 3) return 1;
    BPXArgs a;
    a.N = d->N; a.B = d->B; a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched; a.n = d->n; a.m = d->m;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge; a.lims = lims; a.u = u;
    const bool r2 = d->regType == 2;
    const char *ce = ddp_env(h, ENV_MXG_COAL);                  // 0: one 8-byte element per lane straight from / to global memory (A/B, tests)
    if (ce && ce[0] == '0') {
        if (d->has_lims) return 1;                                // (limits: the piece-wise path only)
        switch (np) {
        case 4: return r2 ? launch_mxg<4, true, false>(h, d, a) : launch_mxg<4, false, false>(h, d, a);
        case 8: return r2 ? launch_mxg<8, true, false>(h, d, a) : launch_mxg<8, false, false>(h, d, a);
        default: return r2 ? launch_mxg<12, true, false>(h, d, a) : launch_mxg<12, false, false>(h, d, a);
        }
    }
    switch (np) {
    case 4: return r2 ? launch_mxg<4, true, true>(h, d, a) : launch_mxg<4, false, true>(h, d, a);
    case 8: return r2 ? launch_mxg<8, true, true>(h, d, a) : launch_mxg<8, false, true>(h, d, a);
    default: return r2 ? launch_mxg<12, true, true>(h, d, a) : launch_mxg<12, false, true>(h, d, a);
    }
}
