// back_pass_mx2.hip — the fp64-MFMA-tile backward pass of back_pass_mx.hip (n = 10, m = 2, no control limits; same tile
// layouts, same arithmetic: src/backward_pass.jl:162-252 + :28-42,:64-76) with TWO waves per trajectory:
//
//   chain wave   the Riccati recursion, nothing else: per step 7 matrix instructions, the 2x2 gain solve, three LDS writes of the
//                step's results into a record.  No global store, no address arithmetic for results, no wait on vector memory.
//   writer wave  (same SIMD as its chain wave, lower priority: it runs in the chain's stalls) copies finished groups of 8 step
//                records LDS -> HBM with 16-byte stores, fetches [cx;cu] of the group after next by a direct-to-LDS load, and
//                forms Vxx_i = ½(V + V') for the records the chain left unsymmetrised (below).
//
// Why (removal experiments on back_pass_mx at B = 1024, profiles/ab_mx_exp.sh): the group write-back costs the lone chain wave
// 0.055 ms of 0.485 (11 vector-memory instructions per 8 steps at ~50 issue cycles each), the LDS round trip of ½(V + V') 0.037,
// both together 0.117 — at one wave per SIMD the step is a chain of dependent instructions and everything on it counts in full.
//
// ½(V + V') every FOURTH step only.  The reference symmetrises Vxx_i in every step (:71-72).  With V = Vs + E (E antisymmetric,
// rounding-sized) carried instead, the next step sees A = V' as its left operand (the accumulator read as an MFMA A operand), i.e.
// the same Vs and -E: Qxx gets -A'EA, Quu the antisymmetric -B'EB, and Vs changes by O(|E|) — E itself is multiplied by the
// OPEN-loop dynamics, |E_i| <= rho(A)² |E_{i+1}| + eps |V|.  Symmetrising exactly (bit-symmetric, through the LDS tile) every
// 4th step bounds |E| by ~rho(A)^6 eps |V| (rho = 3: 1e-13 relative); the records of the steps in between hold V and the writer
// stores ½(V + V'), so every Vxx_i that leaves the kernel is exactly symmetric, as in back_pass_mx.  (Never symmetrising was the
// defect the randomised sweep found in round 2: 5e-8 after 210 steps at rho = 1.05.)
//
// Hand-off: per trajectory two record buffers and two counters in the LDS — `ready` (groups finished by the chain, | FIN at the
// end) and `done` (groups written back).  LDS operations of the CU execute in issue order, so a counter written after the
// records is seen after them; the chain looks at `done` half a group before it needs the buffer, so it never waits in practice.
#include "ddp_internal.h"

namespace {

#include "back_pass_mx_common.h"

constexpr int NT = 4;                               // trajectories (chain waves) per work-group; waves NT .. 2 NT - 1 write back
constexpr int NB = 2;                               // record buffers per trajectory
constexpr int NE = 3;                               // [cx;cu] images per trajectory (the writer fetches group g+2 while it copies group g)
constexpr int SYM_EVERY = 4;                        // the chain symmetrises every SYM_EVERY-th step of a group
constexpr int FIN = 1 << 30, DONE_ALL = 1 << 29;

struct MxLds {                                      // per trajectory
    double tile[TLD * 16 + 16];                     // transpose tile + zero cells
    double lout[NB * LOUT + LDUMP_SZ];              // step records of NB groups; behind them the cells lanes without an output write to
    double leb[NE][128];                            // [cx;cu] of a group: PD records of EREC doubles
    int flags[4];                                   // [0] ready, [1] done
};
static_assert(sizeof(MxLds) % 16 == 0 && NT * sizeof(MxLds) <= 160 * 1024, "LDS budget");

// the counters are read and written as LDS words (a volatile access through a generic pointer becomes a system-scope FLAT access)
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int lds_load_flag(const int *p) { return *(const volatile lds_int *)p; }
__device__ __forceinline__ void lds_store_flag(int *p, int v) { asm volatile("" ::: "memory"); *(volatile lds_int *)p = v; asm volatile("" ::: "memory"); }

// a group's step that the chain symmetrises itself (tau: position in the group, ascending time; the chain walks tau = PD-1 .. 0)
__host__ __device__ constexpr bool sym_tau(int tau) { return tau % SYM_EVERY == 0; }

template <bool FXTV, bool CTV, bool REG2>
__global__ __launch_bounds__(DDP_WAVE * 2 * NT) void back_pass_mx2_kernel(BPXArgs a)
{
    const int wave = threadIdx.x / DDP_WAVE, lane = threadIdx.x % DDP_WAVE, l15 = lane & 15, l4 = lane >> 4;
    const int tr = wave % NT, b = blockIdx.x * NT + tr;
    const bool writer = wave >= NT;
    __shared__ __attribute__((aligned(16))) MxLds sm[NT];
    if (threadIdx.x < NT * 4) sm[threadIdx.x / 4].flags[threadIdx.x % 4] = 0;     // before any wave of the work-group looks at them
    __syncthreads();
    if (b >= a.B) return;
    if (a.active && a.active[b] == 0) return;
    const int N = a.N;
    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    MxLds &L = sm[tr];
    double *lds = L.tile, *lout = L.lout;

    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + mm * N * b,
           *Vxg = a.Vx + (size_t)n * N * b, *Vxxg = a.Vxx + nn * N * b;
    const size_t tl = (size_t)(N - 1);
    const int i0 = N - 2;
    const int NG = N >= 2 ? (i0 + 1) / PD : 0;               // whole groups of PD steps (fewer are walked if the pass diverges)
    if (N < 2 || NG == 0) { if (writer) return; }
    // [cx; cu] of the PD steps starting at step t0 -> PD records of EREC doubles at dst (48 lanes x 16 bytes)
    const int et = lane < 48 ? lane / 6 : 0, ew = lane % 6;
    const unsigned estep = ew < 5 ? n * 8u : m * 8u;
    auto dma_e = [&](double *dst, long t0) {
        const char *pE = ew < 5 ? (const char *)(cx + (long)n * (t0 + et) + 2 * ew) : (const char *)(cu + (long)m * (t0 + et));
        if (lane < 48) __builtin_amdgcn_global_load_lds((glb_void *)pE, (lds_void *)dst, 16, 0, 0);
    };

    if (writer) {
        // ================================================ the writer ====================================================
        // Vxx of one step = 50 lanes x 16 bytes; Vx, K, k, Quu of three steps = 54 lanes x 16 bytes
        const int msub = lane / 18, mw = lane % 18;
        const int mL = REC * msub + (mw < 5 ? R_VX + 2 * mw : (mw < 15 ? R_K + 2 * (mw - 5) : (mw == 15 ? R_KV : R_QUU + 2 * (mw - 16))));
        const unsigned mstep = mw < 5 ? n * 8u : (mw < 15 ? (unsigned)(nm * 8) : (mw == 15 ? m * 8u : (unsigned)(mm * 8)));     // bytes per time step
        const long t0 = (long)N - 2 - (PD - 1);                  // lowest step of the first group
        char *pV = (char *)(Vxxg + (long)nn * t0 + 2 * (lane < 50 ? lane : 0));
        char *mb = mw < 5 ? (char *)(Vxg + 2 * mw) : (mw < 15 ? (char *)(Kg + 2 * (mw - 5)) : (mw == 15 ? (char *)kg : (char *)(Quug + 2 * (mw - 16))));
        char *pM = mb + (long)mstep * (t0 + (lane < 54 ? msub : 0));
        // the transposed partners of this lane's pair (i, j), (i+1, j) of Vxx: (j, i), (j, i+1)
        const int e0 = 2 * (lane < 50 ? lane : 0), pi = e0 % n, pj = e0 / n, tp0 = pj + n * pi, tp1 = pj + n * (pi + 1);
        int g = 0;
        for (;;) {
            const int r = lds_load_flag(&L.flags[0]);
            const int ng = r & (FIN - 1);
            if (g >= ng) {
                if (r & FIN) break;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
            asm volatile("" ::: "memory");
            if (g + 2 < NG) dma_e(L.leb[(g + 2) % NE], t0 - (long)PD * (g + 2));      // [cx;cu] of the group after next
            const double *rec = lout + (g % NB) * LOUT;
            if (lane < 50) {
                static_for<0, PD>([&](auto tc) __attribute__((always_inline)) {
                    constexpr int t = decltype(tc)::value;
                    d2 v = *(const d2 *)(rec + REC * t + 2 * lane);
                    if (!sym_tau(t)) { v.x = 0.5 * (v.x + rec[REC * t + tp0]); v.y = 0.5 * (v.y + rec[REC * t + tp1]); }     // :71-72
                    *(d2 *)(pV + nn * 8 * t) = v;
                });
            }
            if (lane < 54) {
                *(d2 *)pM = *(const d2 *)(rec + mL);
                *(d2 *)(pM + 3 * (size_t)mstep) = *(const d2 *)(rec + mL + 3 * REC);
                if (lane < 36) *(d2 *)(pM + 6 * (size_t)mstep) = *(const d2 *)(rec + mL + 6 * REC);
            }
            pV -= nn * 8 * PD; pM -= (size_t)mstep * PD;
            ++g;
            __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0) lgkmcnt(0): the image has landed, the records are read
            lds_store_flag(&L.flags[1], g);
        }
        __builtin_amdgcn_s_waitcnt(0x0070);
        lds_store_flag(&L.flags[1], DONE_ALL);
        return;
    }

    // ==================================================== the chain =====================================================
    __builtin_amdgcn_s_setprio(3);
    const double *fx = a.fx + (a.fx_batched ? nn * (FXTV ? N : 1) * b : 0);
    const double *fu = a.fu + (a.fx_batched ? nm * (FXTV ? N : 1) * b : 0);
    const double *cxx = a.cxx + (a.cost_batched ? nn * (CTV ? N : 1) * b : 0);
    const double *cxu = a.cxu + (a.cost_batched ? nm * (CTV ? N : 1) * b : 0);
    const double *cuu = a.cuu + (a.cost_batched ? mm * (CTV ? N : 1) * b : 0);
    const double lam = a.lambda[b];

    // ---- terminal step (backward_pass.jl:234-236 / :197-199)
    for (int e = lane; e < n * n; e += DDP_WAVE) Vxxg[nn * tl + e] = cxx[(CTV ? nn * tl : 0) + e];
    if (lane < n) Vxg[(size_t)n * tl + lane] = cx[(size_t)n * tl + lane];
    if (lane < 4) Quug[mm * tl + lane] = cuu[(CTV ? mm * tl : 0) + lane];
    if (lane < 2 * n) Kg[nm * tl + lane] = 0.0;
    if (lane < m) kg[(size_t)m * tl + lane] = 0.0;
    if (N < 2) {
        if (lane == 0) { a.dV[2 * b] = 0.0; a.dV[2 * b + 1] = 0.0; a.diverge[b] = 0; }
        return;
    }
    for (int e = lane; e < TLD * 16 + 16; e += DDP_WAVE) lds[e] = 0.0;

    // ---- per-lane operand streams (as in back_pass_mx.hip) -------------------------------------------------------
    auto h_stream = [&](int row, int col) -> Stream {       // H = [cxx cxu; cxu' cuu] (p x p), zero outside
        if (col < p && row < p) {
            if (row < n && col < n) return Stream{(const char *)(cxx + row + n * col), CTV ? (unsigned)(nn * 8) : 0u, nullptr};
            if (row < n) return Stream{(const char *)(cxu + row + n * (col - n)), CTV ? (unsigned)(nm * 8) : 0u, nullptr};
            if (col < n) return Stream{(const char *)(cxu + col + n * (row - n)), CTV ? (unsigned)(nm * 8) : 0u, nullptr};
            return Stream{(const char *)(cuu + (row - n) + m * (col - n)), CTV ? (unsigned)(mm * 8) : 0u, nullptr};
        }
        return Stream{(const char *)mx_zero, 0u, nullptr};
    };
    auto f_stream = [&](int row, int col) -> Stream {       // F = [fx fu] (n x p), zero outside
        if (row < n && col < n) return Stream{(const char *)(fx + row + n * col), FXTV ? (unsigned)(nn * 8) : 0u, nullptr};
        if (row < n && col < p) return Stream{(const char *)(fu + row + n * (col - n)), FXTV ? (unsigned)(nm * 8) : 0u, nullptr};
        return Stream{(const char *)mx_zero, 0u, nullptr};
    };
    const int urow = n + (l4 & 1);
    Stream hS[4], fS[3];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = s < 3 ? l4 + 4 * s : urow;
        hS[s] = l15 == VC ? Stream{(const char *)mx_zero, 0u, nullptr} : h_stream(row, l15);
        if (s < 3) fS[s] = f_stream(l4 + 4 * s, l15 < p ? l15 : n + (l15 & 1));
    }
    const int eidx = l15 + l4 < p ? l15 + l4 : p - 1;
    Stream eS = eidx < n ? Stream{(const char *)(cx + eidx), (unsigned)(n * 8), nullptr}
                         : Stream{(const char *)(cu + (eidx - n)), (unsigned)(m * 8), nullptr};

    // ---- loop-invariant lane constants ---------------------------------------------------------------------------
    const double mask12 = l15 == VC ? 1.0 : 0.0;
    const double cB = l4 >= 2 ? 1.0 : -lam;                   // regType 1: T = -λK in the 16-lane rows 0, 1
    const bool odd = (l4 & 1) != 0, hi2 = l4 >= 2;
    const int wr = l4 + TLD * l15;
    const int rdT = l15 == VC ? TZERO : l15 + TLD * l4;
    const int rdS = l15 == VC ? 0 : 4 * TLD;
    const bool v_act01 = l15 < n || l15 == VC, v_act2 = v_act01 && l4 < 2;
    const double vscl = l15 == VC ? 1.0 : 0.5;
    char *vst = l15 == VC ? (char *)(Vxg + (size_t)n * (tl - 1) + l4) : (char *)(Vxxg + nn * (tl - 1) + l4 + n * (l15 < n ? l15 : 0));
    const unsigned vst_stride = l15 == VC ? (unsigned)(n * 8) : (unsigned)(nn * 8);
    const bool quu_lane = hi2 && (l15 == n || l15 == n + 1);
    const bool kq_act = hi2 && l15 <= VC;
    const unsigned long long lanes01 = __builtin_amdgcn_ballot_w64(v_act01), lanes2k = __builtin_amdgcn_ballot_w64(hi2 ? kq_act : v_act2);
    const int a2 = hi2 ? l4 - 2 : 0;
    char *kq = !hi2 ? vst + 64
                    : (l15 < n ? (char *)(Kg + nm * (tl - 1) + a2 + m * l15)
                               : (l15 == VC ? (char *)(kg + (size_t)m * (tl - 1) + a2) : (char *)(Quug + mm * (tl - 1) + a2 + m * (l15 < p ? l15 - n : 0))));
    const unsigned kq_stride = !hi2 ? vst_stride : (l15 < n ? (unsigned)(nm * 8) : (l15 == VC ? (unsigned)(m * 8) : (unsigned)(mm * 8)));
    // where this lane's results go in a step record; lanes without an output aim behind the records (the buffer offset of a
    // group is added to the real lanes' index only)
    const bool real1 = l15 < n || l15 == VC;
    const int w1 = l15 < n ? l4 + n * l15 : (l15 == VC ? R_VX + l4 : NB * LOUT + lane);
    const bool real2 = !hi2 ? real1 : l15 < p || l15 == VC;
    const int w2 = !hi2 ? w1 + 8
                        : (l15 < n ? R_K + a2 + m * l15 : (l15 == VC ? R_KV + a2 : (l15 < p ? R_QUU + a2 + m * (l15 - n) : NB * LOUT + lane)));

    // ---- register-resident operands -------------------------------------------------------------------------------
    const double hmask = l15 < p ? 0.5 : 0.0, fmask = l15 < p ? 1.0 : 0.0;
    double F[3], Fh[3], Ff[3], Hc[4];                        // F_s (A of GEMM2); B of GEMM1: ½F_s when A carries V + V', F_s when it carries V
    double er[PD], hr[CTV ? PD : 1][4], fr[FXTV ? PD : 1][3];
#pragma unroll
    for (int j = 0; j < PD; ++j) {
        const int t = i0 - j > 0 ? i0 - j : 0;
        er[j] = 0.0;
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
        for (int s = 0; s < 3; ++s) { F[s] = fS[s].at(0); Fh[s] = hmask * F[s]; Ff[s] = fmask * F[s]; }
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
            S[s] = (l15 < n && row < n) ? 2.0 * hs[s].at((int)tl) : ((l15 == VC && row < n) ? cx[(size_t)n * tl + row] : 0.0);
        }
    }
    const d4 zero4 = d4{0.0, 0.0, 0.0, 0.0};
    const long t0 = (long)N - 2 - (PD - 1);
    if (NG > 0) dma_e(L.leb[0], t0);
    if (NG > 1) dma_e(L.leb[1], t0 - PD);
    wave_sync();
    __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0) lgkmcnt(0): set-up loads and the first images have landed, the counters are 0

    double dVa = 0.0, dVp = 0.0;                    // Σ k'Qu  and the per-row halves of Σ k'(Quu k + Qu)
    int diverge = 0;
    const double *ecur = L.leb[0];
    int wb1 = 0, wb2 = 0;                           // this lane's record indices with the group's buffer offset
    // One time step (see back_pass_mx.hip for the tile algebra).  mode 1: a step of a group — [cx;cu] from the LDS image, results
    // into the step record, ½(V + V') on the chain only if sym_tau; mode 2: the steps below the last whole group — operands and
    // results straight from / to global memory, always symmetrised.  SIN: S holds V + V' (else V) of the step before.
    auto step = [&](const int i, auto slot_c, auto mode_c) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value, mode = decltype(mode_c)::value;
        constexpr int tau = PD - 1 - slot;
        constexpr bool SIN = mode == 2 || slot == 0 || sym_tau(tau + 1);      // (slot 0 follows the last step of the group before, tau = 0)
        constexpr bool SOUT = mode == 2 || sym_tau(tau);
        const bool okp = diverge == 0;
        const d4 c = CTV ? d4{hr[slot][0], hr[slot][1], hr[slot][2], hr[slot][3]} : Hc4;
        const double e = mode == 1 ? 0.0 : er[slot];
        // group steps: the entries of [cx;cu] that column VC wants — cu[parity of my 16-lane row] for the u-rows, e[l4 + 4s] for
        // accumulator register s — straight from the LDS image (plain multiply-adds instead of row broadcasts and their wait states)
        const double eu = mode == 1 ? ecur[EREC * tau + n + (l4 & 1)] : 0.0;
        double es[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) es[s] = mode == 1 ? ecur[EREC * tau + l4 + 4 * s] : 0.0;
        if (FXTV) {
#pragma unroll
            for (int s = 0; s < 3; ++s) { F[s] = fr[slot][s]; if (SIN) Fh[s] = hmask * F[s]; else Ff[s] = fmask * F[s]; }
        }
        const double *Bg = SIN ? Fh : Ff;
        // ================= GEMM1: W = Vxx·F; column VC := Vx (F[:,VC] = 0, S[:,VC] = Vx) ============================
        d4 w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[0], Bg[0], zero4, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[1], Bg[1], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[2], Bg[2], w, 0, 0, 0);
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
            fmac_bcast<10, 0x3, true>(Z, e, mask12);
            fmac_bcast<8, 0xc>(Z, e, mask12);
        }
        double Q0, Q1, F00, F01, F11;
        if (REG2) {                                        // u-rows of F'(W + λF) + H: Qux_reg, QuuF
            const double lamB = SIN ? 2.0 * lam : lam;
            d4 gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], fma(lamB, Bg[0], W[0]), c, 0, 0, 0);
            gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], fma(lamB, Bg[1], W[1]), gr, 0, 0, 0);
            gr = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2], fma(lamB, Bg[2], W[2]), gr, 0, 0, 0);
            double Zr;
            if (mode == 1) Zr = fma(eu, mask12, gr.w);
            else {
                Zr = gr.w + 0.0;
                fmac_bcast<10, 0x3, true>(Zr, e, mask12);
                fmac_bcast<8, 0xc>(Zr, e, mask12);
            }
            spread_pair(Zr, Q0, Q1);
            F00 = row_bcast<n>(Q0); F01 = row_bcast<n + 1>(Q0); F11 = row_bcast<n + 1>(Q1);
        } else {
            spread_pair(Z, Q0, Q1);
            F00 = row_bcast<n>(Q0) + lam; F01 = row_bcast<n + 1>(Q0); F11 = row_bcast<n + 1>(Q1) + lam;
        }
        const double det = fma(F00, F11, -(F01 * F01));
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
        const double Ksel = odd ? K1 : K0;
        double Tsel, Bop;
        if (!REG2) {
            Bop = Ksel * cB;                               // T = -λK (rows 0, 1) | K (rows 2, 3)
            Tsel = Bop;
        } else {
            Tsel = Z;
            fmac_bcast<n, 0xf, true>(Tsel, Z, K0);
            fmac_bcast<n + 1>(Tsel, Z, K1);
            Bop = hi2 ? Ksel : Tsel;
        }
        // ================= value update (:69-72): V = G + [K' Qux']·[T; K] ===========================================
        const double Aop = hi2 ? Z : Ksel;
        const d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop, Bop, g, 0, 0, 0);
        const bool badu = (__builtin_amdgcn_ballot_w64(!(F00 > 0.0)) | __builtin_amdgcn_ballot_w64(!(det > 0.0))) != 0;
        if (__builtin_expect(badu || !okp, 0)) {
            asm volatile("" ::: "memory");
            if (okp) diverge = i + 1;                      // diverge = i (:37-38)
        } else {
            asm volatile("" ::: "memory");
            dVa = fma(K1, Q1, fma(K0, Q0, dVa));
            dVp = fma(Ksel, Tsel, dVp);
        }
        const double kqv = quu_lane ? Z : Ksel;            // K | k | Quu of this step (16-lane rows 2, 3)
        if (SOUT) {
            // ---- ½(V + V') through the transpose tile; registers keep V + V' (column VC: Vx)
            lds[wr] = v.x; lds[wr + 4] = v.y; lds[wr + 8] = v.z;
            wave_sync();
            S[0] = v.x + lds[rdT]; S[1] = v.y + lds[rdT + rdS]; S[2] = v.z + lds[rdT + 2 * rdS];
            if (mode == 1) { S[0] = fma(es[0], mask12, S[0]); S[1] = fma(es[1], mask12, S[1]); S[2] = fma(es[2], mask12, S[2]); }
            else { fmac_bcast<0>(S[0], e, mask12); fmac_bcast<4>(S[1], e, mask12); fmac_bcast<8>(S[2], e, mask12); }
            if (mode == 1) {
                lout[wb1 + REC * tau] = vscl * S[0];
                lout[wb1 + 4 + REC * tau] = vscl * S[1];
                lout[wb2 + REC * tau] = hi2 ? kqv : vscl * S[2];
            } else {
                store2_masked(vst, vscl * S[0], vscl * S[1], lanes01);
                store_masked(kq, hi2 ? kqv : vscl * S[2], lanes2k);
                vst -= vst_stride;
                kq -= kq_stride;
            }
            wave_sync();                                   // the tile is free again
        } else {
            // ---- the record takes V as it is (the writer stores ½(V + V')); registers keep V (column VC: Vx)
            S[0] = fma(es[0], mask12, v.x); S[1] = fma(es[1], mask12, v.y); S[2] = fma(es[2], mask12, v.z);
            lout[wb1 + REC * tau] = S[0];
            lout[wb1 + 4 + REC * tau] = S[1];
            lout[wb2 + REC * tau] = hi2 ? kqv : S[2];
        }
        {   // refill the ring slot with the step PD ahead (clamped: always a valid load)
            asm volatile("" ::: "memory");
            if (mode == 2) er[slot] = eS.next();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (FXTV && s < 3) fr[slot][s] = fS[s].next();
                if (CTV) hr[slot][s] = hS[s].next();
            }
            if (i - PD > 0) {
                if (mode == 2) eS.back();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (FXTV && s < 3) fS[s].back();
                    if (CTV) hS[s].back();
                }
            }
        }
    };
    int i = i0, g = 0, seen = 0;
    while (i >= PD - 1 && diverge == 0) {
        // the buffer of this group was last used by group g - 2, and [cx;cu] of this group was fetched while the writer worked
        // on group g - 2: both are certain once `done` >= g - 1 (read half a group ago; the writer needs ~a tenth of a group)
        while (__builtin_expect(seen < g - 1, 0)) seen = lds_load_flag(&L.flags[1]);
        ecur = L.leb[g % NE];
        const int boff = (g % NB) * LOUT;
        wb1 = real1 ? w1 + boff : w1;
        wb2 = real2 ? w2 + boff : w2;
        static_for<0, PD>([&](auto sc) __attribute__((always_inline)) {
            step(i - decltype(sc)::value, sc, IC<1>{});
            if constexpr (decltype(sc)::value == PD / 2) seen = lds_load_flag(&L.flags[1]);
        });
        ++g;
        lds_store_flag(&L.flags[0], g);
        i -= PD;
    }
    lds_store_flag(&L.flags[0], g | FIN);
    // the steps below the last whole group go the direct way
    vst -= (size_t)vst_stride * (unsigned)(i0 - i); kq -= (size_t)kq_stride * (unsigned)(i0 - i);
#pragma unroll
    for (int j = 0; j < PD - 1; ++j) er[j] = eS.at(i - j > 0 ? i - j : 0);
    static_for<0, PD - 1>([&](auto sc) __attribute__((always_inline)) {
        if (i >= 0 && diverge == 0) { step(i, sc, IC<2>{}); --i; }
    });

    if (diverge) {   // outputs earlier in time than the failing step are zero (backward_pass.jl:226-229)
        const size_t ie = (size_t)diverge;          // = i + 1
        if (NG > 0) while (lds_load_flag(&L.flags[1]) != DONE_ALL) __builtin_amdgcn_s_sleep(4);   // the writer's copies of the garbage steps have left
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): and so have mine
        for (size_t e = lane; e < nm * ie; e += DDP_WAVE) Kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)m * ie; e += DDP_WAVE) kg[e] = 0.0;
        for (size_t e = lane; e < (size_t)n * ie; e += DDP_WAVE) Vxg[e] = 0.0;
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

template <bool REG2>
int launch_mx2(ddp_handle h, const ddp_bp_desc *d, const BPXArgs &a)
{
    const dim3 grid((d->B + NT - 1) / NT), block(DDP_WAVE * 2 * NT);
    const int key = (d->fx_tv ? 2 : 0) | (d->cost_tv ? 1 : 0);
    switch (key) {
    case 0: hipLaunchKernelGGL((back_pass_mx2_kernel<false, false, REG2>), grid, block, 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((back_pass_mx2_kernel<false, true, REG2>), grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((back_pass_mx2_kernel<true, false, REG2>), grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL((back_pass_mx2_kernel<true, true, REG2>), grid, block, 0, h->stream, a); break;
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// returns 1 if this shape / alignment is not handled here (the caller goes on to back_pass_mx), 0 launched, <0 error
int ddp_launch_back_pass_mx2(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const int32_t *active, double *K,
                             double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge)
{
    if (d->has_lims || d->m != 2 || d->n != 10) return 1;
    // the group write-back and the [cx;cu] image need 16-byte aligned arrays (every per-step size of this shape is a multiple of 16 bytes)
    if ((((uintptr_t)cx | (uintptr_t)cu | (uintptr_t)K | (uintptr_t)k | (uintptr_t)Quu | (uintptr_t)Vx | (uintptr_t)Vxx) & 15) != 0) return 1;
    BPXArgs a;
    a.N = d->N; a.B = d->B; a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    return d->regType == 2 ? launch_mx2<true>(h, d, a) : launch_mx2<false>(h, d, a);
}
