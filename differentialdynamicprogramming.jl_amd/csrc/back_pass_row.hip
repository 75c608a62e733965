// back_pass_row.hip — the 16-lane-row backward pass (back_pass_dpp.hip: one DPP row per trajectory, four trajectories per wavefront,
// no LDS on the dependency chain) for a RANGE of shapes instead of the two it is compiled for there.
//
// The reference's back_pass is size-generic (src/backward_pass.jl:162-252: any n, m; + :28-79); before this file every shape but
// (10,2), (4,1) and (64,8) fell to the 64-lanes-per-trajectory kernel of back_pass.hip (LDS-bound, ~0.07 of HBM at the C2 shape).
// `row_newbcast:L` needs the lane as an immediate, so the sizes of the unrolled products must be compile-time constants — but not
// the ACTUAL sizes: the kernel is compiled for padded sizes NP (even) >= n, MP >= m with NP + MP + 1 <= 16 and runs any n <= NP,
// m <= MP inside them:
//   lanes 0..NP-1 hold the x-columns (lanes n..NP-1 and register rows n..NP-1 carry exact zeros: F, cxx, Vxx are zero-masked where they
//   are used, so every product over the padded range adds 0.0), lanes NP..NP+MP-1 the u-columns, lane NP+MP stores k;
//   the m x m system is factorised by the run-time-sized routines of boxqp_dev.h (the ones the general kernel uses), so the padded
//   control dimensions never enter the Cholesky or the boxQP — iteration counts and result codes are those of the actual m;
//   the operands are addressed with run-time strides (0 for time-invariant ones: the kernel re-reads them every step, from the L2;
//   one instantiation serves the LTI, LTV / TI-cost and LTV / TV-cost methods :217, :162, :179 — a variant that loads time-invariant
//   operands once was measured SLOWER, 1.44 vs 1.28 ms at n = 6, m = 2, N = 1000, B = 4096, and dropped), results are stored with the actual n, m.
// Arithmetic per step as back_pass_dpp.hip (same statement order for the actual entries; the padded ones contribute exact zeros).
#include <type_traits>
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

struct BPRArgs {
    int n, m, N, B, regType;
    long fx_t, fx_b, fu_t, fu_b, cxx_t, cxx_b, cxu_t, cxu_b, cuu_t, cuu_b;      // element strides per time step / per trajectory (0: shared)
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
    double *sink;                                   // >= 64 x 8 B that lanes without an output write to: no exec-mask branch around a store
};

template <int L>
__device__ __forceinline__ void fmac_bc(double &acc, double src0, double src1)
{   // acc += src0[lane L of this 16-lane row] * src1   (volatile: see back_pass_dpp.hip)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src0), "v"(src1), "n"(L));
}
template <int L>
__device__ __forceinline__ double row_bcast(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, false); }

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < E) { f(std::integral_constant<int, I>{}); static_for<I + 1, E>(f); }
}

template <int NN>
__device__ __forceinline__ void dpp_fence(double (&v)[NN])
{   // freshly written DPP sources: the hazard (VALU write -> DPP read, 2 wait states) is not tracked into inline asm
#pragma unroll
    for (int i = 0; i < NN; ++i) asm volatile("" : "+v"(v[i]));
    asm volatile("s_nop 1" ::: "memory");
}

template <int NP, int MP, bool LIMS>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_row_kernel(BPRArgs a)
{
    constexpr int PP = NP + MP, G = 16, GPW = DDP_WAVE / G, D = 4, LD = NP + 1;
    static_assert(PP + 1 <= G && NP % 2 == 0, "NP + MP + 1 lanes must fit one 16-lane DPP row");
    const int n = a.n, m = a.m, N = a.N;
    const int lane = threadIdx.x, grp = lane / G, j = lane % G;
    long tb = (long)blockIdx.x * GPW + grp;
    const bool valid = tb < a.B;
    if (!valid) tb = a.B - 1;
    const int b = (int)tb;
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;
    const int ja_raw = j - NP;
    const bool inx = j < n, inu = ja_raw >= 0 && ja_raw < m, ink = j == PP;
    const int jx = inx ? j : 0, ja = inu ? ja_raw : 0;

    __shared__ double tr[GPW][2][NP * LD + 2];

    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    const double *ug = LIMS ? a.u + (size_t)m * N * b : nullptr;
    const double *fx = a.fx + a.fx_b * b, *fu = a.fu + a.fu_b * b;
    const double *cxx = a.cxx + a.cxx_b * b, *cxu = a.cxu + a.cxu_b * b, *cuu = a.cuu + a.cuu_b * b;
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + mm * N * b,
           *Vxg = a.Vx + (size_t)n * N * b, *Vxxg = a.Vxx + nn * N * b;
    const double lam = a.lambda[b];
    const bool reg2 = a.regType == 2, mfull = m == MP;
    bool nolims = true;
    double limlo[MP], limhi[MP];
    if (LIMS) {
        nolims = a.lims[0] > a.lims[m];                             // backward_pass.jl:31
#pragma unroll
        for (int q = 0; q < MP; ++q) { limlo[q] = q < m ? a.lims[q] : -1.0; limhi[q] = q < m ? a.lims[q + m] : 1.0; }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35

    // masks of the padded entries (applied where a loaded value is USED: arithmetic at the load would expose the latency of the ring)
    double mrow[NP], mq[MP];                                         // register row r < n, control row q < m
#pragma unroll
    for (int r = 0; r < NP; ++r) mrow[r] = r < n ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < MP; ++q) mq[q] = q < m ? 1.0 : 0.0;
    const double zF = (inx || inu) ? 1.0 : 0.0, zx = inx ? 1.0 : 0.0;
    double mF[NP], mX[NP], mC[MP];                                   // ... combined with the lane's role
#pragma unroll
    for (int r = 0; r < NP; ++r) { mF[r] = zF * mrow[r]; mX[r] = zx * mrow[r]; }
#pragma unroll
    for (int q = 0; q < MP; ++q) mC[q] = zF * mq[q];
    int rcl[NP], qcl[MP];                                            // clamped row indices: the loads of padded rows re-read a valid one
#pragma unroll
    for (int r = 0; r < NP; ++r) rcl[r] = r < n ? r : n - 1;
#pragma unroll
    for (int q = 0; q < MP; ++q) qcl[q] = q < m ? q : m - 1;

    auto load_F = [&](int i, double (&F)[NP]) {                      // column j of F = [fx fu] at step i
        const double *src = inx ? fx + a.fx_t * i + (size_t)n * jx : fu + a.fu_t * i + (size_t)n * ja;
#pragma unroll
        for (int r = 0; r < NP; ++r) F[r] = src[rcl[r]];
    };
    auto mask_F = [&](double (&F)[NP], const double (&raw)[NP]) {
#pragma unroll
        for (int r = 0; r < NP; ++r) F[r] = mF[r] * raw[r];
    };
    auto load_C = [&](int i, double (&cc)[NP], double (&c2)[MP]) {   // x-lanes: cxx[:, j], cxu[j, :]; u-lanes: cuu[:, j-n]
#pragma unroll
        for (int r = 0; r < NP; ++r) cc[r] = cxx[a.cxx_t * i + (size_t)n * jx + rcl[r]];
        const double *src = inx ? cxu + a.cxu_t * i + jx : cuu + a.cuu_t * i + (size_t)m * ja;
#pragma unroll
        for (int q = 0; q < MP; ++q) c2[q] = src[inx ? (size_t)n * qcl[q] : (size_t)qcl[q]];
    };
    auto mask_C = [&](double (&cc)[NP], double (&c2)[MP], const double (&rawc)[NP], const double (&raw2)[MP]) {
#pragma unroll
        for (int r = 0; r < NP; ++r) cc[r] = mX[r] * rawc[r];
#pragma unroll
        for (int q = 0; q < MP; ++q) c2[q] = mC[q] * raw2[q];
    };
    double Fcol[NP], cxxcol[NP], ccol[MP], Vcol[NP], vj;
    {   // terminal step (backward_pass.jl:234-236 / :197-199)
        const size_t tl = (size_t)(N - 1);
#pragma unroll
        for (int r = 0; r < NP; ++r) Vcol[r] = (inx && r < n) ? cxx[a.cxx_t * tl + (size_t)n * jx + r] : 0.0;
        vj = inx ? cx[(size_t)n * tl + jx] : 0.0;
        if (act) {
            if (inx) {
#pragma unroll
                for (int r = 0; r < NP; ++r) if (r < n) Vxxg[nn * tl + (size_t)n * j + r] = Vcol[r];
                Vxg[(size_t)n * tl + j] = vj;
#pragma unroll
                for (int q = 0; q < MP; ++q) if (q < m) Kg[nm * tl + (size_t)m * j + q] = 0.0;
            }
            if (inu) {
#pragma unroll
                for (int q = 0; q < MP; ++q) if (q < m) Quug[mm * tl + (size_t)m * ja + q] = cuu[a.cuu_t * tl + q + (size_t)m * ja];
            }
            if (ink) {
#pragma unroll
                for (int q = 0; q < MP; ++q) if (q < m) kg[(size_t)m * tl + q] = 0.0;
            }
        }
    }
    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    if (N >= 2) {
        double FuF[MP];
        auto make_FuF = [&]() {                                      // regType 2: λ·F_u'F on the u-rows (:245-247)
#pragma unroll
            for (int q = 0; q < MP; ++q) FuF[q] = 0.0;
            if (reg2) {
                static_for<0, NP>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    static_for<0, MP>([&](auto qc) { constexpr int q = decltype(qc)::value; fmac_bc<NP + q>(FuF[q], Fcol[k], Fcol[k]); });
                });
            }
        };
        double kprev[MP];
#pragma unroll
        for (int q = 0; q < MP; ++q) kprev[q] = 0.0;
        // prefetch rings: per-step gradients (lane j < n: cx[j,i]; u-lanes: cu[j-n,i], u[j-n,i]) D steps ahead; operands DF steps ahead
        double rc[D], ru[LIMS ? D : 1];
        auto fetch_c = [&](int i, int d) {
            rc[d] = inx ? cx[(size_t)n * i + jx] : cu[(size_t)m * i + ja];
            if (LIMS) ru[d] = ug[(size_t)m * i + ja];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) { const int i = N - 2 - d; fetch_c(i >= 0 ? i : 0, d); }
        const double zc = (inx || inu) ? 1.0 : 0.0;                  // the gradient of the spare lanes is not a gradient
        constexpr int DF = NP >= 10 ? 1 : 2;                             // (a step of the larger sizes is longer than a memory latency; registers)
        static_assert(D % DF == 0, "ring slots must be fixed registers of the unrolled loop");
        double Fr[DF][NP], cxxr[DF][NP], ccr[DF][MP];
        load_F(N - 2, Fcol); mask_F(Fcol, Fcol);
        load_C(N - 2, cxxcol, ccol); mask_C(cxxcol, ccol, cxxcol, ccol);
#pragma unroll
        for (int e = 1; e <= DF; ++e) { const int i = N - 2 - e >= 0 ? N - 2 - e : 0; load_F(i, Fr[e % DF]); load_C(i, cxxr[e % DF], ccr[e % DF]); }
        dpp_fence(Fcol);
        make_FuF();
        bool have_prev = false;
        int prev_i = 0;
        // ½(V + V') from the transpose buffer, element e = j + 16 s of the n x n result per lane: whole 128-byte runs per row
        constexpr int NE = (NP * NP + G - 1) / G;
        // (elements past the end of the n x n result repeat its last one: a duplicate store of the same value instead of a masked store)
        constexpr int SURE = NP > 4 ? ((NP - 1) * (NP - 1)) / G : 0;     // s < SURE: every lane's element exists whatever n in (NP - 2, NP] is
        int oA[NE], oB[NE], oE[NE];
#pragma unroll
        for (int s4 = 0; s4 < NE; ++s4) {
            const int e = j + G * s4, ee = e < (int)nn ? e : (int)nn - 1, c = ee / n, r = ee % n;
            oA[s4] = c * LD + r; oB[s4] = r * LD + c; oE[s4] = 8 * (ee - j);
        }
        // Stores without exec-mask branches (a save-exec / branch / restore costs a lone wave ~25 cycles, and there were ~20 per step): every
        // lane has ONE role — x-lane: K_i[:, j] and Vx_i[j]; u-lane: Quu_i[:, j - n]; the spare lane: k_i — and a running byte pointer for it
        // that steps back one time step per step; lanes without a role (and rows of switched-off trajectories) point at the sink with stride 0.
        char *const sinkp = (char *)a.sink + 8 * lane;
        const bool role = act && (inx || inu || ink);
        unsigned long long pS = (unsigned long long)(role ? (inx ? (char *)(Kg + nm * (size_t)(N - 2) + (size_t)m * j) : inu ? (char *)(Quug + mm * (size_t)(N - 2) + (size_t)m * ja) : (char *)(kg + (size_t)m * (N - 2))) : sinkp);
        const unsigned long long sS = role ? 8ull * (inx ? nm : inu ? mm : (size_t)m) : 0ull;
        unsigned long long pX = (unsigned long long)((act && inx) ? (char *)(Vxg + (size_t)n * (N - 2) + j) : sinkp);
        const unsigned long long sX = (act && inx) ? 8ull * n : 0ull;
        // ½(V + V'): element e = j + 16 s of the step at pV + 128 s bytes (pV points at element j)
        unsigned long long pV = (unsigned long long)(act ? (char *)(Vxxg + nn * (size_t)(N - 2) + j) : sinkp);
        const unsigned long long sV = act ? 8ull * nn : 0ull;
        typedef __attribute__((address_space(1))) double gdbl;
        auto store_sym = [&](const double *tbuf) __attribute__((always_inline)) {      // the step pV points at
#pragma unroll
            for (int s4 = 0; s4 < NE; ++s4) {
                const double v = 0.5 * (tbuf[oA[s4]] + tbuf[oB[s4]]);
                if (s4 < SURE) *(gdbl *)(pV + 128 * s4) = v;
                else *(gdbl *)(pV + (unsigned long long)(long)oE[s4]) = v;
            }
            pV -= sV;
        };
        auto step = [&](int i, int d) __attribute__((always_inline)) {
            // ================= P1: w = Vxx·F[:,j],  q = c + F[:,j]'Vx ==================================
            double w[NP], qj = 0.0;
#pragma unroll
            for (int r = 0; r < NP; ++r) w[r] = 0.0;
            static_for<0, NP>([&](auto lc) {
                constexpr int l = decltype(lc)::value;
                static_for<0, NP>([&](auto rcx) { constexpr int r = decltype(rcx)::value; fmac_bc<l>(w[r], Vcol[r], Fcol[l]); });
                fmac_bc<l>(qj, vj, Fcol[l]);
            });
            qj += zc * rc[d];                                            // Qx (x-lanes) / Qu (u-lanes)  (:240-241)
            // ================= P2: g = F'·w  (column j of G = F'VxxF) ===================================
            double g[PP];
#pragma unroll
            for (int r = 0; r < NP; ++r) g[r] = cxxcol[r];               // the sums start at the cost terms of Qxx / Qux | Quu (:242-244): no zero fill, no add behind
#pragma unroll
            for (int q = 0; q < MP; ++q) g[NP + q] = ccol[q];
            dpp_fence(w);
            static_for<0, NP>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                static_for<0, PP>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(g[ii], Fcol[k], w[k]); });
            });
            double gu[MP], gr[MP];                                       // x-lanes: Qux[:, j]; u-lanes: Quu[:, j-n]
#pragma unroll
            for (int q = 0; q < MP; ++q) {
                gu[q] = g[NP + q];                                       // (:242-243)
                gr[q] = gu[q] + (reg2 ? lam * FuF[q] : ((j == NP + q) ? lam : 0.0));     // Qux_reg / QuuF (:246-247)
            }
            // ================= P3: gains ==================================================================
            double Quu[MP * MP], H[MP * MP], R[MP * MP], Qu[MP], kk[MP];
            static_for<0, MP>([&](auto bc) {
                constexpr int bb = decltype(bc)::value;
                Qu[bb] = row_bcast<NP + bb>(qj);
                static_for<0, MP>([&](auto ac) {
                    constexpr int aa = decltype(ac)::value;
                    Quu[aa + MP * bb] = row_bcast<NP + bb>(gu[aa]);
                    H[aa + MP * bb] = row_bcast<NP + bb>(gr[aa]);
                });
            });
            unsigned clamped = 0u;
            int fail;
            double Kc[MP], ri[MP], rH1 = 0.0;
            bool use_rh = false;
#pragma unroll
            for (int q = 0; q < MP; ++q) { ri[q] = 0.0; kk[q] = Qu[q]; }
#pragma unroll
            for (int q = 0; q < MP * MP; ++q) R[q] = 0.0;
            if (!LIMS || nolims) {
                // (m == MP, the usual case, as a literal: the run-time-sized routines then fold their `i < m` tests away)
                if (mfull) { fail = chol_masked_ri<MP>(MP, H, 0u, R, ri); chol_solve_ri<MP>(MP, R, ri, kk); }   // cholesky(Hermitian(QuuF))  (:35)
                else { fail = chol_masked_ri<MP>(m, H, 0u, R, ri); chol_solve_ri<MP>(m, R, ri, kk); }
#pragma unroll
                for (int q = 0; q < MP; ++q) kk[q] = -kk[q];             // k_i = -(R\Qu)  (:41)
            } else {
                double lo[MP], up[MP], uq[MP];
                static_for<0, MP>([&](auto qc) { constexpr int q = decltype(qc)::value; uq[q] = row_bcast<NP + q>(ru[LIMS ? d : 0]); });
#pragma unroll
                for (int q = 0; q < MP; ++q) { lo[q] = limlo[q] - mq[q] * uq[q]; up[q] = limhi[q] - mq[q] * uq[q]; }   // (:45-46)
                int iters, result;
                if constexpr (MP == 1) {
                    result = boxqp_dev1(H[0], Qu[0], lo[0], up[0], kprev[0], qpo, kk[0], rH1, clamped, iters);
                    use_rh = true;
                } else {
                    if constexpr (MP == 2) {                             // m = 2: straight-line (boxqp_dev2: no loop for the four rows of the wave to diverge in)
                        result = mfull ? boxqp_dev2(H, Qu, lo, up, kprev, qpo, kk, R, ri, clamped, iters)
                                       : boxqp_dev_ri<MP>(m, H, Qu, lo, up, kprev, qpo, kk, R, ri, clamped, iters);
                    } else
                    result = mfull ? boxqp_dev_ri<MP>(MP, H, Qu, lo, up, kprev, qpo, kk, R, ri, clamped, iters)        // (:49)
                                   : boxqp_dev_ri<MP>(m, H, Qu, lo, up, kprev, qpo, kk, R, ri, clamped, iters);
                }
                fail = (result < 1);                                     // (:53)
            }
            const bool alive = diverge == 0 && !fail;
            if (diverge == 0 && fail) diverge = i + 1;                   // (:37-38,54-55)
            // K_i[:, j] = -(R'R)\Qux_reg[:, j], clamped rows zero  (:42 / :57-61)
            if (MP == 1 && use_rh) {
                Kc[0] = (clamped & 1u) ? 0.0 : -(gr[0] * rH1);
            } else {
#pragma unroll
                for (int q = 0; q < MP; ++q) Kc[q] = ((clamped >> q) & 1u) ? 0.0 : gr[q];
                if (mfull) chol_solve_ri<MP>(MP, R, ri, Kc); else chol_solve_ri<MP>(m, R, ri, Kc);
#pragma unroll
                for (int q = 0; q < MP; ++q) Kc[q] = ((clamped >> q) & 1u) ? 0.0 : -Kc[q];
            }
#pragma unroll
            for (int q = 0; q < MP; ++q) { Kc[q] *= mq[q]; kk[q] *= mq[q]; }     // the padded control rows stay exactly zero
            double Y[MP], Quuk[MP];
#pragma unroll
            for (int q = 0; q < MP; ++q) {
                double t = gu[q], s = 0.0;                               // T = Quu·K + Qux, Y = T + Qux
#pragma unroll
                for (int q2 = 0; q2 < MP; ++q2) { t += Quu[q + MP * q2] * Kc[q2]; s += Quu[q + MP * q2] * kk[q2]; }
                Y[q] = t + gu[q];
                Quuk[q] = s;                                             // (:64)
            }
            if (alive) {                                                 // (:68)
#pragma unroll
                for (int q = 0; q < MP; ++q) { dV0 += kk[q] * Qu[q]; dV1 += 0.5 * kk[q] * Quuk[q]; }
            }
            // ================= P4: value update (:69-72) ===================================================
            // ½(K'Y + Y'K) accumulates on g with the halves on Y (:70-72 before the symmetrisation of the stored value)
            double hY[MP];
#pragma unroll
            for (int q = 0; q < MP; ++q) hY[q] = 0.5 * Y[q];
            dpp_fence(Kc);
            dpp_fence(hY);
            static_for<0, MP>([&](auto ac) {
                constexpr int aa = decltype(ac)::value;
                static_for<0, NP>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(g[ii], Kc[aa], hY[aa]); });
                static_for<0, NP>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(g[ii], hY[aa], Kc[aa]); });
            });
            double vx = qj;                                              // Vx_i[j] (:69)
#pragma unroll
            for (int q = 0; q < MP; ++q) vx += Kc[q] * (Quuk[q] + Qu[q]) + gu[q] * kk[q];
#pragma unroll
            for (int q = 0; q < MP; ++q)
                if (q < m) *(gdbl *)(pS + 8 * q) = inx ? Kc[q] : (inu ? gu[q] : kk[q]);                    // (:75-76), Quu_i
            *(gdbl *)pX = vx;
            pS -= sS; pX -= sX;
            double *tb0 = &tr[grp][i & 1][0], *tb1 = &tr[grp][(i + 1) & 1][0];
            if (have_prev) store_sym(tb1);
            double vnew[NP];
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                vnew[r] = g[r];                                          // Qxx + ½(S+S'): a padded x-lane holds exact zeros by itself (its F, cxx, K, Y are zero)
                if (j < NP) tb0[j * LD + r] = vnew[r];
            }
            // the recursion continues with the SYMMETRISED value like the reference (:71-72), read back as row j of the buffer
            wave_sync();
#pragma unroll
            for (int r = 0; r < NP; ++r) Vcol[r] = 0.5 * (vnew[r] + tb0[r * LD + (j < NP ? j : 0)]);
            vj = zx * vx;
            have_prev = true; prev_i = i;
#pragma unroll
            for (int q = 0; q < MP; ++q) kprev[q] = kk[q];
            fetch_c(i - D >= 0 ? i - D : 0, d);
            mask_F(Fcol, Fr[(d + 1) % DF]);
            load_F(i - 1 - DF >= 0 ? i - 1 - DF : 0, Fr[(d + 1) % DF]);
            mask_C(cxxcol, ccol, cxxr[(d + 1) % DF], ccr[(d + 1) % DF]);
            load_C(i - 1 - DF >= 0 ? i - 1 - DF : 0, cxxr[(d + 1) % DF], ccr[(d + 1) % DF]);
            dpp_fence(Vcol);
            asm volatile("s_nop 1" : "+v"(vj));
            dpp_fence(Fcol);
            make_FuF();
            wave_sync();
        };
        dpp_fence(Vcol);
        asm volatile("s_nop 1" : "+v"(vj));
        int i0 = N - 2;
        for (; i0 - (D - 1) >= 0; i0 -= D) {
#pragma unroll
            for (int d = 0; d < D; ++d) step(i0 - d, d);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (i0 - d >= 0) step(i0 - d, d);
        }
        if (have_prev) store_sym(&tr[grp][prev_i & 1][0]);
        if (diverge && act) {                                            // outputs earlier in time than a failing step are zero (:37-38 with :226-229)
            const size_t ie = (size_t)diverge;
            for (size_t e = j; e < nm * ie; e += G) Kg[e] = 0.0;
            for (size_t e = j; e < (size_t)m * ie; e += G) kg[e] = 0.0;
            for (size_t e = j; e < (size_t)n * ie; e += G) Vxg[e] = 0.0;
            for (size_t e = j; e < nn * ie; e += G) Vxxg[e] = 0.0;
            for (size_t e = j; e < mm * (ie - 1); e += G) Quug[e] = 0.0;
        }
    }
    if (act && j == 0) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
}

template <int NP, int MP>
int launch_row(ddp_handle h, const ddp_bp_desc *d, const BPRArgs &a)
{
    const int gpw = DDP_WAVE / 16;
    const dim3 grid((unsigned)((d->B + gpw - 1) / gpw)), block(DDP_WAVE);
    if (d->has_lims) hipLaunchKernelGGL((back_pass_row_kernel<NP, MP, true>), grid, block, 0, h->stream, a);
    else hipLaunchKernelGGL((back_pass_row_kernel<NP, MP, false>), grid, block, 0, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

#ifndef DDP_ROW_PART
#define DDP_ROW_PART 0
#endif

// The padded sizes compiled here: NP in {4, 6, 8, 10, 12, 14}, MP in {1, 2, 4} (3 at NP = 12, 1 at NP = 14: NP + MP + 1 <= 16).
// returns 1 if the shape has no row kernel (n > 14, m > 4, n + m > 15), 0 launched, < 0 error
#if DDP_ROW_PART == 0
int ddp_launch_back_pass_row_hi(ddp_handle h, const ddp_bp_desc *d, const void *args);
int ddp_launch_back_pass_row(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge)
{
    const int n = d->n, m = d->m;
    if (n < 1 || m < 1 || m > 4 || n > 14 || n + m > 15 || (n > 12 && m > 1) || (n > 10 && m > 3)) return 1;
    const long N = d->N;
    BPRArgs a;
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
    a.sink = (double *)h->sink;
    if (!a.sink) return 1;
    const int np = n <= 4 ? 4 : (n + 1) & ~1;
    if (np > 8) return ddp_launch_back_pass_row_hi(h, d, &a);
    const int mp = m <= 2 ? m : 4;
#define ROW_CASE(NP_, MP_) if (np == NP_ && mp == MP_) return launch_row<NP_, MP_>(h, d, a);
    ROW_CASE(4, 1) ROW_CASE(4, 2) ROW_CASE(4, 4)
    ROW_CASE(6, 1) ROW_CASE(6, 2) ROW_CASE(6, 4)
    ROW_CASE(8, 1) ROW_CASE(8, 2) ROW_CASE(8, 4)
#undef ROW_CASE
    return 1;
}
#else
// second translation unit (back_pass_row_hi.hip): the larger padded sizes, compiled beside the first
int ddp_launch_back_pass_row_hi(ddp_handle h, const ddp_bp_desc *d, const void *args)
{
    const BPRArgs &a = *(const BPRArgs *)args;
    const int n = d->n, m = d->m;
    const int np = (n + 1) & ~1;
#define ROW_CASE(NP_, MP_) return launch_row<NP_, MP_>(h, d, a);
    if (np == 10) { if (m == 1) ROW_CASE(10, 1) if (m == 2) ROW_CASE(10, 2) ROW_CASE(10, 4) }
    if (np == 12) { if (m == 1) ROW_CASE(12, 1) if (m == 2) ROW_CASE(12, 2) ROW_CASE(12, 3) }
    if (np == 14) { ROW_CASE(14, 1) }
#undef ROW_CASE
    return 1;
}
#endif
