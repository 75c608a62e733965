// back_pass_dpp.hip — backward pass with one 16-lane DPP row per trajectory (4 trajectories per wavefront),
// no LDS on the dependency chain.  Same arithmetic as back_pass.hip / src/backward_pass.jl:162-252 + :28-79.
//
// gfx950 can broadcast one lane of every 16-lane row inside a double-precision FMA
//     v_fmac_f64_dpp acc, src0, src1 row_newbcast:l      acc += src0[lane l of my row] * src1
// at the cost of a plain FMA (profiles/microbench/dpp_fma_bench.hip).  With lane j of a row holding COLUMN j
// of every matrix of its trajectory (registers = rows), both products of the Q-function expansion are
// runs of such instructions:
//     W[r, j] = Σ_l Vxx[r,l]·F[l,j]   ->  w[r] += bcast_l(Vcol[r]) · Fcol[l]      (Vxx[r,l] = lane l, register r)
//     G[i, j] = Σ_k F[k,i]·W[k,j]     ->  g[i] += bcast_i(Fcol[k]) · w[k]
// and so is the value update  Vxx_i = Qxx + ½(S+S'),  (S+S')[i,j] = Σ_a K[a,i]·Y[a,j] + Y[a,i]·K[a,j], Y = T+Qux.
// The m x m system is broadcast inside the row with v_mov_b64_dpp and factorised redundantly by the row's
// lanes (or solved by boxQP when limits are given — wave-uniform per row, rows diverge independently).
// Compared with the one-wave-per-trajectory kernels this issues ~4x fewer instructions per trajectory-step
// (no idle lanes, no LDS traffic, no hand-offs), which is what matters once the batch fills the machine;
// at small batches (<= ~2 trajectories per SIMD) the 64-lane kernels have the shorter dependency chain.
//
// Exact symmetry of the stored Vxx (backward_pass.jl:71-72): the in-register recursion state is symmetric only
// to rounding (lane j forms column j from G[:,j], and G = F'VF is not bitwise symmetric); the stored Vxx_i is
// ½(V + V') taken through an LDS transpose one step later, off the dependency chain.
#include <type_traits>
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

struct BPDArgs {
    int N, B;
    int fx_batched, cost_batched, regType;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
};

typedef double d2 __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ void fmac_bc(double &acc, double src0, double src1)
{   // acc += src0[lane L of this 16-lane row] * src1
    // volatile on purpose: without it the scheduler interleaves the unrolled time steps and spills (256 VGPR + 128 AGPR)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src0), "v"(src1), "n"(L));
}
template <int L>
__device__ __forceinline__ double row_bcast(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, false); }

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
    for (int i = 0; i < NN; ++i) asm volatile("" : "+v"(v[i]));      // all elements materialised ...
    asm volatile("s_nop 1" ::: "memory");                             // ... two wait states before the first DPP read
}

template <int NS, int MS, bool FXTV, bool CTV, bool LIMS>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_dpp_kernel(BPDArgs a)
{
    constexpr int n = NS, m = MS, p = n + m, G = 16, GPW = DDP_WAVE / G, D = 8, LD = n + 2;   // LD even: 16-byte aligned columns
    static_assert(p + 1 <= G, "n + m + 1 lanes must fit one 16-lane DPP row");
    const int N = a.N;
    const int lane = threadIdx.x, grp = lane / G, j = lane % G;
    long tb = (long)blockIdx.x * GPW + grp;
    const bool valid = tb < a.B;
    if (!valid) tb = a.B - 1;                                   // all lanes stay alive (DPP reads every lane)
    const int b = (int)tb;
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;                                    // nothing to do for this wave (trajectories of a batch finish at different iterations)
    const bool inx = j < n, inu = j >= n && j < p, ink = j == p;   // column roles: x-columns, u-columns, spare lane (stores k)
    const int jx = inx ? j : 0, jc = j < p ? j : 0, ja = inu ? j - n : 0;

    __shared__ __attribute__((aligned(16))) double tr[GPW][2][n * LD + 2];   // transpose buffers for the symmetric Vxx output

    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
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
    const bool reg2 = a.regType == 2;
    bool nolims = true;
    double limlo[m], limhi[m];
    if (LIMS) {
        nolims = a.lims[0] > a.lims[m];                             // backward_pass.jl:31
#pragma unroll
        for (int q = 0; q < m; ++q) { limlo[q] = a.lims[q]; limhi[q] = a.lims[q + m]; }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35

    // ---- column j of F = [fx fu], of the cost Hessians, and of the state
    double Fcol[n], cxxcol[n], ccol[m];            // ccol: x-lanes cxu[j, :] (= column j of cxu'), u-lanes cuu[:, j-n]
    // The idle lanes' columns are zeroed by mask_F / mask_C when the values are USED, not when they are loaded: any
    // arithmetic on a loaded value puts the s_waitcnt at the load and would expose the HBM latency of the prefetch ring.
    const double zF = (j < p) ? 1.0 : 0.0, zx = inx ? 1.0 : 0.0, zu = (inx || inu) ? 1.0 : 0.0;
    auto load_F = [&](int i, double (&F)[n]) {
        const double *src = (j < n) ? fx + (FXTV ? nn * i : 0) + (size_t)n * jx : fu + (FXTV ? nm * i : 0) + (size_t)n * ja;
#pragma unroll
        for (int r = 0; r < n; ++r) F[r] = src[r];
    };
    auto mask_F = [&](double (&F)[n], const double (&raw)[n]) {
#pragma unroll
        for (int r = 0; r < n; ++r) F[r] = zF * raw[r];
    };
    auto load_C = [&](int i, double (&cc)[n], double (&c2)[m]) {
#pragma unroll
        for (int r = 0; r < n; ++r) cc[r] = cxx[(CTV ? nn * i : 0) + (size_t)n * jx + r];
        const double *src = inx ? cxu + (CTV ? nm * i : 0) + jx : cuu + (CTV ? mm * i : 0) + (size_t)m * ja;     // stride n | 1
#pragma unroll
        for (int q = 0; q < m; ++q) c2[q] = src[inx ? (size_t)n * q : (size_t)q];
    };
    auto mask_C = [&](double (&cc)[n], double (&c2)[m], const double (&rawc)[n], const double (&raw2)[m]) {
#pragma unroll
        for (int r = 0; r < n; ++r) cc[r] = zx * rawc[r];
#pragma unroll
        for (int q = 0; q < m; ++q) c2[q] = zu * raw2[q];
    };
    double Vcol[n], vj;
    // terminal step (backward_pass.jl:234-236 / :197-199)
    {
        const size_t tl = (size_t)(N - 1);
#pragma unroll
        for (int r = 0; r < n; ++r) Vcol[r] = inx ? cxx[(CTV ? nn * tl : 0) + (size_t)n * jx + r] : 0.0;
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
                for (int q = 0; q < m; ++q) Quug[mm * tl + (size_t)m * ja + q] = cuu[(CTV ? mm * tl : 0) + q + (size_t)m * ja];
            }
            if (ink) {
#pragma unroll
                for (int q = 0; q < m; ++q) kg[(size_t)m * tl + q] = 0.0;
            }
        }
    }
    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;
    if (N >= 2) {
        load_F(FXTV ? N - 2 : 0, Fcol);
        mask_F(Fcol, Fcol);
        load_C(CTV ? N - 2 : 0, cxxcol, ccol);
        mask_C(cxxcol, ccol, cxxcol, ccol);
        // regType 2 adds λ·F_u'F to the u-rows (backward_pass.jl:245-247): FuF[a] = Σ_k F[k,n+a]·F[k,j]
        double FuF[m];
        auto make_FuF = [&]() {
#pragma unroll
            for (int q = 0; q < m; ++q) FuF[q] = 0.0;
            if (reg2) {
                static_for<0, n>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    static_for<0, m>([&](auto qc) { constexpr int q = decltype(qc)::value; fmac_bc<n + q>(FuF[q], Fcol[k], Fcol[k]); });
                });
            }
        };
        dpp_fence(Fcol);
        make_FuF();
        double kprev[m];                              // k[:, i+1]: boxQP warm start (backward_pass.jl:49)
#pragma unroll
        for (int q = 0; q < m; ++q) kprev[q] = 0.0;

        // prefetch ring for the per-step streams: lane j < n: cx[j,i]; u-lanes: cu[j-n,i] (and u[j-n,i] with limits)
        double rc[D], ru[LIMS ? D : 1];
        auto fetch_c = [&](int i, int d) {
            rc[d] = inx ? cx[(size_t)n * i + jx] : cu[(size_t)m * i + ja];
            if (LIMS) ru[d] = ug[(size_t)m * i + ja];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) { const int i = N - 2 - d; fetch_c(i >= 0 ? i : 0, d); }
        // Time-varying operands: a ring DF steps deep.  The loads are UNCONDITIONAL (clamped index) and sit behind the stores of
        // the step: a load inside a branch makes the compiler drain the in-order vmcnt at the join (s_waitcnt vmcnt(0) — the
        // step's own output stores included, ~1 us), and a ring one step deep exposes the HBM latency on every step.
        constexpr int DF = (n <= 6) ? 4 : 2;
        static_assert(D % DF == 0, "ring slots must be fixed registers of the unrolled loop");
        double Fr[FXTV ? DF : 1][FXTV ? n : 1], cxxr[CTV ? DF : 1][CTV ? n : 1], ccr[CTV ? DF : 1][CTV ? m : 1];
        // slot of the operands of step (N-2) - e  is  e % DF
        if constexpr (FXTV) {
#pragma unroll
            for (int e = 1; e <= DF; ++e) load_F(N - 2 - e >= 0 ? N - 2 - e : 0, Fr[e % DF]);
        }
        if constexpr (CTV) {
#pragma unroll
            for (int e = 1; e <= DF; ++e) load_C(N - 2 - e >= 0 ? N - 2 - e : 0, cxxr[e % DF], ccr[e % DF]);
        }
        bool have_prev = false;                       // a column of the previous step waits in the transpose buffer
        int prev_i = 0;
        // Symmetric Vxx_i = ½(V + V') from the transpose buffer, written as 16-byte pieces that are CONTIGUOUS
        // across the lanes of a row (piece q = rows 2(q%PC)..+1 of column q/PC): every store instruction covers
        // whole 128-byte lines (column-per-lane stores would touch 16 B of every 80 B and multiply L2 requests).
        constexpr int PC = n / 2, NPC = n * PC, NS4 = (NPC + G - 1) / G;      // pieces per column / total / per lane
        static_assert(n % 2 == 0, "even n only");
        int oA[NS4], oB[NS4];
#pragma unroll
        for (int s4 = 0; s4 < NS4; ++s4) {
            const int q = j + G * s4, c = (q < NPC ? q : 0) / PC, r = 2 * ((q < NPC ? q : 0) % PC);
            oA[s4] = c * LD + r;                      // V[r..r+1, c]
            oB[s4] = r * LD + c;                      // V[c, r], V[c, r+1] at +LD
        }
        auto store_sym = [&](const double *tb, int istep) __attribute__((always_inline)) {
#pragma unroll
            for (int s4 = 0; s4 < NS4; ++s4) {
                const int q = j + G * s4;
                if (q < NPC) {
                    const d2 va = *(const d2 *)(tb + oA[s4]);
                    const double b0 = tb[oB[s4]], b1 = tb[oB[s4] + LD];
                    *(d2 *)(Vxxg + nn * istep + 2 * (size_t)q) = d2{0.5 * (va.x + b0), 0.5 * (va.y + b1)};
                }
            }
        };

        auto step = [&](int i, int d) __attribute__((always_inline)) {
            // ================= P1: w = Vxx·F[:,j],  q = c + F[:,j]'Vx ==================================
            double w[n], qj = 0.0;
#pragma unroll
            for (int r = 0; r < n; ++r) w[r] = 0.0;
            static_for<0, n>([&](auto lc) {
                constexpr int l = decltype(lc)::value;
                static_for<0, n>([&](auto rcx) { constexpr int r = decltype(rcx)::value; fmac_bc<l>(w[r], Vcol[r], Fcol[l]); });
                fmac_bc<l>(qj, vj, Fcol[l]);
            });
            qj += rc[d];                                                 // Qx (x-lanes) / Qu (u-lanes)  (:240-241)
            // ================= P2: g = F'·w  (column j of G = F'VxxF) ===================================
            double g[p];
#pragma unroll
            for (int r = 0; r < p; ++r) g[r] = 0.0;
            dpp_fence(w);
            static_for<0, n>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                static_for<0, p>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(g[ii], Fcol[k], w[k]); });
            });
#pragma unroll
            for (int r = 0; r < n; ++r) g[r] += cxxcol[r];               // Qxx[:, j]           (:244)
            double gu[m], gr[m];                                         // x-lanes: Qux[:, j]; u-lanes: Quu[:, j-n]
#pragma unroll
            for (int q = 0; q < m; ++q) {
                gu[q] = g[n + q] + ccol[q];                              // (:242-243)
                gr[q] = gu[q] + (reg2 ? lam * FuF[q] : ((j == n + q) ? lam : 0.0));     // Qux_reg / QuuF (:246-247)
            }
            // ================= P3: gains ==================================================================
            double Quu[m * m], H[m * m], R[m * m], Qu[m], kk[m];
            static_for<0, m>([&](auto bc) {
                constexpr int bb = decltype(bc)::value;
                Qu[bb] = row_bcast<n + bb>(qj);
                static_for<0, m>([&](auto ac) {
                    constexpr int aa = decltype(ac)::value;
                    Quu[aa + m * bb] = row_bcast<n + bb>(gu[aa]);
                    H[aa + m * bb] = row_bcast<n + bb>(gr[aa]);
                });
            });
            unsigned clamped = 0u;
            int fail;
            double Kc[m], ri[m], rH1 = 0.0;
            bool use_rh = false;
            if constexpr (!LIMS) {
                // cholesky(Hermitian(QuuF)) (:35) with reciprocal pivots (v_rsq_f64 + 2 Newton steps): the whole
                // factor-and-solve is division-free, which matters because every lane of the row repeats it
                fail = 0;
#pragma unroll
                for (int c = 0; c < m; ++c) {
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
                auto rsolve = [&](double (&bv)[m]) {                     // bv <- -(R'R)\bv
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
                rsolve(kk);                                              // k_i = -(R\Qu)        (:41)
                rsolve(Kc);                                              // K_i[:, j] = -(R\Qux_reg[:, j])  (:42)
            } else if (nolims) {
                fail = chol_masked_ri<m>(m, H, 0u, R, ri);                   // cholesky(Hermitian(QuuF))  (:35)
#pragma unroll
                for (int q = 0; q < m; ++q) kk[q] = Qu[q];
                chol_solve_ri<m>(m, R, ri, kk);
#pragma unroll
                for (int q = 0; q < m; ++q) kk[q] = -kk[q];              // k_i = -(R\Qu)  (:41)
            } else {
                double lo[m], up[m], uq[m];
                static_for<0, m>([&](auto qc) { constexpr int q = decltype(qc)::value; uq[q] = row_bcast<n + q>(ru[LIMS ? d : 0]); });
#pragma unroll
                for (int q = 0; q < m; ++q) { lo[q] = limlo[q] - uq[q]; up[q] = limhi[q] - uq[q]; }   // (:45-46)
                int iters, result;
                if constexpr (m == 1) {                                  // scalar, division-free restatement (boxqp_dev.h)
                    result = boxqp_dev1(H[0], Qu[0], lo[0], up[0], kprev[0], qpo, kk[0], rH1, clamped, iters);
                    use_rh = true;
                } else {
                    if constexpr (m == 2) result = boxqp_dev2(H, Qu, lo, up, kprev, qpo, kk, R, ri, clamped, iters);       // (:49), straight-line
                    else result = boxqp_dev_ri<m>(m, H, Qu, lo, up, kprev, qpo, kk, R, ri, clamped, iters);        // (:49)
                }
                fail = (result < 1);                                     // (:53)
            }
            const bool alive = diverge == 0 && !fail;
            if (diverge == 0 && fail) diverge = i + 1;                   // (:37-38,54-55)
            // K_i[:, j] for the x-lanes: -(R'R)\Qux_reg[:, j], clamped rows zero  (:42 / :57-61).  Lanes that are not
            // x-columns carry don't-care values: they are never a broadcast source and never stored.
            double Y[m];
            if constexpr (LIMS) {
                if (m == 1 && use_rh) {
                    Kc[0] = (clamped & 1u) ? 0.0 : -(gr[0] * rH1);
                } else {
#pragma unroll
                    for (int q = 0; q < m; ++q) Kc[q] = ((clamped >> q) & 1u) ? 0.0 : gr[q];
                    chol_solve_ri<m>(m, R, ri, Kc);
#pragma unroll
                    for (int q = 0; q < m; ++q) Kc[q] = ((clamped >> q) & 1u) ? 0.0 : -Kc[q];
                }
            }
            double Quuk[m];
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
            double P1a[n], P2a[n];                                       // Σ_a K[a,i]Y[a,j]  and  Σ_a Y[a,i]K[a,j]
#pragma unroll
            for (int r = 0; r < n; ++r) { P1a[r] = 0.0; P2a[r] = 0.0; }
            dpp_fence(Kc);
            dpp_fence(Y);
            static_for<0, m>([&](auto ac) {
                constexpr int aa = decltype(ac)::value;
                static_for<0, n>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(P1a[ii], Kc[aa], Y[aa]); });
                static_for<0, n>([&](auto ic) { constexpr int ii = decltype(ic)::value; fmac_bc<ii>(P2a[ii], Y[aa], Kc[aa]); });
            });
            double vx = qj;                                              // Vx_i[j] (:69)
#pragma unroll
            for (int q = 0; q < m; ++q) vx += Kc[q] * (Quuk[q] + Qu[q]) + gu[q] * kk[q];
            // ---- stores of this step (a diverged trajectory keeps writing; its range is zero-filled after the loop)
            if (act) {
                if (inx) {
#pragma unroll
                    for (int q = 0; q < m; ++q) Kg[nm * i + (size_t)m * j + q] = Kc[q];                       // (:76)
                    Vxg[(size_t)n * i + j] = vx;
                }
                if (inu) {
#pragma unroll
                    for (int q = 0; q < m; ++q) Quug[mm * i + (size_t)m * ja + q] = gu[q];
                }
                if (ink) {
#pragma unroll
                    for (int q = 0; q < m; ++q) kg[(size_t)m * i + q] = kk[q];                                 // (:75)
                }
            }
            // ---- symmetric Vxx output of the PREVIOUS step from its transpose buffer, then queue this step
            double *tb0 = &tr[grp][i & 1][0], *tb1 = &tr[grp][(i + 1) & 1][0];
            if (have_prev && act) store_sym(tb1, prev_i);
            double vnew[n];
#pragma unroll
            for (int r = 0; r < n; ++r) {
                vnew[r] = (g[r] + 0.5 * (P1a[r] + P2a[r]));             // Qxx + ½(S+S')
                if (inx) tb0[j * LD + r] = vnew[r];
            }
            // The recursion continues with the SYMMETRISED value like the reference (:71-72), read back as row j of the buffer.
            // Carrying the column as computed looks harmless (it is symmetric up to rounding) but the antisymmetric rounding
            // residue then obeys δ_i = F_cl'·δ_{i+1}·F_cl and grows geometrically for non-contractive dynamics: 5e-8 after 210
            // steps of ρ(A) ≈ 1.1 in the randomised sweep (tests/fuzz_gpu_parity.py), unbounded for longer horizons.
            wave_sync();
#pragma unroll
            for (int r = 0; r < n; ++r) Vcol[r] = 0.5 * (vnew[r] + tb0[r * LD + jx]);
            vj = vx;
            have_prev = true; prev_i = i;
#pragma unroll
            for (int q = 0; q < m; ++q) kprev[q] = kk[q];
            fetch_c(i - D >= 0 ? i - D : 0, d);
            if constexpr (FXTV) {                                        // operands of step i-1 leave the ring, those of i-1-DF enter
                mask_F(Fcol, Fr[(d + 1) % DF]);
                load_F(i - 1 - DF >= 0 ? i - 1 - DF : 0, Fr[(d + 1) % DF]);
            }
            if constexpr (CTV) {
                mask_C(cxxcol, ccol, cxxr[(d + 1) % DF], ccr[(d + 1) % DF]);
                load_C(i - 1 - DF >= 0 ? i - 1 - DF : 0, cxxr[(d + 1) % DF], ccr[(d + 1) % DF]);
            }
            dpp_fence(Vcol);
            asm volatile("s_nop 1" : "+v"(vj));
            if constexpr (FXTV) { dpp_fence(Fcol); make_FuF(); }
            wave_sync();                                                 // orders the transpose buffer writes/reads
        };
        dpp_fence(Vcol);
        asm volatile("s_nop 1" : "+v"(vj));
        // Whole groups of D steps run without a guard: inside a conditional the waitcnt pass cannot count the memory operations
        // of the other steps as "issued after" a ring slot's load and falls back to s_waitcnt vmcnt(0/1) for every slot.
        int i0 = N - 2;
        for (; i0 - (D - 1) >= 0; i0 -= D) {
#pragma unroll
            for (int d = 0; d < D; ++d) step(i0 - d, d);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (i0 - d >= 0) step(i0 - d, d);
        }
        // flush the last queued column (step 0)
        if (have_prev && act) store_sym(&tr[grp][prev_i & 1][0], prev_i);
        // outputs earlier in time than a failing step are zero (backward_pass.jl:37-38 with :226-229); Quu of the
        // failing step itself stays (it was assigned before the failure), earlier Quu is `undef` upstream, zero here
        if (diverge && act) {
            const size_t ie = (size_t)diverge;                      // = failing 0-based step + 1
            for (size_t e = j; e < nm * ie; e += G) Kg[e] = 0.0;
            for (size_t e = j; e < (size_t)m * ie; e += G) kg[e] = 0.0;
            for (size_t e = j; e < (size_t)n * ie; e += G) Vxg[e] = 0.0;
            for (size_t e = j; e < nn * ie; e += G) Vxxg[e] = 0.0;
            for (size_t e = j; e < mm * (ie - 1); e += G) Quug[e] = 0.0;
        }
    }
    if (act && j == 0) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
}

template <int NS, int MS>
int launch_dpp_bp(ddp_handle h, const ddp_bp_desc *d, const BPDArgs &a)
{
    const int gpw = DDP_WAVE / 16;
    const dim3 grid((unsigned)((d->B + gpw - 1) / gpw)), block(DDP_WAVE);
    const int key = (d->fx_tv ? 4 : 0) | (d->cost_tv ? 2 : 0) | (d->has_lims ? 1 : 0);
#define DDP_BPD_CASE(K_, FX_, C_, L_)                                                                          \
    case K_:                                                                                                   \
        hipLaunchKernelGGL((back_pass_dpp_kernel<NS, MS, FX_, C_, L_>), grid, block, 0, h->stream, a);         \
        break;
    switch (key) {
        DDP_BPD_CASE(0, false, false, false)
        DDP_BPD_CASE(1, false, false, true)
        DDP_BPD_CASE(2, false, true, false)
        DDP_BPD_CASE(3, false, true, true)
        DDP_BPD_CASE(4, true, false, false)
        DDP_BPD_CASE(5, true, false, true)
        DDP_BPD_CASE(6, true, true, false)
        DDP_BPD_CASE(7, true, true, true)
    }
#undef DDP_BPD_CASE
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// returns 1 if this shape has no DPP kernel (caller falls back), 0 launched, <0 error
int ddp_launch_back_pass_dpp(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge)
{
    BPDArgs a;
    a.N = d->N; a.B = d->B; a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched; a.regType = d->regType;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.lims = lims;
    a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    if (d->n == 10 && d->m == 2) return launch_dpp_bp<10, 2>(h, d, a);
    if (d->n == 4 && d->m == 1) return launch_dpp_bp<4, 1>(h, d, a);
    return 1;
}
