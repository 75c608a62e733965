// forward_pass_dpp.hip — closed-loop rollout with one 16-lane DPP row per (trajectory, α) rollout.
//
// Same arithmetic as forward_pass.hip / src/forward_pass.jl:9-33; what changes is how the state is
// exchanged.  gfx90a+ (and gfx950) can broadcast one lane of every 16-lane row INSIDE a double-precision
// FMA:   v_fmac_f64_dpp  acc, src0, src1  row_newbcast:l   ==   acc += src0[lane l of my row] * src1
// at the issue cost of a plain v_fmac_f64 (profiles/microbench/dpp_fma_bench.hip: 2.45 vs 2.63 ns per
// wave-instruction).  With lane j of a row holding x̂_j and row j of A, the matrix-vector product
//   x̂⁺_j = Σ_l A[j,l]·x̂_l   is n such instructions — no LDS, no hand-off, no separate broadcast moves.
// The feedback term K·dx is formed the same way from per-lane products K[a,j]·dx_j (so K_i is read with one
// coalesced 16-byte load per lane).  Four rollouts share a wavefront.  The per-step cost is off the
// dependency chain and needs the full x'Qx product, so a separate (time x batch)-parallel kernel evaluates
// it afterwards from xnew/unew (cost_kernel below).
#include <stdlib.h>
#include "ddp_internal.h"

namespace {

struct FDArgs {
    int N, B, nalpha;
    int dyn_tv, dyn_batched, has_policy, has_lims;
    const double *A, *Bm, *K, *k, *x0, *u, *x, *lims;
    const int32_t *active;
    double alpha[16];
    double g, l, h, d;
    double *xnew, *unew;
    // fused cost (ddp_problem::cost_diag): Q, R (their diagonals are used), goal, outputs
    const double *Q, *R;
    double goal[4];
    double *cnew, *csum;
    double *sink;               // >= 64 x 8 B that lanes without an output write to (stores carry no exec-mask branch); NULL: masked stores
    int chunked;                // forward_pend_row_kernel: whole 16-step chunks through LDS (set by its launcher)
    unsigned wrap;              // ddp_problem::diff_wrap: coordinates whose difference x̂ - x is wrapped to [-π, π] (pendulum kernels)
};

typedef double d2 __attribute__((ext_vector_type(2)));

#include "pend_math.h"

// diff_fun for an angle coordinate (src/forward_pass.jl:19 `K*diff_fun(x̂, x)`; ddp_problem::diff_wrap): d - 2π·rint(d / 2π) with 2π in
// two parts — the arithmetic of the run-time-sized kernel (forward_pass.hip: wrap_pi), one v_rndne_f64 + 2 multiply-adds on the chain
__device__ __forceinline__ double wrap_pi2(double d)
{
    const double q = rint(d * 0x1.45f306dc9c883p-3);
    return fma(-q, 0x1.1a62633145c07p-52, fma(-q, 0x1.921fb54442d18p+2, d));
}

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

// acc += src0[lane L of this 16-lane row] * src1
template <int L>
__device__ __forceinline__ void fmac_bc(double &acc, double src0, double src1)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src0), "v"(src1), "n"(L));
}
// a VGPR written by a VALU instruction needs 2 wait states before a DPP instruction reads it; the hazard
// recogniser does not look inside inline asm, so freshly produced DPP sources pass through this fence
__device__ __forceinline__ void dpp_fence(double &v) { asm volatile("s_nop 1" : "+v"(v)); }
__device__ __forceinline__ void dpp_fence(double &a, double &b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void dpp_fence(double &a, double &b, double &c) { asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c)); }

template <int L>
__device__ __forceinline__ double row_bcast(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, false); }

// s += Σ_{l<NN} src[lane l] * w[l]   (two interleaved accumulators halve the dependent chain)
template <int NN, int L = 0>
struct RowDot {
    static __device__ __forceinline__ void run(double &s0, double &s1, double src, const double (&w)[NN])
    {
        if constexpr (L < NN) {
            if constexpr (L % 2 == 0) fmac_bc<L>(s0, src, w[L]); else fmac_bc<L>(s1, src, w[L]);
            RowDot<NN, L + 1>::run(s0, s1, src, w);
        }
    }
};
// s += Σ_{l<NN} src[lane l]
template <int NN, int L = 0>
struct RowSum {
    static __device__ __forceinline__ void run(double &s0, double &s1, double src, double one)
    {
        if constexpr (L < NN) {
            if constexpr (L % 2 == 0) fmac_bc<L>(s0, src, one); else fmac_bc<L>(s1, src, one);
            RowSum<NN, L + 1>::run(s0, s1, src, one);
        }
    }
};

// FUSE (ddp_problem::cost_diag, Q and R diagonal): the per-step cost is evaluated HERE from the values the lanes already hold —
// lane j < n contributes ½Q[j,j]·(x̂_j - goal_j)², lane n+q contributes ½R[q,q]·u_q² — instead of by a second kernel that re-reads
// xnew, unew from HBM (98 MB of the 431 MB a C2 pass moved).  The contributions of 16 steps wait in an LDS tile; then lane t of
// the row sums step t, stores cnew[t] (one 128-byte line per row) and keeps its part of sum(cnew): off the dependency chain,
// ~4 instructions per step.
// FAST: time-invariant dynamics and a sink for the lanes without an output — the step then has no branch at all (a taken branch costs a
// lone wave ~28 cycles, the save-exec / branch / restore around a masked store ~25: profiles/microbench/branch_cost.hip)
template <int KIND, int NS, int MS, bool POLICY, bool LIMS, bool FUSE, bool FAST = false>
__global__ __launch_bounds__(DDP_WAVE) void forward_dpp_kernel(FDArgs a)
{
    constexpr int n = NS, m = MS, G = 16, GPW = DDP_WAVE / G, TS = 17;             // TS: padded row of the cost tile (no bank conflicts)
    constexpr bool pendk = KIND == DDP_PROBLEM_PENDCART;
    __shared__ double ctile[FUSE ? GPW * 16 * TS : 1];
    static_assert(NS <= G && MS <= G, "state must fit one DPP row");
    const int N = a.N, B = a.B;
    const int lane = threadIdx.x, grp = lane / G, j = lane % G;
    const long total = (long)B * a.nalpha;
    // the step sizes of ONE trajectory are neighbours (rows of the same wave, neighbouring waves): their loads of K_i, k_i, x_i, ū_i hit
    // the same cache lines, and a finished trajectory takes whole waves out of a line search
    long lin = (long)blockIdx.x * GPW + grp;
    const bool valid = lin < total;
    if (!valid) lin = total - 1;                                // keep all lanes alive (DPP reads every lane)
    const int b = (int)(lin / a.nalpha), ai = (int)(lin % a.nalpha);
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;                                    // every rollout of this wave belongs to a finished trajectory
    const double alpha = a.alpha[ai];
    // FAST, LQ: the idle lanes n+m .. 15 of a row MIRROR lanes 0 .. : same row of A and B, same loads, same x̂, same store to the same
    // address — every lane stores every step without an exec mask, and nobody writes to a shared dump address (with a machine-filling
    // batch 8 192 waves storing their idle lanes to ONE line made the rollout 9 % slower than the masked variant).  Their cost weight is 0.
    constexpr bool MIR = FAST && KIND == DDP_PROBLEM_LQ;
    const int jm = MIR ? j % (n + m) : j;
    const bool inx = jm < n, inu = j < m;
    const int jx = inx ? jm : 0, ju = inu ? j : 0;

    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m;
    const double *ug = a.u + (size_t)m * N * b;
    const double *xg = POLICY ? a.x + (size_t)n * N * b : nullptr;
    const double *Kg = POLICY ? a.K + nm * N * b : nullptr;
    const double *kg = POLICY ? a.k + (size_t)m * N * b : nullptr;
    double *xo = a.xnew + (size_t)n * N * ((size_t)b + (size_t)B * ai);
    double *uo = a.unew + (size_t)m * N * ((size_t)b + (size_t)B * ai);
    const double *Ab = nullptr, *Bb = nullptr;
    if (KIND == DDP_PROBLEM_LQ) {
        Ab = a.A + (a.dyn_batched ? nn * (a.dyn_tv ? N : 1) * b : 0);
        Bb = a.Bm + (a.dyn_batched ? nm * (a.dyn_tv ? N : 1) * b : 0);
    }
    // row j of A and B (zero rows for the idle lanes j >= n, so their x̂ stays 0)
    double Arow[n], Brow[m];
    auto load_dyn = [&](int i) {
        if (KIND == DDP_PROBLEM_LQ) {
            const size_t oa = a.dyn_tv ? nn * i : 0, ob = a.dyn_tv ? nm * i : 0;
            const double z = inx ? 1.0 : 0.0;
#pragma unroll
            for (int l = 0; l < n; ++l) Arow[l] = z * Ab[oa + jx + n * l];
#pragma unroll
            for (int q = 0; q < m; ++q) Brow[q] = z * Bb[ob + jx + n * q];
        }
    };
    load_dyn(0);
    double lo[m], hi[m];
#pragma unroll
    for (int q = 0; q < m; ++q) { lo[q] = LIMS ? a.lims[q] : 0.0; hi[q] = LIMS ? a.lims[q + m] : 0.0; }
    bool minmax_ok = true;                                      // clamp by v_max / v_min equals the reference's clamp only for ordered bounds
#pragma unroll
    for (int q = 0; q < m; ++q) minmax_ok = minmax_ok && (lo[q] <= hi[q]);
    double one = 1.0;
    asm volatile("" : "+v"(one));                               // keep 1.0 in a VGPR (DPP src1 must be a VGPR)

    double xh = inx ? a.x0[(size_t)n * b + jx] : 0.0;           // x̂_j
    // fused cost: weight and offset of this lane's stored value; per-lane part of sum(cnew)
    double cw = 0.0, cg = 0.0, cacc = 0.0, xlast = 0.0;
    double *co = nullptr;
    const int CL = pendk ? N + 1 : N;
    if (FUSE) {
        if (j < n) { cw = 0.5 * a.Q[jx + n * jx]; cg = pendk ? a.goal[jx & 3] : 0.0; }
        else if (j < n + m) cw = 0.5 * a.R[(j - n) + m * (j - n)];
        co = a.cnew + (size_t)CL * ((size_t)b + (size_t)B * ai);
    }
    double *ct = &ctile[FUSE ? grp * 16 * TS : 0];
    double *ctw = ct + j;                                        // this lane's column of tile rows (i & 8) .. (i & 8) + 7, set per group of 8 steps
    // cost of the steps [i0c, i0c + cnt) that wait in the tile: lane t sums step i0c + t
    auto flush_cost = [&](int i0c, int cnt) {
        wave_sync();
        double c = 0.0;
#pragma unroll
        for (int l = 0; l < n + m; ++l) c += ct[j * TS + l];
        if (j < cnt) {
            if (act) co[i0c + j] = c;
            cacc += c;
        }
        wave_sync();
    };

    // one predicated store per step: lane j < n writes xnew[j,i], lanes n..n+m-1 write unew[j-n,i]
    const bool st_u = jm >= n && jm < n + m;
    const bool st_on = act && (inx || st_u);
    double *st_base = (FAST && !st_on) ? a.sink + lane : (st_u ? uo + (jm - n) : xo + jx);
    const unsigned st_stride = (FAST && !st_on) ? 0u : (st_u ? m : n) * (unsigned)sizeof(double);

    // Loads are UNCONDITIONAL (clamped lane index): a load inside an exec-masked branch makes the compiler
    // drain all outstanding loads (s_waitcnt vmcnt(0)) at the join and would serialise the prefetch ring.
    struct Ops { double u[m], k[m], K[m], x; };                 // ū_i, k_i (row-uniform), K_i[:, j], x_i[j]
    auto fetch = [&](int i, Ops &o) {
        if constexpr (FAST && POLICY && m == 2) {
            // ū_i, k_i, K_i[:, j] as ONE 16-byte load each (the launcher has checked the alignment): 4 instead of 7 vector-memory
            // instructions per step — with a machine-filling batch the rollout is bound by the address unit its four SIMDs share
            typedef double d2v __attribute__((ext_vector_type(2)));
            const d2v uv = *(const d2v *)(ug + (size_t)m * i), kv = *(const d2v *)(kg + (size_t)m * i),
                      Kv = *(const d2v *)(Kg + nm * i + m * jx);
            o.u[0] = uv.x; o.u[1] = uv.y; o.k[0] = kv.x; o.k[1] = kv.y; o.K[0] = Kv.x; o.K[1] = Kv.y;
            o.x = xg[(size_t)n * i + jx];
            return;
        }
#pragma unroll
        for (int q = 0; q < m; ++q) o.u[q] = ug[(size_t)m * i + q];
        if (POLICY) {
#pragma unroll
            for (int q = 0; q < m; ++q) {
                o.k[q] = kg[(size_t)m * i + q];
                o.K[q] = Kg[nm * i + q + m * jx];
            }
            o.x = xg[(size_t)n * i + jx];
        }
    };
    auto step = [&](int i, const Ops &o, bool advance) {
        // ---- controls (forward_pass.jl:17-24): u = ū + α k + K (x̂ - x), clamp, NaN -> 0 (inside f)
        double uu[m];
        double ax = 0.0;                                                 // (A x̂)_j once it has been formed
        bool ax_done = false;
        if (POLICY) {
            double pr[m];
            const double dx = xh - o.x;
#pragma unroll
            for (int q = 0; q < m; ++q) pr[q] = o.K[q] * dx;
            if constexpr (m == 2) dpp_fence(pr[0], pr[1]);
            else {
#pragma unroll
                for (int q = 0; q < m; ++q) dpp_fence(pr[q]);
            }
#pragma unroll
            for (int q = 0; q < m; ++q) {
                double s0 = o.u[q] + o.k[q] * alpha, s1 = 0.0;            // unew .+= k*α, then .+= K*dx
                RowSum<n>::run(s0, s1, pr[q], one);
                uu[q] = s0 + s1;
            }
        } else {
#pragma unroll
            for (int q = 0; q < m; ++q) uu[q] = o.u[q];
        }
        {   // clamp (forward_pass.jl:21-23), then u[isnan.(u)] .= 0 inside f (demo_linear.jl:36, system_pendcart.jl:120).  A NaN control
            // is rare: one test of the sum of the unclamped entries (NaN if any entry is; Inf - Inf is a false alarm the slow path sorts
            // out) and a wave-uniform branch replace a compare and two selects per entry and step; without a NaN in the wave the clamp is
            // v_max / v_min (two instructions on the chain u -> x̂⁺ instead of two compares and four selects; needs lo <= hi).
            if constexpr (m == 1) {                                       // one entry: the plain selects are cheaper than the branch (measured:
                if (LIMS) uu[0] = clampd(uu[0], lo[0], hi[0]);            // pendulum rollout 0.227 ms against 0.236 ms)
                if (uu[0] != uu[0]) uu[0] = 0.0;
            } else {
                double t = uu[0];
#pragma unroll
                for (int q = 1; q < m; ++q) t += uu[q];
                unsigned long long nanmask = __builtin_amdgcn_ballot_w64(t != t);
                asm volatile("" : "+s"(nanmask));                         // the compare is issued HERE ...
                if (KIND == DDP_PROBLEM_LQ && advance && (FAST || !a.dyn_tv)) {
                    // ... and the part of the dynamics that does not depend on the controls runs while its result travels to the
                    // scalar unit (the branch below would otherwise wait for it)
                    double s0 = 0.0, s1 = 0.0;
                    RowDot<n>::run(s0, s1, xh, Arow);                     // Σ_l A[j,l] x̂_l
                    ax = s0 + s1; ax_done = true;
                }
                if (__builtin_expect(nanmask != 0 || !minmax_ok, 0)) {
#pragma unroll
                    for (int q = 0; q < m; ++q) {
                        if (LIMS) uu[q] = clampd(uu[q], lo[q], hi[q]);
                        if (uu[q] != uu[q]) uu[q] = 0.0;
                    }
                } else if (LIMS) {
#pragma unroll
                    for (int q = 0; q < m; ++q) uu[q] = fmin(fmax(uu[q], lo[q]), hi[q]);
                }
            }
        }
        {
            double v = xh;
#pragma unroll
            for (int q = 0; q < m; ++q) v = (jm == n + q) ? uu[q] : v;
            if (FAST || st_on) *(double *)((char *)st_base + (size_t)i * st_stride) = v;
            if (FUSE) {
                const double dv = pendk ? v - cg : v;
                const double pc = (cw * dv) * dv;
                ctw[(i & 7) * TS] = pc;                                  // row (i & 15) of the tile; idle lanes write 0 (cw = 0)
                if (pendk && i == N - 1) xlast = inx ? pc : 0.0;         // the extra entry re-counts x[:,N] without a control
            }
        }
        // ---- dynamics (f is also called at i == N in the reference, its result is discarded)
        if (advance) {
            double xp;
            if (KIND == DDP_PROBLEM_LQ) {
                if (!FAST && a.dyn_tv) load_dyn(i);
                double t = 0.0;
                if (!ax_done) {
                    double s0 = 0.0, s1 = 0.0;
                    RowDot<n>::run(s0, s1, xh, Arow);                    // Σ_l A[j,l] x̂_l
                    ax = s0 + s1;
                }
                if (ax_done) {                                           // two dependent multiply-adds on the chain u -> x̂⁺ instead of three operations
                    xp = ax;
#pragma unroll
                    for (int q = 0; q < m; ++q) xp = fma(Brow[q], uu[q], xp);
                } else {
#pragma unroll
                    for (int q = 0; q < m; ++q) t += Brow[q] * uu[q];
                    xp = ax + t;                                         // A*x + B*u
                }
            } else {                                                     // system_pendcart.jl:83-89
                const double x0v = row_bcast<0>(xh), x1v = row_bcast<1>(xh), x3v = row_bcast<3>(xh);
                const double gl = a.g / a.l, h = a.h;
                double sn, cs;
                sincos(x0v, &sn, &cs);                                   // one argument reduction for both
                const double f1 = x1v + h * (-gl * sn + uu[0] / a.l * cs - a.d * x1v);
                xp = (j == 0) ? x0v + h * x1v : (j == 1) ? f1 : (j == 2) ? xh + h * x3v : x3v + h * uu[0];
                xp = inx ? xp : 0.0;
            }
            xh = xp;
            dpp_fence(xh);
        }
    };
    dpp_fence(xh);
    // A time step is ~60 instructions (~0.15 us) but an HBM load takes ~1 us: the operands of step i+D are
    // requested while step i runs (ring of D register sets, loop unrolled by D so every slot is a fixed
    // register).  The main loop is branch-free; the last < 2D steps run in a guarded copy.
    constexpr int D = 8;
    Ops ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d < N ? d : N - 1, ring[d]);
    int i0 = 0;
    static_assert(D == 8, "the cost tile is addressed in groups of 8 steps");
    for (; i0 + 2 * D <= N; i0 += D) {
        if (FUSE) ctw = ct + j + (i0 & 8) * TS;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            step(i0 + d, ring[d], true);
            fetch(i0 + d + D, ring[d]);
        }
        if (FUSE && (i0 & 8)) flush_cost(i0 - 8, 16);                    // steps i0-8 .. i0+7 are in the tile
    }
    for (; i0 < N; i0 += D) {
        if (FUSE) ctw = ct + j + (i0 & 8) * TS;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = i0 + d;
            if (i < N) {
                step(i, ring[d], i < N - 1);
                fetch(i + D < N ? i + D : N - 1, ring[d]);
            }
        }
        if (FUSE && ((i0 & 8) || i0 + D >= N)) {
            const int c0 = i0 & ~15;
            flush_cost(c0, (N - c0) < 16 ? N - c0 : 16);
        }
    }
    if (FUSE) {
        if (pendk) {                                                     // c[N] = ½ (x_N - goal)'Q(x_N - goal)  (system_pendcart.jl:105)
            double s0 = 0.0, s1 = 0.0;
            dpp_fence(xlast);
            RowSum<n>::run(s0, s1, xlast, one);
            const double cN = s0 + s1;
            if (act && j == 0) co[N] = cN;
            cacc += (j == 0) ? cN : 0.0;
        }
        double s0 = 0.0, s1 = 0.0;
        dpp_fence(cacc);
        RowSum<16>::run(s0, s1, cacc, one);
        if (act && j == 0) a.csum[(size_t)b + (size_t)B * ai] = s0 + s1;
    }
}


// ---- pendcart, ONE LANE per (trajectory, α) rollout.  In the row kernel above the 16 lanes of a row all evaluate the same
// sincos (and only 5 of them carry state): with the 6-11 step sizes of a line search there are enough rollouts to give every
// lane its own — 16x fewer wave-instructions per rollout.  Same arithmetic (system_pendcart.jl:83-89, forward_pass.jl:17-24;
// K·dx summed in index order).  Operands are prefetched DL steps ahead (a lane's loads are its own 32-byte pieces).
template <bool POLICY, bool LIMS, bool FUSE, bool WRAP = false>
__global__ __launch_bounds__(DDP_WAVE) void forward_lane_pendcart_kernel(FDArgs a)
{
    constexpr int n = 4, DL = 4;
    const int N = a.N, B = a.B;
    const long total = (long)B * a.nalpha;
    // the step sizes of ONE trajectory sit in neighbouring lanes: their loads of K_i, k_i, x_i, ū_i hit the same cache lines, so a
    // load instruction touches 64/nalpha lines instead of 64 (the address unit works line by line)
    long lin = (long)blockIdx.x * DDP_WAVE + threadIdx.x;
    const bool valid = lin < total;
    if (!valid) lin = total - 1;
    const int b = (int)(lin / a.nalpha), ai = (int)(lin % a.nalpha);
    const long rho = (long)b + (long)B * ai;                    // index of the rollout in xnew / unew / cnew / csum
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;                                    // every rollout of this wave belongs to a finished trajectory
    const double alpha = a.alpha[ai];
    const double *ug = a.u + (size_t)N * b;
    const double *xg = POLICY ? a.x + (size_t)n * N * b : nullptr;
    const double *Kg = POLICY ? a.K + (size_t)n * N * b : nullptr;
    const double *kg = POLICY ? a.k + (size_t)N * b : nullptr;
    double *xo = a.xnew + (size_t)n * N * rho, *uo = a.unew + (size_t)N * rho;
    const double lo = LIMS ? a.lims[0] : 0.0, hi = LIMS ? a.lims[1] : 0.0;
    const double gl = a.g / a.l, il = 1.0 / a.l, h = a.h;
    PendTrig trig;
    trig.init();
    struct Ops { double u, k; d2 K0, K1, x0, x1; };
    auto fetch = [&](int i, Ops &o) {
        o.u = ug[i];
        if (POLICY) {
            o.k = kg[i];
            o.K0 = *(const d2 *)(Kg + (size_t)n * i); o.K1 = *(const d2 *)(Kg + (size_t)n * i + 2);
            o.x0 = *(const d2 *)(xg + (size_t)n * i); o.x1 = *(const d2 *)(xg + (size_t)n * i + 2);
        }
    };
    double x0v = a.x0[(size_t)n * b], x1v = a.x0[(size_t)n * b + 1], x2v = a.x0[(size_t)n * b + 2], x3v = a.x0[(size_t)n * b + 3];
    // fused cost (cost_diag): c_i = ½((x_i-goal)'Q(x_i-goal) + R u_i²), c_{N+1} = ½(x_N-goal)'Q(x_N-goal)  (system_pendcart.jl:97-106)
    double q0 = 0, q1 = 0, q2 = 0, q3 = 0, rr = 0, cacc = 0.0, qxl = 0.0;
    double *co = nullptr;
    if (FUSE) {
        q0 = a.Q[0]; q1 = a.Q[5]; q2 = a.Q[10]; q3 = a.Q[15]; rr = a.R[0];
        co = a.cnew + (size_t)(N + 1) * rho;
    }
    auto step = [&](int i, const Ops &o, bool advance) {
        double uu = o.u;
        if (POLICY) {
            uu += o.k * alpha;                                           // unew .+= k*α
            double e0 = x0v - o.x0.x, e1 = x1v - o.x0.y, e2 = x2v - o.x1.x, e3 = x3v - o.x1.y;
            if (WRAP) {                                                  // diff_fun: the named coordinates wrapped to [-π, π] (wave-uniform tests)
                if (a.wrap & 1u) e0 = wrap_pi2(e0);
                if (a.wrap & 2u) e1 = wrap_pi2(e1);
                if (a.wrap & 4u) e2 = wrap_pi2(e2);
                if (a.wrap & 8u) e3 = wrap_pi2(e3);
            }
            double s = o.K0.x * e0;
            s += o.K0.y * e1;
            s += o.K1.x * e2;
            s += o.K1.y * e3;
            uu += s;                                                     // unew .+= K*diff_fun(x̂, x)
        }
        if (LIMS) uu = clampd(uu, lo, hi);
        if (uu != uu) uu = 0.0;
        if (act) {
            *(d2 *)(xo + (size_t)n * i) = d2{x0v, x1v};
            *(d2 *)(xo + (size_t)n * i + 2) = d2{x2v, x3v};
            uo[i] = uu;
        }
        if (FUSE) {
            const double d0 = x0v - a.goal[0], d1 = x1v - a.goal[1], d2 = x2v - a.goal[2], d3 = x3v - a.goal[3];
            const double qx = ((q0 * d0) * d0 + (q1 * d1) * d1) + ((q2 * d2) * d2 + (q3 * d3) * d3);
            const double c = 0.5 * (qx + (rr * uu) * uu);
            if (act) co[i] = c;
            cacc += c;
            qxl = qx;
        }
        if (advance) {                                                   // system_pendcart.jl:83-89
            double sn, cs;
            pend_sincos(trig, x0v, sn, cs);
            double ul = uu * il;
            ul = __builtin_fma(__builtin_fma(-ul, a.l, uu), il, ul);     // u / l: one Newton correction of u · (1/l)
            const double f1 = x1v + h * (-gl * sn + ul * cs - a.d * x1v);
            const double n0 = x0v + h * x1v, n2 = x2v + h * x3v, n3 = x3v + h * uu;
            x0v = n0; x1v = f1; x2v = n2; x3v = n3;
        }
    };
    Ops ring[DL];
#pragma unroll
    for (int d = 0; d < DL; ++d) fetch(d < N ? d : N - 1, ring[d]);
    int i0 = 0;
    for (; i0 + 2 * DL <= N; i0 += DL) {
#pragma unroll
        for (int d = 0; d < DL; ++d) {
            step(i0 + d, ring[d], true);
            fetch(i0 + d + DL, ring[d]);
        }
    }
    for (; i0 < N; i0 += DL) {
#pragma unroll
        for (int d = 0; d < DL; ++d) {
            const int i = i0 + d;
            if (i < N) {
                step(i, ring[d], i < N - 1);
                fetch(i + DL < N ? i + DL : N - 1, ring[d]);
            }
        }
    }
    if (FUSE && act) {
        co[N] = 0.5 * qxl;
        a.csum[rho] = cacc + 0.5 * qxl;
    }
}


// ---- pendcart, one 16-lane row per rollout, written for the step count of a LONE wave (B = 4 096 single-α rollouts are one wave per
// SIMD: the pass takes as long as one wave needs for its N steps, ~6.5 cycles per vector instruction and ~60 per vector-memory
// instruction).  Against the generic row kernel above (132 vector + 5 vector-memory instructions per step):
//   * ONE load per step: lanes 0-3 fetch K_i[j], lanes 4-7 x_i[j-4], lane 8 ū_i, lane 9 k_i — every lane walks its own array with its
//     own stride; x_i moves to the state lanes by a row shift, ū_i and k_i enter the multiply-adds as row broadcasts;
//   * sin, cos by pend_sincos (pend_math.h, ~42 instructions instead of ~70), u / l as u · (1/l) with one Newton correction (3
//     instructions; a division is ~10);
//   * the four components of x̂⁺ = x̂ + h·(x̂_1, a, x̂_3, u) as ONE multiply-add on a lane-selected increment instead of a branch tree;
//   * the clamp by v_max / v_min when the bounds are ordered (the loop exists twice; the other copy keeps the reference's compares),
//     the NaN test on the unclamped control.
// Lanes 0-3 hold x̂, lane 4 the control of the step (store, cost).  Same statements as forward_pass.jl:17-24 and
// system_pendcart.jl:83-89; cost tile as in the kernel above.
template <bool V> struct BoolC { static constexpr bool value = V; };
// every lane of a 16-lane row <- lane L of the row (no `old` operand: all lanes are written, so no zero has to be moved in first)
template <int L>
__device__ __forceinline__ double row_bcast_all(double x)
{
    double d;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(x), "n"(L));
    return d;
}
template <int I> struct PIC { static constexpr int value = I; };
template <int I, int E, class Fn>
__device__ __forceinline__ void pend_static_for(Fn &&f)
{
    if constexpr (I < E) { f(PIC<I>{}); pend_static_for<I + 1, E>(f); }
}
typedef double pend_d4 __attribute__((ext_vector_type(4)));
// Memory side (round 6, profiles/microbench/narrow_streams.hip): one 8-byte element per lane and step from six streams per rollout tops out at
// 1.7-1.85 TB/s however many rollouts run (every request is 8-32 bytes wide) — above the 208 ns compute floor of a step from ~2 600 rollouts
// on; the same bytes in 512-byte runs reach 5.3-7.7 TB/s.  So the whole 16-step chunks of a rollout go through LDS: per chunk a lane
// fetches 32 bytes of K and of x (the step 16 c + j of its row) and 8 of ū and k, one chunk ahead; a step reads its K_i | x_i element from
// the LDS image (one ds_read in place of the global load, a step ahead), takes ū_i and k_i from the chunk registers by the same row
// broadcast as before (lane d of the row instead of lanes 8 / 9), and puts x̂_i | u_i into an LDS image that leaves as 32 + 8 bytes per
// lane at the end of the chunk.  The N mod 16 last steps run the element-wise path.
#ifndef PEND_ROW_CHUNK
#define PEND_ROW_CHUNK 1
#endif
template <bool POLICY, bool LIMS, bool FUSE, bool WRAP = false>
__global__ __launch_bounds__(DDP_WAVE) void forward_pend_row_kernel(FDArgs a)
{
    constexpr int n = 4, G = 16, GPW = DDP_WAVE / G, TS = 17, D = 8;
    const bool wrapj = WRAP && (((a.wrap >> (threadIdx.x & 3)) & 1u) != 0) && (threadIdx.x % G) < n;      // this lane's coordinate is an angle
    __shared__ double ctile[FUSE ? GPW * 16 * TS : 1];
    constexpr int CIN = 128, COUT = 208;                                      // doubles per rollout: [K 16 x 4 | x 16 x 4] per buffer; [x̂ 16 x 4 | u 16 x 4 (stride 4) | dump]
    __shared__ __attribute__((aligned(16))) double cin[(PEND_ROW_CHUNK && POLICY) ? GPW * 2 * CIN : 2];
    __shared__ __attribute__((aligned(16))) double cout[PEND_ROW_CHUNK ? GPW * COUT : 2];
    const int N = a.N, B = a.B;
    const int lane = threadIdx.x, grp = lane / G, j = lane % G;
    const long total = (long)B * a.nalpha;
    long lin = (long)blockIdx.x * GPW + grp;
    const bool valid = lin < total;
    if (!valid) lin = total - 1;
    const int b = (int)(lin / a.nalpha), ai = (int)(lin % a.nalpha);
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;
    const double alpha = a.alpha[ai];
    const bool inx = j < n;
    const int jx = inx ? j : 0;
    // this lane's operand stream: (pointer to its entry of step 0, bytes per step)
    const double *ug = a.u + (size_t)N * b;
    const char *ldb = (const char *)ug;
    unsigned lds = 0;                                            // lanes without an operand re-read ū_0
    if (POLICY) {
        if (j < 4) { ldb = (const char *)(a.K + (size_t)n * N * b + j); lds = n * 8; }
        else if (j < 8) { ldb = (const char *)(a.x + (size_t)n * N * b + (j - 4)); lds = n * 8; }
        else if (j == 9) { ldb = (const char *)(a.k + (size_t)N * b); lds = 8; }
    }
    if (j == 8) lds = 8;
    auto fetch = [&](int i) -> double { return *(const double *)(ldb + (size_t)lds * (unsigned)i); };
    // this lane's result stream: x̂_i[j] (lanes 0-3), u_i (lane 4); the others write to the sink
    const size_t rho = (size_t)b + (size_t)B * ai;
    const bool st_on = act && j <= n;
    char *stb = !st_on ? (char *)(a.sink + lane) : (j < n ? (char *)(a.xnew + (size_t)n * N * rho + j) : (char *)(a.unew + (size_t)N * rho));
    const unsigned sts = !st_on ? 0u : (j < n ? n * 8u : 8u);
    const double lo = LIMS ? a.lims[0] : 0.0, hi = LIMS ? a.lims[1] : 0.0;
    const double gl = a.g / a.l, il = 1.0 / a.l, h = a.h, dd = a.d;
    double one = 1.0;
    asm volatile("" : "+v"(one));
    double xh = inx ? a.x0[(size_t)n * b + jx] : 0.0;
    // fused cost
    double cw = 0.0, cg = 0.0, cacc = 0.0, xlast = 0.0;
    double *co = nullptr;
    if (FUSE) {
        if (inx) { cw = 0.5 * a.Q[jx + n * jx]; cg = a.goal[jx]; }
        else if (j == n) cw = 0.5 * a.R[0];
        co = a.cnew + (size_t)(N + 1) * rho;
    }
    double *ct = &ctile[FUSE ? grp * 16 * TS : 0];
    double *ctw = ct + j;
    auto flush_cost = [&](int i0c, int cnt) {
        wave_sync();
        double c = 0.0;
#pragma unroll
        for (int l = 0; l < n + 1; ++l) c += ct[j * TS + l];
        if (j < cnt) {
            if (act) co[i0c + j] = c;
            cacc += c;
        }
        wave_sync();
    };
    const bool is1 = j == 1, is3 = j == 3, isu = j == n;
    double *co_ = &cout[PEND_ROW_CHUNK ? grp * COUT : 0];
    double *cow = co_ + (j < n ? j : (j == n ? 64 : 128 + j));            // my cell of a step's result (+ 4 per step); lanes > n: dump cells
    PendTrig trig;
    trig.init();
    dpp_fence(xh);
    auto run = [&](auto minmax_c) __attribute__((always_inline)) {
    constexpr bool MINMAX = decltype(minmax_c)::value;
    // ld: K_i | x_i in lanes 0-7; ū_i in lane UL of usrc, k_i in lane KL of ksrc (element-wise path: all three are the one loaded register)
    auto step = [&](int i, double ld, double usrc, double ksrc, auto ul_c, auto kl_c, bool advance, auto store_c) __attribute__((always_inline)) {
        constexpr int UL = decltype(ul_c)::value, KL = decltype(kl_c)::value;
        constexpr bool TO_LDS = decltype(store_c)::value >= 0;
        // ---- control (forward_pass.jl:17-24): u = ū + α k + K (x̂ - x), clamp, NaN -> 0 (inside f, system_pendcart.jl:120)
        double uu = row_bcast_all<UL>(usrc);                             // ū_i  (from memory: no VALU -> DPP hazard)
        if (POLICY) {
            const double xi = __builtin_amdgcn_update_dpp(0.0, ld, 0x104, 0xf, 0xf, true);      // row_shl:4: x_i[j] from lane j + 4
            double dxj = xh - xi;
            if (WRAP) dxj = wrapj ? wrap_pi2(dxj) : dxj;                 // diff_fun (forward_pass.jl:19)
            double pr = ld * dxj;                                        // K_i[j] diff(x̂_j, x_j) in lanes 0-3
            fmac_bc<KL>(uu, ksrc, alpha);                                // unew .+= k*α
            dpp_fence(pr);
            double s1 = 0.0;
            RowSum<n>::run(uu, s1, pr, one);                             // unew .+= K*dx, two interleaved partial sums
            uu += s1;
        }
        const bool nan = uu != uu;
        if (LIMS) uu = MINMAX ? fmin(fmax(uu, lo), hi) : clampd(uu, lo, hi);
        uu = nan ? 0.0 : uu;
        const double v = isu ? uu : xh;
        if constexpr (TO_LDS) cow[4 * decltype(store_c)::value] = v;     // x̂_i | u_i into the chunk image
        else *(double *)(stb + (size_t)sts * (unsigned)i) = v;
        if (FUSE) {
            const double dv = v - cg;
            const double pc = (cw * dv) * dv;
            ctw[(i & 7) * TS] = pc;
            if (i == N - 1) xlast = inx ? pc : 0.0;                      // c[N+1] re-counts x[:,N] without a control (system_pendcart.jl:105)
        }
        if (advance) {                                                   // system_pendcart.jl:83-89
            const double x0v = row_bcast_all<0>(xh), x1v = row_bcast_all<1>(xh);
            double sn, cs;
            pend_sincos(trig, x0v, sn, cs);
            double ul = uu * il;                                         // u / l, correctly rounded but for rare cases, in three instructions
            ul = __builtin_fma(__builtin_fma(-ul, a.l, uu), il, ul);     // (a division is ~10): one Newton correction of u · (1/l)
            double acc = -gl * sn + ul * cs - dd * x1v;
            asm("" : "+v"(acc));                                         // in every lane: selects below, not an exec-mask branch around three instructions
            const double nxt = __builtin_amdgcn_update_dpp(0.0, xh, 0xf9, 0xf, 0xf, true);      // quad_perm [1,2,3,3]: x̂_{j+1} in lanes 0 and 2
            const double inc = is1 ? acc : (is3 ? uu : nxt);             // lanes >= 4: their right-hand neighbour's 0
            xh = xh + h * inc;
            dpp_fence(xh);
        }
    };
    int i0 = 0;
    if (PEND_ROW_CHUNK && a.chunked && N >= 16) {
        const int nch = N / 16;
        const bool lastadv = 16 * nch < N;                                  // the last step of the last chunk advances unless it is step N-1
        const pend_d4 *K4 = (const pend_d4 *)(a.K + (size_t)n * N * b) + j, *x4 = (const pend_d4 *)(a.x + (size_t)n * N * b) + j;
        const double *u1 = ug + j, *k1 = POLICY ? a.k + (size_t)N * b + j : ug + j;
        pend_d4 Kn = pend_d4{0, 0, 0, 0}, xn = Kn;
        double uch = u1[0], kch = POLICY ? k1[0] : 0.0, un, kn = 0.0;
        double *ci = &cin[POLICY ? grp * 2 * CIN : 0];
        const int rd = j < 4 ? j : (j < 8 ? 64 + (j - 4) : 0);           // my element of a step in an input buffer (+ 4 per step)
        if (POLICY) {
            Kn = K4[0]; xn = x4[0];
            *(pend_d4 *)(ci + 4 * j) = Kn; *(pend_d4 *)(ci + 64 + 4 * j) = xn;
        }
        pend_d4 *xo4 = (pend_d4 *)(a.xnew + (size_t)n * N * rho) + j;
        double *uo1 = a.unew + (size_t)N * rho + j;
        wave_sync();
        double ldc = POLICY ? (ci + rd)[0] : 0.0;                          // K_i | x_i of the step to come: read a step ahead, across chunks too
        for (int c = 0; c < nch; ++c) {
            const int cn = c + 1 < nch ? c + 1 : c;                        // the chunk after this one (clamped: always a valid load)
            if (POLICY) { Kn = K4[16 * cn]; xn = x4[16 * cn]; kn = k1[16 * cn]; }
            un = u1[16 * cn];
            const double *cr = ci + (c & 1) * CIN + rd;
            const bool advl = c + 1 < nch || lastadv;
            pend_static_for<0, 16>([&](auto dc) __attribute__((always_inline)) {
                constexpr int d = decltype(dc)::value;
                if (FUSE && (d == 0 || d == 8)) ctw = ct + j + d * TS;
                const double ld = ldc;
                if (POLICY && d < 15) ldc = cr[4 * (d + 1)];
                step(16 * c + d, ld, uch, kch, dc, dc, d < 15 ? true : advl, dc);
            });
            // end of the chunk: one round of LDS traffic between two fences (the four rollouts of the wave are the only users of these tiles,
            // and a wave's LDS operations stay in order) — the next input image, the result image and the cost tile of the chunk
            wave_sync();
            double *cw2 = ci + ((c + 1) & 1) * CIN;
            if (POLICY) { *(pend_d4 *)(cw2 + 4 * j) = Kn; *(pend_d4 *)(cw2 + 64 + 4 * j) = xn; }
            const pend_d4 xv = *(const pend_d4 *)(co_ + 4 * j);
            const double uv = co_[64 + 4 * j];
            double cs = 0.0;
            if (FUSE) {
#pragma unroll
                for (int l = 0; l < n + 1; ++l) cs += ct[j * TS + l];          // (flush_cost(16 c, 16))
            }
            wave_sync();
            if (POLICY) ldc = (cw2 + rd)[0];
            if (act) { xo4[16 * c] = xv; uo1[16 * c] = uv; }
            if (FUSE) {
                if (act) co[16 * c + j] = cs;
                cacc += cs;
            }
            uch = un; kch = kn;
            dpp_fence(uch, kch);
        }
        i0 = 16 * nch;
    }
    double ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = fetch(i0 + d < N ? i0 + d : N - 1);
    for (; i0 + 2 * D <= N; i0 += D) {
        if (FUSE) ctw = ct + j + (i0 & 8) * TS;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            step(i0 + d, ring[d], ring[d], ring[d], PIC<8>{}, PIC<9>{}, true, PIC<-1>{});
            ring[d] = fetch(i0 + d + D);
        }
        if (FUSE && (i0 & 8)) flush_cost(i0 - 8, 16);
    }
    for (; i0 < N; i0 += D) {
        if (FUSE) ctw = ct + j + (i0 & 8) * TS;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = i0 + d;
            if (i < N) {
                step(i, ring[d], ring[d], ring[d], PIC<8>{}, PIC<9>{}, i < N - 1, PIC<-1>{});
                ring[d] = fetch(i + D < N ? i + D : N - 1);
            }
        }
        if (FUSE && ((i0 & 8) || i0 + D >= N)) {
            const int c0 = i0 & ~15;
            flush_cost(c0, (N - c0) < 16 ? N - c0 : 16);
        }
    }
    };
    if (!LIMS || lo <= hi) run(BoolC<true>{}); else run(BoolC<false>{});
    if (FUSE) {
        double s0 = 0.0, s1 = 0.0;
        dpp_fence(xlast);
        RowSum<n>::run(s0, s1, xlast, one);
        const double cN = s0 + s1;
        if (act && j == 0) co[N] = cN;
        cacc += (j == 0) ? cN : 0.0;
        s0 = 0.0; s1 = 0.0;
        dpp_fence(cacc);
        RowSum<16>::run(s0, s1, cacc, one);
        if (act && j == 0) a.csum[rho] = s0 + s1;
    }
}

// ---- per-step cost + its sum: one wave per rollout, lanes over time (costfun of the registered families)
//   LQ        c_i = .5 x_i'Q x_i + .5 u_i'R u_i                      (src/demo_linear.jl:49, split per step)
//   pendcart  c_i = .5 ((x_i-goal)'Q(x_i-goal) + R u_i^2), c_{N+1} = .5 (x_N-goal)'Q(x_N-goal)
//                                                                     (src/system_pendcart.jl:97-106)
struct CostArgs {
    int kind, n, m, N, B, nalpha;
    const double *Q, *R;
    const int32_t *active;
    double goal[4];
    const double *xnew, *unew;
    double *cnew, *csum;
};

template <int KIND, int NS, int MS>
__global__ __launch_bounds__(DDP_WAVE) void cost_kernel(CostArgs a)
{
    constexpr int n = NS, m = MS;
    constexpr bool pend = KIND == DDP_PROBLEM_PENDCART;
    const int N = a.N, B = a.B;
    const long rho = blockIdx.x;
    const int b = (int)(rho % B);
    if (a.active && a.active[b] == 0) return;
    const int lane = threadIdx.x;
    const int CL = pend ? N + 1 : N;
    const double *x = a.xnew + (size_t)n * N * rho, *u = a.unew + (size_t)m * N * rho;
    double *c = a.cnew + (size_t)CL * rho;
    __shared__ double qr[n * n + m * m];             // Q | R, read back as wave-uniform (broadcast) operands
    for (int e = lane; e < n * n; e += DDP_WAVE) qr[e] = a.Q[e];
    for (int e = lane; e < m * m; e += DDP_WAVE) qr[n * n + e] = a.R[e];
    wave_sync();
    const double *Q = qr, *R = qr + n * n;
    double acc = 0.0;
    for (int t = lane; t < CL; t += DDP_WAVE) {
        const int tx = t < N ? t : N - 1;            // pendcart: the extra entry re-counts x[:,N] with u = 0
        double xt[n], ut[m];
#pragma unroll
        for (int i = 0; i < n; ++i) xt[i] = x[(size_t)n * tx + i] - (pend ? a.goal[i & 3] : 0.0);
#pragma unroll
        for (int i = 0; i < m; ++i) ut[i] = u[(size_t)m * tx + i];
        double qx = 0.0;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
#pragma unroll
            for (int jj = 0; jj < n; ++jj) s += Q[i + n * jj] * xt[jj];
            qx += xt[i] * s;
        }
        double ru = 0.0;
        if (t < N) {
#pragma unroll
            for (int i = 0; i < m; ++i) {
                double s = 0.0;
#pragma unroll
                for (int jj = 0; jj < m; ++jj) s += R[i + m * jj] * ut[jj];
                ru += ut[i] * s;
            }
        }
        const double ct = pend ? 0.5 * (qx + ru) : 0.5 * qx + 0.5 * ru;
        c[t] = ct;
        acc += ct;
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) a.csum[rho] = acc;
}

template <int KIND, int NS, int MS, bool FUSE>
int launch_dpp(ddp_handle h, const FDArgs &a)
{
    const int key = (a.has_policy ? 2 : 0) | (a.has_lims ? 1 : 0);
    const long total = (long)a.B * a.nalpha;
    const int gpw = DDP_WAVE / 16;
    const dim3 grid((unsigned)((total + gpw - 1) / gpw)), block(DDP_WAVE);
    const char *fv = ddp_env(h, ENV_FORWARD_FAST);                // 0: the variant with the run-time dyn_tv test and masked stores (A/B, tests)
    const bool al16 = MS != 2 || ((((uintptr_t)a.u | (uintptr_t)a.k | (uintptr_t)a.K) & 15) == 0);     // 16-byte loads of ū_i, k_i, K_i[:, j]
    if (a.has_policy && !a.dyn_tv && a.sink && al16 && !(fv && fv[0] == '0')) {
        if (a.has_lims) hipLaunchKernelGGL((forward_dpp_kernel<KIND, NS, MS, true, true, FUSE, true>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((forward_dpp_kernel<KIND, NS, MS, true, false, FUSE, true>), grid, block, 0, h->stream, a);
        DDP_HIP(hipGetLastError());
        return 0;
    }
    switch (key) {
    case 0: hipLaunchKernelGGL((forward_dpp_kernel<KIND, NS, MS, false, false, FUSE>), grid, block, 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((forward_dpp_kernel<KIND, NS, MS, false, true, FUSE>), grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((forward_dpp_kernel<KIND, NS, MS, true, false, FUSE>), grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL((forward_dpp_kernel<KIND, NS, MS, true, true, FUSE>), grid, block, 0, h->stream, a); break;
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

template <bool FUSE>
int launch_pend_row(ddp_handle h, const FDArgs &a0)
{
    FDArgs a = a0;
    const long total = (long)a.B * a.nalpha;
    // 16-step chunks through LDS from 3 584 rollouts on: below that the element-wise streams keep up with the 208 ns step (0.133 ms at
    // 2 048 rollouts of N = 600 against 0.142 chunked: ~15 ticks of chunk bookkeeping per step), above it they are the bound (4 096:
    // 0.189 -> 0.147 ms).  DDP_PEND_CHUNK=0 / 1: never / always (A/B, tests).
    const char *pc = ddp_env(h, ENV_PEND_CHUNK);
    a.chunked = pc ? (pc[0] != '0') : (total >= 3584);
    const dim3 grid((unsigned)((total + 3) / 4)), block(DDP_WAVE);
    const int key = (a.has_policy ? 2 : 0) | (a.has_lims ? 1 : 0);
    if (a.wrap != 0 && a.has_policy) {                          // diff_fun with wrapped coordinates (only a policy has a difference to wrap)
        if (a.has_lims) hipLaunchKernelGGL((forward_pend_row_kernel<true, true, FUSE, true>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((forward_pend_row_kernel<true, false, FUSE, true>), grid, block, 0, h->stream, a);
        DDP_HIP(hipGetLastError());
        return 0;
    }
    switch (key) {
    case 0: hipLaunchKernelGGL((forward_pend_row_kernel<false, false, FUSE>), grid, block, 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((forward_pend_row_kernel<false, true, FUSE>), grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((forward_pend_row_kernel<true, false, FUSE>), grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL((forward_pend_row_kernel<true, true, FUSE>), grid, block, 0, h->stream, a); break;
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// returns 1 when the shape has no DPP kernel (caller falls back to the group kernel), 0 launched, <0 error
int ddp_launch_forward_dpp(ddp_handle h, const ddp_problem *p, const double *K, const double *k, const double *x0,
                           const double *u, const double *x, const double *alpha, int nalpha, const double *lims,
                           const int32_t *active, double *xnew, double *unew, double *cnew, double *csum)
{
    const bool lq = p->kind == DDP_PROBLEM_LQ;
    if (!((lq && p->n == 10 && p->m == 2) || (p->kind == DDP_PROBLEM_PENDCART))) return 1;
    FDArgs a;
    a.N = p->N; a.B = p->B; a.nalpha = nalpha;
    a.dyn_tv = p->dyn_tv; a.dyn_batched = p->dyn_batched; a.has_policy = K != nullptr; a.has_lims = lims != nullptr;
    a.A = p->A; a.Bm = p->Bm; a.K = K; a.k = k; a.x0 = x0; a.u = u; a.x = x; a.lims = lims; a.active = active;
    for (int i = 0; i < 16; ++i) a.alpha[i] = i < nalpha ? alpha[i] : 0.0;
    a.g = p->g; a.l = p->l; a.h = p->h; a.d = p->d;
    a.xnew = xnew; a.unew = unew; a.sink = (double *)h->sink; a.wrap = p->diff_wrap; a.chunked = 0;
    // wrapped differences exist in the pendulum's own kernels only (row and lane); everything else goes to the run-time-sized kernel
    if (p->diff_wrap != 0 && (lq || !a.sink || (ddp_env(h, ENV_FORWARD_PEND) && ddp_env(h, ENV_FORWARD_PEND)[0] == '0'))) return 1;
    const char *fuse_env = ddp_env(h, ENV_FORWARD_FUSE);           // 0: keep the separate cost kernel (A/B timing, tests)
    const bool fuse = p->cost_diag != 0 && !(fuse_env && fuse_env[0] == '0');     // Q, R declared diagonal: cost inside the rollout kernel
    a.Q = p->Q; a.R = p->R; a.cnew = cnew; a.csum = csum;
    for (int i = 0; i < 4; ++i) a.goal[i] = p->goal[i];
    int rc;
    const char *lane_env = ddp_env(h, ENV_FORWARD_LANE);          // 1 / 0 forces the lane-per-rollout pendcart kernel on / off
    const long total = (long)p->B * nalpha;
    const bool lane = !lq && (lane_env ? lane_env[0] == '1' : total >= 3L * 4096);       // at least ~3 rollouts per lane of a row kernel wave
    if (lane) {
        const dim3 grid((unsigned)((total + DDP_WAVE - 1) / DDP_WAVE)), block(DDP_WAVE);
        const int key = (a.has_policy ? 2 : 0) | (a.has_lims ? 1 : 0);
#define DDP_LANE(P_, L_)                                                                                            \
    do {                                                                                                              \
        if (fuse) hipLaunchKernelGGL((forward_lane_pendcart_kernel<P_, L_, true>), grid, block, 0, h->stream, a);     \
        else hipLaunchKernelGGL((forward_lane_pendcart_kernel<P_, L_, false>), grid, block, 0, h->stream, a);         \
    } while (0)
        if (a.wrap != 0 && a.has_policy) {                          // diff_fun with wrapped coordinates
            if (a.has_lims && fuse) hipLaunchKernelGGL((forward_lane_pendcart_kernel<true, true, true, true>), grid, block, 0, h->stream, a);
            else if (a.has_lims) hipLaunchKernelGGL((forward_lane_pendcart_kernel<true, true, false, true>), grid, block, 0, h->stream, a);
            else if (fuse) hipLaunchKernelGGL((forward_lane_pendcart_kernel<true, false, true, true>), grid, block, 0, h->stream, a);
            else hipLaunchKernelGGL((forward_lane_pendcart_kernel<true, false, false, true>), grid, block, 0, h->stream, a);
        } else
        switch (key) {
        case 0: DDP_LANE(false, false); break;
        case 1: DDP_LANE(false, true); break;
        case 2: DDP_LANE(true, false); break;
        case 3: DDP_LANE(true, true); break;
        }
#undef DDP_LANE
        DDP_HIP(hipGetLastError());
        rc = 0;
    } else if (!lq && a.sink && !(ddp_env(h, ENV_FORWARD_PEND) && ddp_env(h, ENV_FORWARD_PEND)[0] == '0')) {
        // the pendulum's own row kernel (DDP_FORWARD_PEND=0: the generic row kernel, for A/B timing and the tests of both)
        rc = fuse ? launch_pend_row<true>(h, a) : launch_pend_row<false>(h, a);
    } else {
        if (fuse) rc = lq ? launch_dpp<DDP_PROBLEM_LQ, 10, 2, true>(h, a) : launch_dpp<DDP_PROBLEM_PENDCART, 4, 1, true>(h, a);
        else rc = lq ? launch_dpp<DDP_PROBLEM_LQ, 10, 2, false>(h, a) : launch_dpp<DDP_PROBLEM_PENDCART, 4, 1, false>(h, a);
    }
    if (rc) return rc;
    if (fuse) return 0;                                         // cnew, csum already written
    CostArgs c;
    c.kind = p->kind; c.n = p->n; c.m = p->m; c.N = p->N; c.B = p->B; c.nalpha = nalpha; c.Q = p->Q; c.R = p->R;
    c.active = active;
    for (int i = 0; i < 4; ++i) c.goal[i] = p->goal[i];
    c.xnew = xnew; c.unew = unew; c.cnew = cnew; c.csum = csum;
    const dim3 cgrid((unsigned)((long)p->B * nalpha)), cblock(DDP_WAVE);
    if (lq) hipLaunchKernelGGL((cost_kernel<DDP_PROBLEM_LQ, 10, 2>), cgrid, cblock, 0, h->stream, c);
    else hipLaunchKernelGGL((cost_kernel<DDP_PROBLEM_PENDCART, 4, 1>), cgrid, cblock, 0, h->stream, c);
    DDP_HIP(hipGetLastError());
    return 0;
}
