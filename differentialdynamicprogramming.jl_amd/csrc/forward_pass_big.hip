// forward_pass_big.hip — closed-loop rollout for large states (n <= 64, m <= 8): one wavefront per (trajectory, α)
// rollout, lane j holds x̂_j.  Same arithmetic as forward_pass.hip / src/forward_pass.jl:9-33 (LQ family).
// Per step: x̂ and dx go through LDS once, lanes a < m form the controls, then lane j forms row j of A x̂ + B u with
// the row of A streamed from global memory (coalesced across lanes for every column).  The per-step cost is
// evaluated afterwards by a (time x batch)-parallel kernel, as in forward_pass_dpp.hip.
#include "ddp_internal.h"

namespace {

struct FBArgs {
    int n, m, N, B, nalpha;
    int dyn_tv, dyn_batched, has_policy, has_lims;
    const double *A, *Bm, *Q, *R, *K, *k, *x0, *u, *x, *lims;
    const int32_t *active;
    double alpha[16];
    double *xnew, *unew, *cnew, *csum;
};

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

__global__ __launch_bounds__(DDP_WAVE) void forward_big_kernel(FBArgs a)
{
    const int n = a.n, m = a.m, N = a.N, B = a.B;
    const long rho = blockIdx.x;
    const int b = (int)(rho % B), ai = (int)(rho / B);
    if (a.active && a.active[b] == 0) return;
    const int j = threadIdx.x;
    const bool inx = j < n, inu = j < m;
    const int jx = inx ? j : 0, ju = inu ? j : 0;
    const double alpha = a.alpha[ai];
    __shared__ double xs[DDP_WAVE], dxs[DDP_WAVE], us[DDP_MAX_M];
    const size_t nn = (size_t)n * n, nm = (size_t)n * m;
    const double *ug = a.u + (size_t)m * N * b;
    const double *xg = a.has_policy ? a.x + (size_t)n * N * b : nullptr;
    const double *Kg = a.has_policy ? a.K + nm * N * b : nullptr;
    const double *kg = a.has_policy ? a.k + (size_t)m * N * b : nullptr;
    double *xo = a.xnew + (size_t)n * N * ((size_t)b + (size_t)B * ai);
    double *uo = a.unew + (size_t)m * N * ((size_t)b + (size_t)B * ai);
    const double *Ab = a.A + (a.dyn_batched ? nn * (a.dyn_tv ? N : 1) * b : 0);
    const double *Bb = a.Bm + (a.dyn_batched ? nm * (a.dyn_tv ? N : 1) * b : 0);
    const double lo = (a.has_lims && inu) ? a.lims[ju] : 0.0, hi = (a.has_lims && inu) ? a.lims[ju + m] : 0.0;

    // Σ_l w[l·stride]·v[l], l < n: EIGHT requests in flight per round trip (the loop with two accumulators of rounds 1-4 waited for
    // global memory n / 2 times per step: 25 µs per step at n = 48); the partial sums meet pairwise
    auto dot8 = [&](const double *w, size_t stride, const double *v) -> double {
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int l = 0;
        for (; l + 8 <= n; l += 8) {
            double wv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) wv[q] = w[stride * (l + q)];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += wv[q] * v[l + q];
        }
        for (; l < n; ++l) acc[l & 7] += w[stride * l] * v[l];
        return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    };
    double xh = inx ? a.x0[(size_t)n * b + jx] : 0.0;
    for (int i = 0; i < N; ++i) {
        xs[j] = xh;
        dxs[j] = a.has_policy ? xh - (inx ? xg[(size_t)n * i + jx] : 0.0) : 0.0;
        wave_sync();
        const double *Ai = Ab + (a.dyn_tv ? nn * i : 0), *Bi = Bb + (a.dyn_tv ? nm * i : 0);
        double ax = 0.0;
        if (i < N - 1) ax = dot8(Ai + jx, (size_t)n, xs);         // (A x̂)_j does not wait for the controls
        if (inu) {                                               // controls (forward_pass.jl:17-24)
            double v = ug[(size_t)m * i + ju];
            if (a.has_policy) {
                v += kg[(size_t)m * i + ju] * alpha;             // unew .+= k*α
                v += dot8(Kg + nm * i + ju, (size_t)m, dxs);      // unew .+= K*dx
            }
            if (a.has_lims) v = clampd(v, lo, hi);
            if (v != v) v = 0.0;                                 // u[isnan.(u)] .= 0 inside f
            us[ju] = v;
            uo[(size_t)m * i + ju] = v;
        }
        if (inx) xo[(size_t)n * i + jx] = xh;
        wave_sync();
        if (i < N - 1) {                                         // x+ = A x + B u (src/demo_linear.jl:42-46)
            double t = 0.0;
            for (int q = 0; q < m; ++q) t += Bi[jx + (size_t)n * q] * us[q];
            xh = inx ? ax + t : 0.0;
        }
        wave_sync();
    }
}

// run-time-sized cost: one wave per rollout, lanes over time; LQ family only (demo_linear.jl:49 split per step)
__global__ __launch_bounds__(DDP_WAVE) void cost_rt_kernel(FBArgs a)
{
    const int n = a.n, m = a.m, N = a.N, B = a.B;
    const long rho = blockIdx.x;
    const int b = (int)(rho % B);
    if (a.active && a.active[b] == 0) return;
    const int lane = threadIdx.x;
    const double *x = a.xnew + (size_t)n * N * rho, *u = a.unew + (size_t)m * N * rho;
    double *c = a.cnew + (size_t)N * rho;
    extern __shared__ double qr[];
    for (int e = lane; e < n * n; e += DDP_WAVE) qr[e] = a.Q[e];
    for (int e = lane; e < m * m; e += DDP_WAVE) qr[n * n + e] = a.R[e];
    wave_sync();
    const double *Q = qr, *R = qr + n * n;
    double acc = 0.0;
    for (int t = lane; t < N; t += DDP_WAVE) {
        const double *xt = x + (size_t)n * t, *ut = u + (size_t)m * t;
        double qx = 0.0, ru = 0.0;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int jj = 0; jj < n; ++jj) s += Q[i + n * jj] * xt[jj];
            qx += xt[i] * s;
        }
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int jj = 0; jj < m; ++jj) s += R[i + m * jj] * ut[jj];
            ru += ut[i] * s;
        }
        const double ct = 0.5 * qx + 0.5 * ru;
        c[t] = ct;
        acc += ct;
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) a.csum[rho] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// n = 64, m = 8 (BASELINE config 4): the rollout is a stream of 41 KB per step (A_i, B_i, K_i) behind a short
// dependent chain, so the kernel is organised around keeping one whole step of operands in flight per wave:
//   * 16-byte loads only: lane (h, j') with h = lane/32 holds rows 2j', 2j'+1 of the columns [36h, 36h+36) of
//     [A_i B_i] (36 loads), its share of K_i as 4 loads of two consecutive gain rows, and pairs of x, u, k;
//   * the operands of step i+1 are requested before step i is computed (two register buffers, loop unrolled by 2);
//   * x̂, dx and u go through LDS once per step; the two column halves are summed with one v_permlane32_swap per
//     dword, K·dx is reduced over the 16 lanes that share a gain-row pair with four xor-shuffles.
// Same arithmetic as forward_big_kernel (src/forward_pass.jl:9-33), sums in a different (fixed) order.
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double sum_halves(double v)      // v(lane) + v(lane ^ 32) in every lane
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u2v e = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);      // .x = lower half everywhere, .y = upper half
    const u2v f = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)f.x, (int)e.x) + __hiloint2double((int)f.y, (int)e.y);
}

struct FB64Buf { d2 ab[36], kr[4], xo, uo, ko; };

// NA line-search candidates (step sizes α) of ONE trajectory share a wave: A_i, B_i, K_i, x_i, u_i, k_i are the same for all of
// them, so the 41 KB per step are fetched once per NA rollouts (the full line search of an iLQG iteration evaluates 4-11 α).
template <bool POL, int NA>
__global__ __launch_bounds__(DDP_WAVE) void forward_big64_kernel(FBArgs a)
{
    constexpr int n = 64, m = 8, NC = 36;
    const int N = a.N, B = a.B;
    const int b = blockIdx.x, a0 = NA * blockIdx.y;              // candidates a0 .. a0+NA-1 (those >= nalpha compute, never store)
    if (a.active && a.active[b] == 0) return;
    const int lane = threadIdx.x, h = lane >> 5, r0 = 2 * (lane & 31), q0 = 2 * (lane & 3);
    double alpha[NA];
    bool live[NA];
#pragma unroll
    for (int c = 0; c < NA; ++c) { live[c] = a0 + c < a.nalpha; alpha[c] = a.alpha[live[c] ? a0 + c : a0]; }
    __shared__ __attribute__((aligned(16))) double zs[NA][n + m], dxs[NA][n];
    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m;
    const double *ug = a.u + (size_t)m * N * b;
    const double *xg = POL ? a.x + (size_t)n * N * b : nullptr;
    const double *Kg = POL ? a.K + nm * N * b : nullptr;
    const double *kg = POL ? a.k + (size_t)m * N * b : nullptr;
    double *xo[NA], *uo[NA];
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        const size_t rho = (size_t)b + (size_t)B * (live[c] ? a0 + c : a0);
        xo[c] = a.xnew + (size_t)n * N * rho;
        uo[c] = a.unew + (size_t)m * N * rho;
    }
    const double *Ab = a.A + (a.dyn_batched ? nn * (a.dyn_tv ? N : 1) * b : 0);
    const double *Bb = a.Bm + (a.dyn_batched ? nm * (a.dyn_tv ? N : 1) * b : 0);
    const bool tv = a.dyn_tv, lims = a.has_lims;
    const double lo0 = lims ? a.lims[q0] : 0.0, hi0 = lims ? a.lims[q0 + m] : 0.0;
    const double lo1 = lims ? a.lims[q0 + 1] : 0.0, hi1 = lims ? a.lims[q0 + 1 + m] : 0.0;

    auto fetch = [&](int i, FB64Buf &f) {
        const double *Ai = Ab + (tv ? nn * i : 0), *Bi = Bb + (tv ? nm * i : 0);
        const double *pa = Ai + n * NC * h + r0;                      // column 36h of A, my row pair
        const double *pb = h ? Bi + r0 - n * (n - NC) : Ai + r0;      // t >= 28: half 1 continues in B (column 36+t-64), half 0 in A (column t)
#pragma unroll
        for (int t = 0; t < NC; ++t) f.ab[t] = *(const d2 *)((t < n - NC ? pa : pb) + n * t);
        if (POL) {
#pragma unroll
            for (int t = 0; t < 4; ++t) f.kr[t] = *(const d2 *)(Kg + nm * i + 2 * lane + 2 * DDP_WAVE * t);
            f.xo = *(const d2 *)(xg + (size_t)n * i + r0);
            f.ko = *(const d2 *)(kg + (size_t)m * i + q0);
        }
        f.uo = *(const d2 *)(ug + (size_t)m * i + q0);
    };
    double xa[NA], xb[NA];
#pragma unroll
    for (int c = 0; c < NA; ++c) { xa[c] = a.x0[(size_t)n * b + r0]; xb[c] = a.x0[(size_t)n * b + r0 + 1]; }
    auto step = [&](int i, const FB64Buf &f) {
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const d2 xh = d2{xa[c], xb[c]};
            if (h == 0) {
                *(d2 *)(zs[c] + r0) = xh;
                if (POL) *(d2 *)(dxs[c] + r0) = xh - f.xo;
                if (live[c]) *(d2 *)(xo[c] + (size_t)n * i + r0) = xh;
            }
        }
        wave_sync();
        // x part of A x̂ + B u: columns [0,28) of my half are state columns for both halves
        double sa0[NA], sa1[NA], sb0[NA], sb1[NA];
#pragma unroll
        for (int c = 0; c < NA; ++c) { sa0[c] = 0.0; sa1[c] = 0.0; sb0[c] = 0.0; sb1[c] = 0.0; }
        if (i < N - 1) {
#pragma unroll
            for (int t = 0; t < n - NC; t += 2) {
#pragma unroll
                for (int c = 0; c < NA; ++c) {
                    const d2 z = *(const d2 *)(zs[c] + NC * h + t);
                    sa0[c] += f.ab[t].x * z.x; sb0[c] += f.ab[t].y * z.x;
                    sa1[c] += f.ab[t + 1].x * z.y; sb1[c] += f.ab[t + 1].y * z.y;
                }
            }
        }
        // controls (forward_pass.jl:17-24): gain rows q0, q0+1, state columns (lane>>2) + 16t
        double v0[NA], v1[NA];
#pragma unroll
        for (int c = 0; c < NA; ++c) { v0[c] = f.uo.x; v1[c] = f.uo.y; }
        if (POL) {
            double p0[NA], p1[NA];
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                p0[c] = 0.0; p1[c] = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) { const double d = dxs[c][(lane >> 2) + 16 * t]; p0[c] += f.kr[t].x * d; p1[c] += f.kr[t].y * d; }
            }
#pragma unroll
            for (int off = 4; off < DDP_WAVE; off <<= 1) {
#pragma unroll
                for (int c = 0; c < NA; ++c) { p0[c] += __shfl_xor(p0[c], off, DDP_WAVE); p1[c] += __shfl_xor(p1[c], off, DDP_WAVE); }
            }
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                v0[c] += f.ko.x * alpha[c]; v1[c] += f.ko.y * alpha[c];   // unew .+= k*α
                v0[c] += p0[c]; v1[c] += p1[c];                           // unew .+= K*dx
            }
        }
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            if (lims) { v0[c] = clampd(v0[c], lo0, hi0); v1[c] = clampd(v1[c], lo1, hi1); }
            if (v0[c] != v0[c]) v0[c] = 0.0;                              // u[isnan.(u)] .= 0 inside f
            if (v1[c] != v1[c]) v1[c] = 0.0;
            if (lane < 4) {
                *(d2 *)(zs[c] + n + q0) = d2{v0[c], v1[c]};
                if (live[c]) *(d2 *)(uo[c] + (size_t)m * i + q0) = d2{v0[c], v1[c]};
            }
        }
        wave_sync();
        if (i < N - 1) {                                             // the last 8 columns of each half: x̂[28..36) | u
#pragma unroll
            for (int t = n - NC; t < NC; t += 2) {
#pragma unroll
                for (int c = 0; c < NA; ++c) {
                    const d2 z = *(const d2 *)(zs[c] + NC * h + t);
                    sa0[c] += f.ab[t].x * z.x; sb0[c] += f.ab[t].y * z.x;
                    sa1[c] += f.ab[t + 1].x * z.y; sb1[c] += f.ab[t + 1].y * z.y;
                }
            }
#pragma unroll
            for (int c = 0; c < NA; ++c) { xa[c] = sum_halves(sa0[c] + sa1[c]); xb[c] = sum_halves(sb0[c] + sb1[c]); }
        }
        wave_sync();
    };
    FB64Buf f0, f1;
    fetch(0, f0);
    int i = 0;
    for (; i + 1 < N; i += 2) {
        fetch(i + 1, f1);
        step(i, f0);
        fetch(i + 2 < N ? i + 2 : i + 1, f0);                        // (re-reads the last step at the end: valid memory, unused)
        step(i + 1, f1);
    }
    if (i < N) step(i, f0);
}

// cost of the n = 64, m = 8 rollouts: one wave per (rollout, 16 time steps); lane j keeps row j of Q in registers,
// x̂_l comes from v_readlane as a scalar operand, the 64 products x_j·(Qx)_j of a time step are summed through LDS.
__global__ __launch_bounds__(DDP_WAVE) void cost_big64_kernel(FBArgs a)
{
    constexpr int n = 64, m = 8, TB = 16;
    const int N = a.N, B = a.B;
    const long rho = blockIdx.x;
    const int b = (int)(rho % B), t0 = blockIdx.y * TB;
    if (a.active && a.active[b] == 0) return;
    const int lane = threadIdx.x;
    const double *x = a.xnew + (size_t)n * N * rho, *u = a.unew + (size_t)m * N * rho;
    __shared__ double prod[TB][n + 1];
    double q[n];
#pragma unroll
    for (int l = 0; l < n; ++l) q[l] = a.Q[lane + n * l];
    const int nt = min(TB, N - t0);
    for (int t = 0; t < nt; ++t) {
        const double xj = x[(size_t)n * (t0 + t) + lane];
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int l = 0; l < n; l += 2) {
            s0 += q[l] * __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xj), l), __builtin_amdgcn_readlane(__double2loint(xj), l));
            s1 += q[l + 1] * __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xj), l + 1), __builtin_amdgcn_readlane(__double2loint(xj), l + 1));
        }
        prod[t][lane] = xj * (s0 + s1);
    }
    wave_sync();
    if (lane < nt) {
        double qx = 0.0, ru = 0.0;
        for (int jj = 0; jj < n; ++jj) qx += prod[lane][jj];
        const double *ut = u + (size_t)m * (t0 + lane);
        double uu[m];
#pragma unroll
        for (int i = 0; i < m; ++i) uu[i] = ut[i];
#pragma unroll
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
#pragma unroll
            for (int jj = 0; jj < m; ++jj) s += a.R[i + m * jj] * uu[jj];
            ru += uu[i] * s;
        }
        a.cnew[(size_t)N * rho + t0 + lane] = 0.5 * qx + 0.5 * ru;
    }
}

// csum[rho] = sum_t cnew[t, rho] in a fixed order (one wave per rollout)
__global__ __launch_bounds__(DDP_WAVE) void cost_sum_kernel(FBArgs a)
{
    const int N = a.N;
    const long rho = blockIdx.x;
    if (a.active && a.active[(int)(rho % a.B)] == 0) return;
    const int lane = threadIdx.x;
    double acc = 0.0;
    for (int t = lane; t < N; t += DDP_WAVE) acc += a.cnew[(size_t)N * rho + t];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, DDP_WAVE);
    if (lane == 0) a.csum[rho] = acc;
}


// ---------------------------------------------------------------------------------------------------------------
// 14 < n <= 32, m <= 8 (what no 16-lane row holds): one wave per rollout like forward_big_kernel, organised like the n = 64 kernel —
// every operand of step i+1 (A_i, B_i, K_i, x_i, ū_i, k_i) is requested before step i is computed (two register buffers, the loop
// unrolled by 2), so a step no longer waits for n dependent global loads (forward_big_kernel at n = 24, m = 4, N = 300, B = 1 024:
// 2.65 ms, 8.8 µs per step).  Lane (j, h) = (lane & 31, lane >> 5) holds row j of the columns [h·CA, h·CA + CA) of A_i and
// [4h, 4h + 4) of B_i (n <= 2·CA: CA = 8, 12, 16); the column halves meet in one v_permlane32_swap per dword.  For K_i·dx the same
// lanes are (a, p) = (lane >> 3, lane & 7): gain row a, state columns p + 8q, summed over p by three DPP moves.  x̂, dx and u go
// through zero-padded LDS vectors, so columns past n / m need no select: their (clamped, finite) operands meet a zero.
// Limits are data (±inf without them: Base.clamp is then the identity); the empty policy skips the gain part (uniform branch).
// Same arithmetic as forward_big_kernel (src/forward_pass.jl:9-33), the row sums of A x̂ + B u in two halves.
#ifndef FM_DEPTH
#define FM_DEPTH 2
#endif
template <int CA>
struct FMBuf { double fa[CA], fb[4], kq[CA / 4], xo, uo, ko; };

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int CA>
__global__ __launch_bounds__(DDP_WAVE) void forward_mid_kernel(FBArgs a)
{
    constexpr int CB = 4, KQ = CA / 4, NP = 2 * CA;
    const int n = a.n, m = a.m, N = a.N, B = a.B;
    const long rho = blockIdx.x;
    const int b = (int)(rho % B), ai = (int)(rho / B);
    if (a.active && a.active[b] == 0) return;
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5, ga = lane >> 3, gp = lane & 7;
    const bool inx = j < n, stx = inx && h == 0, stu = ga < m && gp == 0, pol = a.has_policy != 0;
    const int jc = inx ? j : n - 1, ac = ga < m ? ga : m - 1;
    const double alpha = a.alpha[ai];
    __shared__ __attribute__((aligned(16))) double xs[NP], dxs[NP], us[2 * CB];
    const size_t nn = (size_t)n * n, nm = (size_t)n * m;
    const char *ug = (const char *)(a.u + (size_t)m * N * b);
    // without a policy the gain / nominal-state / k requests aim at u_0 with stride 0 (their values are never used): NO branch around a
    // load — under a run-time condition the compiler cannot count what is outstanding and waits for everything (s_waitcnt vmcnt(0)) at
    // the first use, i.e. for the requests of two steps ahead it has just issued
    const char *xg = pol ? (const char *)(a.x + (size_t)n * N * b) : ug;
    const char *Kg = pol ? (const char *)(a.K + nm * N * b) : ug;
    const char *kg = pol ? (const char *)(a.k + (size_t)m * N * b) : ug;
    const size_t sK = pol ? nm * 8 : 0, sx = pol ? (size_t)n * 8 : 0, sk = pol ? (size_t)m * 8 : 0;
    char *xo = (char *)(a.xnew + (size_t)n * N * ((size_t)b + (size_t)B * ai));
    char *uo = (char *)(a.unew + (size_t)m * N * ((size_t)b + (size_t)B * ai));
    const char *Ab = (const char *)(a.A + (a.dyn_batched ? nn * (a.dyn_tv ? N : 1) * b : 0));
    const char *Bb = (const char *)(a.Bm + (a.dyn_batched ? nm * (a.dyn_tv ? N : 1) * b : 0));
    const size_t sA = a.dyn_tv ? nn * 8 : 0, sB = a.dyn_tv ? nm * 8 : 0;
    const double inf = __builtin_huge_val();
    const double lo = a.has_lims ? a.lims[ac] : -inf, hi = a.has_lims ? a.lims[ac + m] : inf;
    // fixed per-lane byte offsets inside a step's block (columns past the end repeat the last one: they meet zeros of xs / us / dxs)
    unsigned offA[CA], offB[CB], offK[KQ];
#pragma unroll
    for (int c = 0; c < CA; ++c) { const int l = h * CA + c; offA[c] = 8u * (unsigned)(jc + n * (l < n ? l : n - 1)); }
#pragma unroll
    for (int c = 0; c < CB; ++c) { const int q = h * CB + c; offB[c] = 8u * (unsigned)(jc + n * (q < m ? q : m - 1)); }
#pragma unroll
    for (int q = 0; q < KQ; ++q) { const int l = gp + 8 * q; offK[q] = pol ? 8u * (unsigned)(ac + m * (l < n ? l : n - 1)) : 0u; }
    const unsigned offx = 8u * (unsigned)jc, offu = 8u * (unsigned)ac, offxp = pol ? offx : 0u, offkp = pol ? offu : 0u;
    const int ifl = N >= 2 ? N - 2 : 0;                               // the last step whose A_i, B_i are used (and certainly exist)

    auto fetch = [&](int i, FMBuf<CA> &f) {
        const int fi = i < ifl ? i : ifl;
        const char *Ai = Ab + sA * fi, *Bi = Bb + sB * fi;
#pragma unroll
        for (int c = 0; c < CA; ++c) f.fa[c] = *(const double *)(Ai + offA[c]);
#pragma unroll
        for (int c = 0; c < CB; ++c) f.fb[c] = *(const double *)(Bi + offB[c]);
        const char *Ki = Kg + sK * i;
#pragma unroll
        for (int q = 0; q < KQ; ++q) f.kq[q] = *(const double *)(Ki + offK[q]);
        f.xo = *(const double *)(xg + sx * i + offxp);
        f.ko = *(const double *)(kg + sk * i + offkp);
        f.uo = *(const double *)(ug + (size_t)m * 8 * i + offu);
    };
    if (lane < NP) { xs[lane] = 0.0; dxs[lane] = 0.0; }
    if (lane < 2 * CB) us[lane] = 0.0;
    double xh = inx ? a.x0[(size_t)n * b + jc] : 0.0;
    wave_sync();
    auto step = [&](int i, const FMBuf<CA> &f) {
        if (stx) {
            xs[j] = xh;
            if (pol) dxs[j] = xh - f.xo;
            *(double *)(xo + (size_t)n * 8 * i + offx) = xh;
        }
        wave_sync();
        // state part of A x̂ + B u (does not wait for the controls)
        double s0 = 0.0, s1 = 0.0;
        if (i < N - 1) {
#pragma unroll
            for (int c = 0; c < CA; c += 2) {
                const d2 z = *(const d2 *)(xs + h * CA + c);
                s0 += f.fa[c] * z.x; s1 += f.fa[c + 1] * z.y;
            }
        }
        // controls (forward_pass.jl:17-24)
        double v = f.uo;
        if (pol) {
            double p0 = 0.0;
#pragma unroll
            for (int q = 0; q < KQ; ++q) p0 += f.kq[q] * dxs[gp + 8 * q];
            p0 += dpp_f64<0xB1>(p0);                                  // quad_perm [1,0,3,2]
            p0 += dpp_f64<0x4E>(p0);                                  // quad_perm [2,3,0,1]
            p0 += dpp_f64<0x141>(p0);                                 // row_half_mirror: the other quad of the 8 lanes
            v += f.ko * alpha;                                       // unew .+= k*α
            v += p0;                                                 // unew .+= K*dx
        }
        v = clampd(v, lo, hi);
        if (v != v) v = 0.0;                                         // u[isnan.(u)] .= 0 inside f
        if (stu) {
            us[ga] = v;
            *(double *)(uo + (size_t)m * 8 * i + offu) = v;
        }
        wave_sync();
        if (i < N - 1) {
#pragma unroll
            for (int c = 0; c < CB; c += 2) {
                const d2 z = *(const d2 *)(us + h * CB + c);
                s0 += f.fb[c] * z.x; s1 += f.fb[c + 1] * z.y;
            }
            const double t = sum_halves(s0 + s1);
            xh = inx ? t : 0.0;
        }
        wave_sync();
    };
    const int il = N - 1;
#if FM_DEPTH == 2
    // operands two steps ahead (three register buffers, the loop unrolled by 3): one step ahead the wave waited for HBM every step
    FMBuf<CA> f0, f1, f2;
    fetch(0, f0);
    fetch(1 < il ? 1 : il, f1);
    int i = 0;
    for (; i + 2 < N; i += 3) {
        fetch(i + 2, f2);
        step(i, f0);
        fetch(i + 3 < il ? i + 3 : il, f0);                          // (re-reads the last step at the end: valid memory, unused)
        step(i + 1, f1);
        fetch(i + 4 < il ? i + 4 : il, f1);
        step(i + 2, f2);
    }
    if (i < N) step(i, f0);
    if (i + 1 < N) step(i + 1, f1);
#else
    FMBuf<CA> f0, f1;
    fetch(0, f0);
    int i = 0;
    for (; i + 1 < N; i += 2) {
        fetch(i + 1, f1);
        step(i, f0);
        fetch(i + 2 < il ? i + 2 : il, f0);
        step(i + 1, f1);
    }
    if (i < N) step(i, f0);
#endif
}

// cost of those rollouts (LQ family, full Q and R; demo_linear.jl:49 split per step): one wave per (rollout, 64 time steps), lane =
// time step; the x̂ and u of the 64 steps arrive by coalesced loads into LDS tiles with odd row pitches; the lane's x̂ sits in NPC >= n
// registers (zeros past n), Q' zero-padded to NPC x NPC in the LDS so that a row of it is NPC / 2 16-byte broadcast reads — with
// run-time loop bounds the kernel spent two LDS reads and a loop iteration per product (0.11 ms at n = 24, N = 300, B = 1 024).
// Sums in cost_rt_kernel's order (the padding adds zeros); the sum over time is cost_sum_kernel's.
template <int NPC>
__global__ __launch_bounds__(DDP_WAVE) void cost_mid_kernel(FBArgs a)
{
    constexpr int TB = DDP_WAVE, PX = NPC + 1;
    const int n = a.n, m = a.m, N = a.N, B = a.B;
    const long rho = blockIdx.x;
    const int b = (int)(rho % B), t0 = blockIdx.y * TB;
    if (a.active && a.active[b] == 0) return;
    const int lane = threadIdx.x;
    const int nt = min(TB, N - t0), pm = m | 1;
    __shared__ __attribute__((aligned(16))) double qt[NPC * NPC], xt[TB * PX], ut[TB * 9], rr[8 * 8];
    const double *x = a.xnew + (size_t)n * N * rho + (size_t)n * t0, *u = a.unew + (size_t)m * N * rho + (size_t)m * t0;
    for (int e = lane; e < TB * PX; e += DDP_WAVE) xt[e] = 0.0;
    for (int e = lane; e < NPC * NPC; e += DDP_WAVE) { const int jj = e % NPC, i = e / NPC; qt[e] = (i < n && jj < n) ? a.Q[i + n * jj] : 0.0; }   // qt[jj + NPC i] = Q[i, jj]
    for (int e = lane; e < m * m; e += DDP_WAVE) rr[e] = a.R[e];
    wave_sync();
    for (int e = lane; e < n * nt; e += DDP_WAVE) xt[(e / n) * PX + e % n] = x[e];
    for (int e = lane; e < m * nt; e += DDP_WAVE) ut[(e / m) * pm + e % m] = u[e];
    wave_sync();
    if (lane >= nt) return;
    double xr[NPC];
#pragma unroll
    for (int jj = 0; jj < NPC; ++jj) xr[jj] = xt[lane * PX + jj];
    double qx = 0.0, ru = 0.0;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        double s = 0.0;
#pragma unroll
        for (int jj = 0; jj < NPC; jj += 2) {
            const d2 q2 = *(const d2 *)(qt + NPC * i + jj);
            s += q2.x * xr[jj];
            s += q2.y * xr[jj + 1];
        }
        qx += xr[i] * s;
        asm volatile("" : "+v"(qx) :: "memory");                   // (row by row: with every read of Q' hoisted to the front a few hundred registers spilled)
    }
    const double *ul = ut + lane * pm;
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int jj = 0; jj < m; ++jj) s += rr[i + m * jj] * ul[jj];
        ru += ul[i] * s;
    }
    a.cnew[(size_t)N * rho + t0 + lane] = 0.5 * qx + 0.5 * ru;
}

}   // namespace

// returns 1 when not applicable (caller falls back), 0 launched, <0 error
int ddp_launch_forward_big(ddp_handle h, const ddp_problem *p, const double *K, const double *k, const double *x0,
                           const double *u, const double *x, const double *alpha, int nalpha, const double *lims,
                           const int32_t *active, double *xnew, double *unew, double *cnew, double *csum)
{
    if (p->kind != DDP_PROBLEM_LQ || p->n > 64 || p->m > DDP_MAX_M) return 1;
    h->last_kernel[1] = "forward_big_kernel";                         // (the mid-size branch below renames it)
    FBArgs a;
    a.n = p->n; a.m = p->m; a.N = p->N; a.B = p->B; a.nalpha = nalpha;
    a.dyn_tv = p->dyn_tv; a.dyn_batched = p->dyn_batched; a.has_policy = K != nullptr; a.has_lims = lims != nullptr;
    a.A = p->A; a.Bm = p->Bm; a.Q = p->Q; a.R = p->R; a.K = K; a.k = k; a.x0 = x0; a.u = u; a.x = x; a.lims = lims;
    a.active = active;
    for (int i = 0; i < 16; ++i) a.alpha[i] = i < nalpha ? alpha[i] : 0.0;
    a.xnew = xnew; a.unew = unew; a.cnew = cnew; a.csum = csum;
    const dim3 grid((unsigned)((long)p->B * nalpha)), block(DDP_WAVE);
    const char *env = ddp_env(h, ENV_FORWARD64);                       // DDP_FORWARD64=0: run-time-sized kernels also at n = 64, m = 8
    if (p->n == 64 && p->m == 8 && !(env && env[0] == '0')) {
        // up to 4 step sizes of a trajectory per wave (operands fetched once); a single α keeps the one-rollout instantiation
        const int na = nalpha >= 3 ? 4 : (nalpha == 2 ? 2 : 1);
        const dim3 g64((unsigned)p->B, (unsigned)((nalpha + na - 1) / na));
        if (a.has_policy) {
            if (na == 4) hipLaunchKernelGGL((forward_big64_kernel<true, 4>), g64, block, 0, h->stream, a);
            else if (na == 2) hipLaunchKernelGGL((forward_big64_kernel<true, 2>), g64, block, 0, h->stream, a);
            else hipLaunchKernelGGL((forward_big64_kernel<true, 1>), g64, block, 0, h->stream, a);
        } else {
            hipLaunchKernelGGL((forward_big64_kernel<false, 1>), dim3((unsigned)p->B, (unsigned)nalpha), block, 0, h->stream, a);
        }
        hipLaunchKernelGGL(cost_big64_kernel, dim3(grid.x, (unsigned)((p->N + 15) / 16)), block, 0, h->stream, a);
        hipLaunchKernelGGL(cost_sum_kernel, grid, block, 0, h->stream, a);
        DDP_HIP(hipGetLastError());
        return 0;
    }
    const char *mid = ddp_env(h, ENV_FORWARD_MID);                     // DDP_FORWARD_MID=0: the run-time-sized kernels below n = 64 too (A/B, tests)
    if (p->n <= 32 && p->m <= 8 && !(mid && mid[0] == '0')) {
        if (p->n <= 16) hipLaunchKernelGGL((forward_mid_kernel<8>), grid, block, 0, h->stream, a);
        else if (p->n <= 24) hipLaunchKernelGGL((forward_mid_kernel<12>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((forward_mid_kernel<16>), grid, block, 0, h->stream, a);
        const dim3 cgrid(grid.x, (unsigned)((p->N + DDP_WAVE - 1) / DDP_WAVE));
        if (p->n <= 16) hipLaunchKernelGGL((cost_mid_kernel<16>), cgrid, block, 0, h->stream, a);
        else if (p->n <= 24) hipLaunchKernelGGL((cost_mid_kernel<24>), cgrid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((cost_mid_kernel<32>), cgrid, block, 0, h->stream, a);
        hipLaunchKernelGGL(cost_sum_kernel, grid, block, 0, h->stream, a);
        DDP_HIP(hipGetLastError());
        h->last_kernel[1] = "forward_mid_kernel";
        return 0;
    }
    hipLaunchKernelGGL(forward_big_kernel, grid, block, 0, h->stream, a);
    if (p->n > 32 && !(mid && mid[0] == '0')) {                        // the cost kernel of the mid-size rollouts at the larger paddings
        const dim3 cgrid(grid.x, (unsigned)((p->N + DDP_WAVE - 1) / DDP_WAVE));
        if (p->n <= 48) hipLaunchKernelGGL((cost_mid_kernel<48>), cgrid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((cost_mid_kernel<64>), cgrid, block, 0, h->stream, a);
        hipLaunchKernelGGL(cost_sum_kernel, grid, block, 0, h->stream, a);
        DDP_HIP(hipGetLastError());
        return 0;
    }
    const size_t shmem = ((size_t)p->n * p->n + (size_t)p->m * p->m) * sizeof(double);
    hipLaunchKernelGGL(cost_rt_kernel, grid, block, shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}
