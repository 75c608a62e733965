// forward_pass_row.hip — the 16-lane-row rollout (forward_pass_dpp.hip: one DPP row per (trajectory, α), four rollouts per wavefront)
// for a RANGE of linear-quadratic shapes instead of (10, 2) only: src/forward_pass.jl:9-33 is size-generic, and before this file every
// other n <= 16 went to the group-of-lanes kernel of forward_pass.hip (LDS hand-off per step; n = 12, m = 3, N = 500, B = 2 048: 5.1 ms
// against 1.6 ms for the backward pass of the same problem).
//
// As in back_pass_row.hip the kernel is compiled for padded sizes NP (even) >= n, MP >= m, NP + MP <= 16 and runs any n <= NP, m <= MP:
// lane j < n holds x̂_j and row j of A and B (the padded lanes and columns carry exact zeros), the controls are formed redundantly by
// every lane of the row (u = ū + α k + K (x̂ - x): MP row sums of K[q, j]·dx_j, forward_pass.jl:17-19), lanes NP .. NP + m - 1 store them.
// Dynamics with run-time strides (0: time-invariant / shared), requested DA steps ahead; ū, k, K[:, j], x[j] D steps ahead.
// The per-step cost needs x'Qx with a full Q and is off the dependency chain: cost_row_kernel, one lane per time step, afterwards.
#include "ddp_internal.h"

namespace {

struct FRArgs {
    int n, m, N, B, nalpha, has_policy, has_lims;
    long A_t, A_b, B_t, B_b;                        // element strides of A, B per time step / per trajectory
    const double *A, *Bm, *K, *k, *x0, *u, *x, *lims, *Q, *R;
    const int32_t *active;
    double alpha[16];
    double *xnew, *unew, *cnew, *csum;
};

template <int L>
__device__ __forceinline__ void fmac_bc(double &acc, double src0, double src1)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src0), "v"(src1), "n"(L));
}
__device__ __forceinline__ void dpp_fence(double &v) { asm volatile("s_nop 1" : "+v"(v)); }

template <int NN, int L = 0>
struct RowDot {          // s += Σ_{l<NN} src[lane l] * w[l]   (two interleaved accumulators)
    static __device__ __forceinline__ void run(double &s0, double &s1, double src, const double (&w)[NN])
    {
        if constexpr (L < NN) {
            if constexpr (L % 2 == 0) fmac_bc<L>(s0, src, w[L]); else fmac_bc<L>(s1, src, w[L]);
            RowDot<NN, L + 1>::run(s0, s1, src, w);
        }
    }
};
template <int NN, int L = 0>
struct RowSum {          // s += Σ_{l<NN} src[lane l]
    static __device__ __forceinline__ void run(double &s0, double &s1, double src, double one)
    {
        if constexpr (L < NN) {
            if constexpr (L % 2 == 0) fmac_bc<L>(s0, src, one); else fmac_bc<L>(s1, src, one);
            RowSum<NN, L + 1>::run(s0, s1, src, one);
        }
    }
};

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }   // Base.clamp

template <int NP, int MP, bool POLICY, bool LIMS>
__global__ __launch_bounds__(DDP_WAVE) void forward_row_kernel(FRArgs a)
{
    constexpr int G = 16, GPW = DDP_WAVE / G, D = (NP >= 12 || MP > 2) ? 4 : 8, DA = 2;   // (ring depth: registers)
    static_assert(NP + MP <= G && NP % 2 == 0, "NP + MP lanes must fit one 16-lane DPP row");
    const int n = a.n, m = a.m, N = a.N, B = a.B;
    const int lane = threadIdx.x, grp = lane / G, j = lane % G;
    const long total = (long)B * a.nalpha;
    long lin = (long)blockIdx.x * GPW + grp;
    const bool valid = lin < total;
    if (!valid) lin = total - 1;
    const int b = (int)(lin / a.nalpha), ai = (int)(lin % a.nalpha);
    const bool act = valid && !(a.active && a.active[b] == 0);
    if (!__any(act)) return;
    const double alpha = a.alpha[ai];
    const bool inx = j < n, st_u = j >= NP && j - NP < m;
    const int jx = inx ? j : 0;
    const double zx = inx ? 1.0 : 0.0;

    const size_t nm = (size_t)n * m;
    const double *ug = a.u + (size_t)m * N * b;
    const double *xg = POLICY ? a.x + (size_t)n * N * b : nullptr;
    const double *Kg = POLICY ? a.K + nm * N * b : nullptr;
    const double *kg = POLICY ? a.k + (size_t)m * N * b : nullptr;
    double *xo = a.xnew + (size_t)n * N * ((size_t)b + (size_t)B * ai);
    double *uo = a.unew + (size_t)m * N * ((size_t)b + (size_t)B * ai);
    const double *Ab = a.A + a.A_b * b, *Bb = a.Bm + a.B_b * b;

    int lcl[NP], qcl[MP];                                       // clamped column indices: the loads of padded entries re-read a valid one
#pragma unroll
    for (int l = 0; l < NP; ++l) lcl[l] = l < n ? l : n - 1;
#pragma unroll
    for (int q = 0; q < MP; ++q) qcl[q] = q < m ? q : m - 1;
    struct Dyn { double A[NP], B[MP]; };                        // row j of A and B at one step (raw: masked when used)
    auto load_dyn = [&](int i, Dyn &d) {
#pragma unroll
        for (int l = 0; l < NP; ++l) d.A[l] = Ab[a.A_t * i + jx + (size_t)n * lcl[l]];
#pragma unroll
        for (int q = 0; q < MP; ++q) d.B[q] = Bb[a.B_t * i + jx + (size_t)n * qcl[q]];
    };
    double lo[MP], hi[MP];
#pragma unroll
    for (int q = 0; q < MP; ++q) { lo[q] = (LIMS && q < m) ? a.lims[q] : 0.0; hi[q] = (LIMS && q < m) ? a.lims[q + m] : 0.0; }
    double one = 1.0;
    asm volatile("" : "+v"(one));                               // DPP src1 must be a VGPR

    double xh = inx ? a.x0[(size_t)n * b + jx] : 0.0;           // x̂_j
    struct Ops { double u[MP], k[MP], K[MP], x; };              // ū_i, k_i (row-uniform), K_i[:, j], x_i[j]
    auto fetch = [&](int i, Ops &o) {
#pragma unroll
        for (int q = 0; q < MP; ++q) o.u[q] = ug[(size_t)m * i + qcl[q]];
        if (POLICY) {
#pragma unroll
            for (int q = 0; q < MP; ++q) { o.k[q] = kg[(size_t)m * i + qcl[q]]; o.K[q] = Kg[nm * i + qcl[q] + (size_t)m * jx]; }
            o.x = xg[(size_t)n * i + jx];
        }
    };
    double *const st_ptr = st_u ? uo + (j - NP) : xo + jx;
    const size_t st_stride = st_u ? (size_t)m : (size_t)n;
    const bool st_on = act && (inx || st_u);
    auto step = [&](int i, const Ops &o, const Dyn &dy, bool advance) {
        // ---- controls (forward_pass.jl:17-24): u = ū + α k + K (x̂ - x), clamp, NaN -> 0 (inside f: demo_linear.jl:36)
        double uu[MP];
        if (POLICY) {
            double pr[MP];
            const double dx = zx * (xh - o.x);
#pragma unroll
            for (int q = 0; q < MP; ++q) { pr[q] = o.K[q] * dx; dpp_fence(pr[q]); }
#pragma unroll
            for (int q = 0; q < MP; ++q) {
                double s0 = o.u[q] + o.k[q] * alpha, s1 = 0.0;            // unew .+= k*α, then .+= K*dx
                RowSum<NP>::run(s0, s1, pr[q], one);
                uu[q] = s0 + s1;
            }
        } else {
#pragma unroll
            for (int q = 0; q < MP; ++q) uu[q] = o.u[q];
        }
#pragma unroll
        for (int q = 0; q < MP; ++q) {
            if (LIMS) uu[q] = clampd(uu[q], lo[q], hi[q]);
            if (uu[q] != uu[q]) uu[q] = 0.0;
            if (q >= m) uu[q] = 0.0;
        }
        {
            double v = xh;
#pragma unroll
            for (int q = 0; q < MP; ++q) v = (j == NP + q) ? uu[q] : v;
            if (st_on) st_ptr[(size_t)i * st_stride] = v;
        }
        // ---- dynamics x̂⁺ = A x̂ + B u (f is also called at i == N in the reference, its result is discarded)
        if (advance) {
            double Arow[NP];
#pragma unroll
            for (int l = 0; l < NP; ++l) Arow[l] = (l < n) ? zx * dy.A[l] : 0.0;
            double s0 = 0.0, s1 = 0.0, t = 0.0;
            RowDot<NP>::run(s0, s1, xh, Arow);
#pragma unroll
            for (int q = 0; q < MP; ++q) t += (zx * dy.B[q]) * uu[q];
            xh = (s0 + s1) + t;                                          // A*x + B*u
            dpp_fence(xh);
        }
    };
    dpp_fence(xh);
    Ops ring[D];
    Dyn dring[DA];
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d < N ? d : N - 1, ring[d]);
#pragma unroll
    for (int d = 0; d < DA; ++d) load_dyn(d < N ? d : N - 1, dring[d]);
    static_assert(D % DA == 0, "ring slots are fixed registers of the unrolled loop");
    int i0 = 0;
    for (; i0 + 2 * D <= N; i0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            step(i0 + d, ring[d], dring[d % DA], true);
            fetch(i0 + d + D, ring[d]);
            load_dyn(i0 + d + DA, dring[d % DA]);
        }
    }
    for (; i0 < N; i0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = i0 + d;
            if (i < N) {
                step(i, ring[d], dring[d % DA], i < N - 1);
                fetch(i + D < N ? i + D : N - 1, ring[d]);
                load_dyn(i + DA < N ? i + DA : N - 1, dring[d % DA]);
            }
        }
    }
}

// cnew[t] = ½ x_t'Q x_t + ½ u_t'R u_t (demo_linear.jl:24-26 with a general Q, R) and its sum: one wave per rollout, one lane per time step
template <int NMAX, int MMAX>
__global__ __launch_bounds__(DDP_WAVE) void cost_row_kernel(FRArgs a)
{
    const int n = a.n, m = a.m, N = a.N, B = a.B;
    const long rho = blockIdx.x;
    const int b = (int)(rho % B);
    if (a.active && a.active[b] == 0) return;
    const int lane = threadIdx.x;
    const double *x = a.xnew + (size_t)n * N * rho, *u = a.unew + (size_t)m * N * rho;
    double *c = a.cnew + (size_t)N * rho;
    __shared__ double qr[NMAX * NMAX + MMAX * MMAX];
    for (int e = lane; e < NMAX * NMAX; e += DDP_WAVE) { const int i = e % NMAX, jj = e / NMAX; qr[e] = (i < n && jj < n) ? a.Q[i + n * jj] : 0.0; }
    for (int e = lane; e < MMAX * MMAX; e += DDP_WAVE) { const int i = e % MMAX, jj = e / MMAX; qr[NMAX * NMAX + e] = (i < m && jj < m) ? a.R[i + m * jj] : 0.0; }
    wave_sync();
    const double *Q = qr, *R = qr + NMAX * NMAX;
    double acc = 0.0;
    for (int t = lane; t < N; t += DDP_WAVE) {
        double xt[NMAX], ut[MMAX];
#pragma unroll
        for (int i = 0; i < NMAX; ++i) xt[i] = i < n ? x[(size_t)n * t + i] : 0.0;
#pragma unroll
        for (int i = 0; i < MMAX; ++i) ut[i] = i < m ? u[(size_t)m * t + i] : 0.0;
        double qx = 0.0, ru = 0.0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            double s = 0.0;
#pragma unroll
            for (int jj = 0; jj < NMAX; ++jj) s += Q[i + NMAX * jj] * xt[jj];
            qx += xt[i] * s;
        }
#pragma unroll
        for (int i = 0; i < MMAX; ++i) {
            double s = 0.0;
#pragma unroll
            for (int jj = 0; jj < MMAX; ++jj) s += R[i + MMAX * jj] * ut[jj];
            ru += ut[i] * s;
        }
        const double ct = 0.5 * qx + 0.5 * ru;
        c[t] = ct;
        acc += ct;
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) a.csum[rho] = acc;
}

template <int NP, int MP>
int launch_frow(ddp_handle h, const FRArgs &a)
{
    const long total = (long)a.B * a.nalpha;
    const dim3 grid((unsigned)((total + 3) / 4)), block(DDP_WAVE);
    const int key = (a.has_policy ? 2 : 0) | (a.has_lims ? 1 : 0);
    switch (key) {
    case 0: hipLaunchKernelGGL((forward_row_kernel<NP, MP, false, false>), grid, block, 0, h->stream, a); break;
    case 1: hipLaunchKernelGGL((forward_row_kernel<NP, MP, false, true>), grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL((forward_row_kernel<NP, MP, true, false>), grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL((forward_row_kernel<NP, MP, true, true>), grid, block, 0, h->stream, a); break;
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// LQ problems with n <= 14, m <= 4, n + m <= 16 minus the combinations the padded sizes do not hold; returns 1 when the shape (or the
// problem kind, or a wrapped diff_fun) is not handled here, 0 launched, < 0 error
int ddp_launch_forward_row(ddp_handle h, const ddp_problem *p, const double *K, const double *k, const double *x0,
                           const double *u, const double *x, const double *alpha, int nalpha, const double *lims,
                           const int32_t *active, double *xnew, double *unew, double *cnew, double *csum)
{
    const int n = p->n, m = p->m;
    if (p->kind != DDP_PROBLEM_LQ || p->diff_wrap != 0) return 1;
    if (n < 1 || m < 1 || m > 4 || n > 14 || (n > 12 && m > 2) || (n > 10 && m > 4)) return 1;
    const long N = p->N;
    FRArgs a;
    a.n = n; a.m = m; a.N = p->N; a.B = p->B; a.nalpha = nalpha; a.has_policy = K != nullptr; a.has_lims = lims != nullptr;
    const long nn = (long)n * n, nm = (long)n * m;
    a.A_t = p->dyn_tv ? nn : 0; a.A_b = p->dyn_batched ? nn * (p->dyn_tv ? N : 1) : 0;
    a.B_t = p->dyn_tv ? nm : 0; a.B_b = p->dyn_batched ? nm * (p->dyn_tv ? N : 1) : 0;
    a.A = p->A; a.Bm = p->Bm; a.K = K; a.k = k; a.x0 = x0; a.u = u; a.x = x; a.lims = lims; a.Q = p->Q; a.R = p->R; a.active = active;
    for (int i = 0; i < 16; ++i) a.alpha[i] = i < nalpha ? alpha[i] : 0.0;
    a.xnew = xnew; a.unew = unew; a.cnew = cnew; a.csum = csum;
    const int np = n <= 4 ? 4 : (n + 1) & ~1, mp = m <= 2 ? 2 : 4;
    int rc = 1;
#define FROW_CASE(NP_, MP_) if (np == NP_ && mp == MP_) rc = launch_frow<NP_, MP_>(h, a);
    FROW_CASE(4, 2) FROW_CASE(4, 4) FROW_CASE(6, 2) FROW_CASE(6, 4) FROW_CASE(8, 2) FROW_CASE(8, 4)
    FROW_CASE(10, 2) FROW_CASE(10, 4) FROW_CASE(12, 2) FROW_CASE(12, 4) FROW_CASE(14, 2)
#undef FROW_CASE
    if (rc) return rc;
    const dim3 cgrid((unsigned)((long)p->B * nalpha)), cblock(DDP_WAVE);
    if (n <= 8 && m <= 2) hipLaunchKernelGGL((cost_row_kernel<8, 2>), cgrid, cblock, 0, h->stream, a);
    else hipLaunchKernelGGL((cost_row_kernel<14, 4>), cgrid, cblock, 0, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}
