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

    double xh = inx ? a.x0[(size_t)n * b + jx] : 0.0;
    for (int i = 0; i < N; ++i) {
        xs[j] = xh;
        dxs[j] = a.has_policy ? xh - (inx ? xg[(size_t)n * i + jx] : 0.0) : 0.0;
        wave_sync();
        if (inu) {                                               // controls (forward_pass.jl:17-24)
            double v = ug[(size_t)m * i + ju];
            if (a.has_policy) {
                v += kg[(size_t)m * i + ju] * alpha;             // unew .+= k*α
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += Kg[nm * i + ju + (size_t)m * l] * dxs[l];
                v += s;                                          // unew .+= K*dx
            }
            if (a.has_lims) v = clampd(v, lo, hi);
            if (v != v) v = 0.0;                                 // u[isnan.(u)] .= 0 inside f
            us[ju] = v;
            uo[(size_t)m * i + ju] = v;
        }
        if (inx) xo[(size_t)n * i + jx] = xh;
        wave_sync();
        if (i < N - 1) {                                         // x+ = A x + B u (src/demo_linear.jl:42-46)
            const double *Ai = Ab + (a.dyn_tv ? nn * i : 0), *Bi = Bb + (a.dyn_tv ? nm * i : 0);
            double s0 = 0.0, s1 = 0.0, t = 0.0;
            for (int l = 0; l < n; l += 2) {
                s0 += Ai[jx + (size_t)n * l] * xs[l];
                if (l + 1 < n) s1 += Ai[jx + (size_t)n * (l + 1)] * xs[l + 1];
            }
            for (int q = 0; q < m; ++q) t += Bi[jx + (size_t)n * q] * us[q];
            xh = inx ? (s0 + s1) + t : 0.0;
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

}   // namespace

// returns 1 when not applicable (caller falls back), 0 launched, <0 error
int ddp_launch_forward_big(ddp_handle h, const ddp_problem *p, const double *K, const double *k, const double *x0,
                           const double *u, const double *x, const double *alpha, int nalpha, const double *lims,
                           const int32_t *active, double *xnew, double *unew, double *cnew, double *csum)
{
    if (p->kind != DDP_PROBLEM_LQ || p->n > 64 || p->m > DDP_MAX_M) return 1;
    FBArgs a;
    a.n = p->n; a.m = p->m; a.N = p->N; a.B = p->B; a.nalpha = nalpha;
    a.dyn_tv = p->dyn_tv; a.dyn_batched = p->dyn_batched; a.has_policy = K != nullptr; a.has_lims = lims != nullptr;
    a.A = p->A; a.Bm = p->Bm; a.Q = p->Q; a.R = p->R; a.K = K; a.k = k; a.x0 = x0; a.u = u; a.x = x; a.lims = lims;
    a.active = active;
    for (int i = 0; i < 16; ++i) a.alpha[i] = i < nalpha ? alpha[i] : 0.0;
    a.xnew = xnew; a.unew = unew; a.cnew = cnew; a.csum = csum;
    const dim3 grid((unsigned)((long)p->B * nalpha)), block(DDP_WAVE);
    hipLaunchKernelGGL(forward_big_kernel, grid, block, 0, h->stream, a);
    const size_t shmem = ((size_t)p->n * p->n + (size_t)p->m * p->m) * sizeof(double);
    hipLaunchKernelGGL(cost_rt_kernel, grid, block, shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}
