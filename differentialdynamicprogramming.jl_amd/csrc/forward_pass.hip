// forward_pass.hip — batched closed-loop rollout / line-search candidates for gfx950.
//
// Replaces  forward_pass(traj_new,x0,u,x,α,f,costfun,lims,diff)  (src/forward_pass.jl:9-33) for the
// registered problem families (the Julia closures f / costfun cannot run on the device):
//   LQ        f: src/demo_linear.jl:42-46,  costfun: src/demo_linear.jl:49
//   pendcart  f: src/system_pendcart.jl:83-89, costfun: src/system_pendcart.jl:97-106
// All nalpha step sizes of the serial backtracking search (src/iLQG.jl:267-281) are rolled out
// concurrently; the caller picks the FIRST α (list order) that passes the acceptance test, which is
// what the serial loop returns.
//
// Mapping: a group of G lanes (G = 4/16/32/64 >= n) owns one (trajectory, α) rollout, 64/G rollouts
// per wavefront.  Lane j of a group holds x̂_j; per time step the state is exchanged through LDS once
// (one wave-level hand-off), every lane forms the m controls redundantly (so no second exchange is
// needed), then lane j forms row j of the dynamics.  The step-(i+1) operands K,k,x,u are fetched while
// step i computes.  The time loop is a strict dependency chain: throughput comes from the
// (trajectory, α) batch, not from the horizon.
#include <stdlib.h>
#include <vector>
#include "ddp_internal.h"

namespace {

struct FPArgs {
    int n, m, N, B, nalpha;
    int dyn_tv, dyn_batched, has_policy, has_lims;
    unsigned wrap;                                  // ddp_problem::diff_wrap
    const double *A, *Bm, *Q, *R, *K, *k, *x0, *u, *x, *lims;
    const int32_t *active;
    double alpha[16];
    double g, l, h, d, goal[4];
    double *xnew, *unew, *cnew, *csum;
};

// diff_fun for an angle coordinate (ddp_amd.h, ddp_problem::diff_wrap): d - 2π·rint(d / 2π) with 2π in two parts, so that the result is
// within an ulp of rem2pi(d, RoundNearest) for every difference a rollout can produce
__device__ __forceinline__ double wrap_pi(double d)
{
    const double q = rint(d * 0x1.45f306dc9c883p-3);                       // 1 / 2π
    return fma(-q, 0x1.1a62633145c07p-52, fma(-q, 0x1.921fb54442d18p+2, d));   // 2π = 0x1.921fb54442d18p+2 + 0x1.1a62633145c07p-52
}

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x > hi ? hi : (x < lo ? lo : x); }

template <int G>
__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

template <int KIND, int NS, int MS, int G>
__global__ __launch_bounds__(DDP_WAVE) void forward_pass_kernel(FPArgs a)
{
    constexpr int NMAX = NS ? NS : G;
    constexpr int MM = MS ? MS : DDP_MAX_M;
    constexpr bool PF = NS != 0;          // register prefetch of K_i, x_i only for compiled sizes
    constexpr int KP = PF ? MM * NMAX : 1, XP = PF ? NMAX : 1;
    constexpr int GPW = DDP_WAVE / G;                         // rollouts per wavefront
    const int n = NS ? NS : a.n, m = MS ? MS : a.m, N = a.N, B = a.B;
    const int lane = threadIdx.x, grp = lane / G, jl = lane % G;
    const long total = (long)B * a.nalpha;
    long rho = (long)blockIdx.x * GPW + grp;
    const bool valid = rho < total;
    if (!valid) rho = total - 1;                              // keep the wave converged; stores are masked
    const int b = (int)(rho % B), ai = (int)(rho / B);
    const bool act = valid && !(a.active && a.active[b] == 0);
    const double alpha = a.alpha[ai];
    const int CL = (KIND == DDP_PROBLEM_PENDCART) ? N + 1 : N;

    __shared__ double xs[GPW][G];

    const size_t nn = (size_t)n * n, nm = (size_t)n * m;
    const double *ug = a.u + (size_t)m * N * b;
    const double *xg = a.has_policy ? a.x + (size_t)n * N * b : nullptr;
    const double *Kg = a.has_policy ? a.K + nm * N * b : nullptr;
    const double *kg = a.has_policy ? a.k + (size_t)m * N * b : nullptr;
    double *xo = a.xnew + (size_t)n * N * ((size_t)b + (size_t)B * ai);
    double *uo = a.unew + (size_t)m * N * ((size_t)b + (size_t)B * ai);
    double *co = a.cnew + (size_t)CL * ((size_t)b + (size_t)B * ai);

    // loop-invariant rows (LTI): row jl of A, B, Q
    double Arow[NMAX], Brow[MM], Qrow[NMAX], Rm[MM * MM], lo[MM], hi[MM];
    const double *Ab = nullptr, *Bb = nullptr;
    if (KIND == DDP_PROBLEM_LQ) {
        Ab = a.A + (a.dyn_batched ? nn * (a.dyn_tv ? N : 1) * b : 0);
        Bb = a.Bm + (a.dyn_batched ? nm * (a.dyn_tv ? N : 1) * b : 0);
    }
#pragma unroll
    for (int l = 0; l < NMAX; ++l) {
        const bool in = (jl < n && l < n);
        Arow[l] = (KIND == DDP_PROBLEM_LQ && !a.dyn_tv && in) ? Ab[jl + n * l] : 0.0;
        Qrow[l] = in ? a.Q[jl + n * l] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < MM; ++q) {
        Brow[q] = (KIND == DDP_PROBLEM_LQ && !a.dyn_tv && jl < n && q < m) ? Bb[jl + n * q] : 0.0;
        lo[q] = (a.has_lims && q < m) ? a.lims[q] : 0.0;
        hi[q] = (a.has_lims && q < m) ? a.lims[q + m] : 0.0;
#pragma unroll
        for (int q2 = 0; q2 < MM; ++q2) Rm[q + MM * q2] = (q < m && q2 < m) ? a.R[q + m * q2] : 0.0;
    }

    double xh = (jl < n) ? a.x0[(size_t)n * b + jl] : 0.0;
    double csum = 0.0;

    // operands of step i, fetched one step ahead
    double uc[MM], kc[MM], Kc[KP], xc[XP];
    auto fetch = [&](int i) {
#pragma unroll
        for (int q = 0; q < MM; ++q) {
            uc[q] = (q < m) ? ug[(size_t)m * i + q] : 0.0;
            kc[q] = (a.has_policy && q < m) ? kg[(size_t)m * i + q] : 0.0;
        }
        if (PF && a.has_policy) {
#pragma unroll
            for (int l = 0; l < NMAX; ++l) {
                xc[l] = (l < n) ? xg[(size_t)n * i + l] : 0.0;
#pragma unroll
                for (int q = 0; q < MM; ++q) Kc[q + MM * l] = (q < m && l < n) ? Kg[nm * i + q + m * l] : 0.0;
            }
        }
        if (KIND == DDP_PROBLEM_LQ && a.dyn_tv) {
#pragma unroll
            for (int l = 0; l < NMAX; ++l) Arow[l] = (jl < n && l < n) ? Ab[nn * i + jl + n * l] : 0.0;
#pragma unroll
            for (int q = 0; q < MM; ++q) Brow[q] = (jl < n && q < m) ? Bb[nm * i + jl + n * q] : 0.0;
        }
    };

    double un[MM], kn[MM], Kn[KP], xn_[XP], An[NMAX], Bn[MM];
    fetch(0);
    for (int i = 0; i < N; ++i) {
        // stash step i's operands, start fetching step i+1 (lands during the dependent chain below)
#pragma unroll
        for (int q = 0; q < MM; ++q) { un[q] = uc[q]; kn[q] = kc[q]; Bn[q] = Brow[q]; }
#pragma unroll
        for (int l = 0; l < NMAX; ++l) {
            An[l] = Arow[l];
            if (PF) {
                xn_[l] = xc[l];
#pragma unroll
                for (int q = 0; q < MM; ++q) Kn[q + MM * l] = Kc[q + MM * l];
            }
        }
        if (i + 1 < N) fetch(i + 1);

        xs[grp][jl] = xh;
        wave_sync();
        double xv[NMAX];
#pragma unroll
        for (int l = 0; l < NMAX; ++l) xv[l] = (l < n) ? xs[grp][l] : 0.0;

        // ---- controls (forward_pass.jl:17-24), every lane of the group redundantly
        double uu[MM];
#pragma unroll
        for (int q = 0; q < MM; ++q) {
            double v = un[q];
            if (a.has_policy) {
                v += kn[q] * alpha;                                  // unew .+= k*α
                double s = 0.0;
                if (a.wrap == 0) {                                   // diff_fun = `-` (wave-uniform: the default pays nothing for the hook)
#pragma unroll
                    for (int l = 0; l < NMAX; ++l) {
                        if (PF) s += Kn[q + MM * l] * (xv[l] - xn_[l]);
                        else if (l < n && q < m) s += Kg[nm * i + q + m * l] * (xv[l] - xg[(size_t)n * i + l]);
                    }
                } else {
#pragma unroll
                    for (int l = 0; l < NMAX; ++l) {
                        if (!(PF || (l < n && q < m))) continue;
                        double dxl = PF ? xv[l] - xn_[l] : xv[l] - xg[(size_t)n * i + l];
                        if ((a.wrap >> l) & 1u) dxl = wrap_pi(dxl);
                        s += (PF ? Kn[q + MM * l] : Kg[nm * i + q + m * l]) * dxl;
                    }
                }
                v += s;                                              // unew .+= K*dx
            }
            if (a.has_lims) v = clampd(v, lo[q], hi[q]);
            if (v != v) v = 0.0;                                     // u[isnan.(u)] .= 0 inside f
            uu[q] = (q < m) ? v : 0.0;
        }
        // ---- dynamics row jl and this lane's share of the cost
        double xnext = 0.0, cpart = 0.0;
        if (KIND == DDP_PROBLEM_LQ) {
            double s = 0.0, t = 0.0, qx = 0.0;
#pragma unroll
            for (int l = 0; l < NMAX; ++l) { s += An[l] * xv[l]; qx += Qrow[l] * xv[l]; }
#pragma unroll
            for (int q = 0; q < MM; ++q) t += Bn[q] * uu[q];
            xnext = s + t;                                           // A*x + B*u
            cpart = 0.5 * xh * qx;                                   // .5 x.*(Q*x), row jl
            if (jl == 0) {
                double ru = 0.0;
#pragma unroll
                for (int q = 0; q < MM; ++q) {
                    double r = 0.0;
#pragma unroll
                    for (int q2 = 0; q2 < MM; ++q2) r += Rm[q + MM * q2] * uu[q2];
                    ru += uu[q] * r;
                }
                cpart += 0.5 * ru;
            }
        } else {
            const double gl = a.g / a.l, h = a.h;
            if (jl == 0) xnext = xv[0] + h * xv[1];
            else if (jl == 1) xnext = xv[1] + h * (-gl * sin(xv[0]) + uu[0] / a.l * cos(xv[0]) - a.d * xv[1]);
            else if (jl == 2) xnext = xv[2] + h * xv[3];
            else xnext = xv[3] + h * uu[0];
            double qd = 0.0;
#pragma unroll
            for (int l = 0; l < NMAX; ++l) qd += Qrow[l] * (xv[l] - a.goal[l < 4 ? l : 0]);
            cpart = 0.5 * (xh - a.goal[jl < 4 ? jl : 0]) * qd;
            if (jl == 0) cpart += 0.5 * uu[0] * Rm[0] * uu[0];
        }
        const double ci = group_sum<G>(cpart);
        if (act) {
            if (jl < n) xo[(size_t)n * i + jl] = xh;
            if (jl < m) {
                double v = uu[0];
#pragma unroll
                for (int q = 1; q < MM; ++q) v = (jl == q) ? uu[q] : v;
                uo[(size_t)m * i + jl] = v;
            }
            if (jl == 0) co[i] = ci;
        }
        csum += ci;
        if (KIND == DDP_PROBLEM_PENDCART && i == N - 1) {            // c[end] re-counts x[:,N] with u = 0
            const double cend = group_sum<G>(cpart - ((jl == 0) ? 0.5 * uu[0] * Rm[0] * uu[0] : 0.0));
            if (act && jl == 0) co[N] = cend;
            csum += cend;
        }
        if (i < N - 1) xh = xnext;                                   // f is also called at i == N, result discarded
    }
    if (act && jl == 0) a.csum[(size_t)b + (size_t)B * ai] = csum;
}

template <int KIND, int NS, int MS, int G>
int launch_fp(ddp_handle h, const FPArgs &a)
{
    const long total = (long)a.B * a.nalpha;
    const int gpw = DDP_WAVE / G;
    const dim3 grid((unsigned)((total + gpw - 1) / gpw)), block(DDP_WAVE);
    hipLaunchKernelGGL((forward_pass_kernel<KIND, NS, MS, G>), grid, block, 0, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

int ddp_cost_len(const ddp_problem *p) { return p->kind == DDP_PROBLEM_PENDCART ? p->N + 1 : p->N; }

int ddp_check_cost_diag_host(const ddp_problem *p)
{
    if (!p->cost_diag || !p->Q || !p->R) return 0;
    const int n = p->n, m = p->m;
    int ok = 1;
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) if (i != j && p->Q[i + (size_t)n * j] != 0.0) ok = 0;
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) if (i != j && p->R[i + (size_t)m * j] != 0.0) ok = 0;
    DDP_CHECK(ok, "ddp_problem.cost_diag = 1 but Q or R has a non-zero off-diagonal entry (the fused rollout cost would drop it); set cost_diag = 0");
    return 0;
}

int ddp_check_cost_diag(ddp_handle h, const ddp_problem *p)
{
    if (h->diag_skip > 0) return 0;
    for (const auto &e : h->diag_cache)
        if (e.Q == p->Q && e.R == p->R && e.n == p->n && e.m == p->m && e.Q) {
            DDP_CHECK(e.ok, "forward_pass: ddp_problem.cost_diag = 1 but Q or R has a non-zero off-diagonal entry (the fused rollout cost would drop it); set cost_diag = 0");
            return 0;
        }
    const int n = p->n, m = p->m;
    std::vector<double> q((size_t)n * n), r((size_t)m * m);
    DDP_HIP(hipMemcpyAsync(q.data(), p->Q, q.size() * 8, hipMemcpyDeviceToHost, h->stream));
    DDP_HIP(hipMemcpyAsync(r.data(), p->R, r.size() * 8, hipMemcpyDeviceToHost, h->stream));
    DDP_HIP(hipStreamSynchronize(h->stream));
    int ok = 1;
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) if (i != j && q[i + (size_t)n * j] != 0.0) ok = 0;
    for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) if (i != j && r[i + (size_t)m * j] != 0.0) ok = 0;
    // only a PASS is remembered (a refused pair is looked at again on the next call: the caller may have repaired Q in place); the
    // entry dies with ddp_free of the allocation it points into and with ddp_reload_env
    if (ok) { auto &e = h->diag_cache[h->diag_next++ % 8]; e.Q = p->Q; e.R = p->R; e.n = n; e.m = m; e.ok = 1; }
    DDP_CHECK(ok, "forward_pass: ddp_problem.cost_diag = 1 but Q or R has a non-zero off-diagonal entry (the fused rollout cost would drop it); set cost_diag = 0");
    return 0;
}

int ddp_forward_pass_f64_dev(ddp_handle h, const ddp_problem *p, const double *K, const double *k,
                             const double *x0, const double *u, const double *x, const double *alpha,
                             int nalpha, const double *lims, const int32_t *active, double *xnew,
                             double *unew, double *cnew, double *csum)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && p, "forward_pass: null handle/problem");
    DDP_CHECK(nalpha >= 1 && nalpha <= 16, "forward_pass: nalpha=%d out of [1,16]", nalpha);
    DDP_CHECK((K == nullptr) == (k == nullptr), "forward_pass: K and k must both be given or both NULL");
    DDP_CHECK(!K || x, "forward_pass: a non-empty policy needs the nominal trajectory x");
    DDP_CHECK(p->m <= DDP_MAX_M && p->n <= 64, "forward_pass: n=%d m=%d unsupported (n<=64, m<=%d)", p->n, p->m, DDP_MAX_M);
    // the trailing field of ddp_problem (library 0.2.0): a caller built against the older layout, or one that does not zero the struct,
    // hands over garbage here — anything but 0 / 1 is refused instead of silently selecting the diagonal-cost rollout
    DDP_CHECK(p->cost_diag == 0 || p->cost_diag == 1, "forward_pass: ddp_problem.cost_diag = %d (0 or 1; zero-initialise the struct)", p->cost_diag);
    if (p->cost_diag) { const int rd = ddp_check_cost_diag(h, p); if (rd) return rd; }
    // diff_fun with wrapped coordinates: the pendulum's row / lane kernels and the run-time-sized kernel below implement it
    DDP_CHECK(p->diff_wrap == 0 || (p->n <= DDP_MAX_N_GENERIC && (p->n >= 32 || (p->diff_wrap >> p->n) == 0)),
              "forward_pass: ddp_problem.diff_wrap = 0x%x needs n <= %d and no bits at or above n = %d (zero-initialise the struct)", p->diff_wrap, DDP_MAX_N_GENERIC, p->n);
    if (p->n > DDP_MAX_N_GENERIC || (p->diff_wrap == 0 && ddp_env(h, ENV_FORWARD) && ddp_env(h, ENV_FORWARD)[0] == 'b')) {   // large states
        const int rc = ddp_launch_forward_big(h, p, K, k, x0, u, x, alpha, nalpha, lims, active, xnew, unew, cnew, csum);
        if (rc <= 0) return rc;                             // (the launcher names the kernel)
        DDP_CHECK(p->n <= DDP_MAX_N_GENERIC, "forward_pass: n=%d m=%d has no kernel", p->n, p->m);
    }
    // DDP_FORWARD=group forces the group-of-lanes kernel (A/B timing, tests of both code paths)
    const char *fwd_env = ddp_env(h, ENV_FORWARD);               // read per call so tests can switch paths
    const bool force_group = (fwd_env && fwd_env[0] == 103);
    if (!force_group && p->diff_wrap != 0) {                     // diff_fun with wrapped coordinates: the pendulum's row / lane kernels have it
        const int rc = ddp_launch_forward_dpp(h, p, K, k, x0, u, x, alpha, nalpha, lims, active, xnew, unew, cnew, csum);
        if (rc <= 0) { h->last_kernel[1] = "forward_dpp_kernel"; return rc; }
    }
    if (!force_group && p->diff_wrap == 0) {
        const int rp = ddp_launch_forward_pipe(h, p, K, k, x0, u, x, alpha, nalpha, lims, active, xnew, unew, cnew, csum);
        if (rp <= 0) return rp;                             // (the launcher names the kernel)
        const int rc = ddp_launch_forward_dpp(h, p, K, k, x0, u, x, alpha, nalpha, lims, active, xnew, unew, cnew, csum);
        if (rc <= 0) { h->last_kernel[1] = "forward_dpp_kernel"; return rc; }
        // every other LQ shape a 16-lane row holds: the row kernel compiled for padded sizes (DDP_FORWARD=group keeps the old path)
        const int rr = ddp_launch_forward_row(h, p, K, k, x0, u, x, alpha, nalpha, lims, active, xnew, unew, cnew, csum);
        if (rr <= 0) { h->last_kernel[1] = "forward_row_kernel"; return rr; }
        // what no row holds (n > 14 or m > 4): one wave per rollout with the operands of a step requested a step ahead
        // (forward_mid_kernel, n <= 32; n = 24, m = 4, N = 300, B = 1 024 LTV: 0.56 ms against 3.1 on forward_big_kernel and 6.0 on the
        // group-of-lanes kernel below), forward_big_kernel above n = 32
        // (the row launcher also declines n > 12 with m > 2 — (13,3) (13,4) (14,3) (14,4): they used to drop to the group-of-lanes kernel — ADVICE r5)
        if (p->kind == DDP_PROBLEM_LQ && (p->n > 14 || p->m > 4 || (p->n > 12 && p->m > 2))) {
            const int rb = ddp_launch_forward_big(h, p, K, k, x0, u, x, alpha, nalpha, lims, active, xnew, unew, cnew, csum);
            if (rb <= 0) return rb;
        }
    }
    h->last_kernel[1] = "forward_pass_kernel";
    FPArgs a;
    a.n = p->n; a.m = p->m; a.N = p->N; a.B = p->B; a.nalpha = nalpha;
    a.dyn_tv = p->dyn_tv; a.dyn_batched = p->dyn_batched; a.has_policy = K != nullptr; a.has_lims = lims != nullptr;
    a.wrap = p->diff_wrap;
    a.A = p->A; a.Bm = p->Bm; a.Q = p->Q; a.R = p->R; a.K = K; a.k = k; a.x0 = x0; a.u = u; a.x = x; a.lims = lims;
    a.active = active;
    for (int i = 0; i < 16; ++i) a.alpha[i] = i < nalpha ? alpha[i] : 0.0;
    a.g = p->g; a.l = p->l; a.h = p->h; a.d = p->d;
    for (int i = 0; i < 4; ++i) a.goal[i] = p->goal[i];
    a.xnew = xnew; a.unew = unew; a.cnew = cnew; a.csum = csum;
    if (p->kind == DDP_PROBLEM_PENDCART) {
        DDP_CHECK(p->n == 4 && p->m == 1, "forward_pass: pendcart needs n=4, m=1");
        return launch_fp<DDP_PROBLEM_PENDCART, 4, 1, 4>(h, a);
    }
    DDP_CHECK(p->kind == DDP_PROBLEM_LQ, "forward_pass: unknown problem kind %d", p->kind);
    if (p->n == 10 && p->m == 2) return launch_fp<DDP_PROBLEM_LQ, 10, 2, 16>(h, a);
    if (p->n == 4 && p->m == 1) return launch_fp<DDP_PROBLEM_LQ, 4, 1, 4>(h, a);
    if (p->n == 6 && p->m == 3) return launch_fp<DDP_PROBLEM_LQ, 6, 3, 8>(h, a);
    if (p->n <= 16) return launch_fp<DDP_PROBLEM_LQ, 0, 0, 16>(h, a);
    return launch_fp<DDP_PROBLEM_LQ, 0, 0, 32>(h, a);
}
