// back_pass_gps_lane.hip — back_pass_gps (src/backward_pass.jl:259-350) with ONE LANE per trajectory for n = 4, m = 1 | 2
// (the pendcart of BASELINE config 5).  Every matrix of a step lives in the registers of one lane — plain FMAs, no cross-lane
// traffic, 64 trajectories per wave.  The run-time-sized kernel this replaces for the shape (one wave per trajectory, operands
// in LDS, 550 spilled VGPRs) took 117 ms for 4 096 trajectories of 600 steps.
// Arithmetic and failure semantics as back_pass_kernel<…, GPS> in back_pass.hip, summed in the index order of the
// reference's matrix products; Quui = inv(Quu) in closed form (m <= 2).
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

struct GLArgs {
    int N, B, fx_batched, cost_batched, eta_tv;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lims, *u;
    const double *cxkl, *cukl, *cxxkl, *cxukl, *cuukl, *eta;
    const int32_t *active;
    double *K, *k, *Quu, *Quui, *Vx, *Vxx, *dV;
    int32_t *diverge;
};

template <int MS>
__device__ __forceinline__ void inv_sym(const double (&Q)[MS * MS], double (&Qi)[MS * MS])
{
    static_assert(MS == 1 || MS == 2, "closed-form inverse");
    if constexpr (MS == 1) {
        Qi[0] = 1.0 / Q[0];
    } else {
        const double det = Q[0] * Q[3] - Q[1] * Q[2], r = 1.0 / det;
        Qi[0] = Q[3] * r; Qi[1] = -Q[1] * r; Qi[2] = -Q[2] * r; Qi[3] = Q[0] * r;
    }
}

template <int NS, int MS, bool LIMS>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_gps_lane_kernel(GLArgs a)
{
    constexpr int n = NS, m = MS, DR = 2;                      // DR: steps of operands in flight per lane
    constexpr size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const int N = a.N;
    long tb = (long)blockIdx.x * DDP_WAVE + threadIdx.x;
    const bool valid = tb < a.B;
    if (!valid) tb = a.B - 1;
    const int b = (int)tb;
    const bool act = valid && !(a.active && a.active[b] == 0);
    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    const double *ug = LIMS ? a.u + (size_t)m * N * b : nullptr;
    const double *fx = a.fx + (a.fx_batched ? nn * N * b : 0), *fu = a.fu + (a.fx_batched ? nm * N * b : 0);
    const double *cxx = a.cxx + (a.cost_batched ? nn * N * b : 0), *cxu = a.cxu + (a.cost_batched ? nm * N * b : 0),
                 *cuu = a.cuu + (a.cost_batched ? mm * N * b : 0);
    const double *cxkl = a.cxkl + (size_t)n * N * b, *cukl = a.cukl + (size_t)m * N * b, *cxxkl = a.cxxkl + nn * N * b,
                 *cxukl = a.cxukl + nm * N * b, *cuukl = a.cuukl + mm * N * b;
    const double *etag = a.eta + (a.eta_tv ? (size_t)N * b : b);
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + mm * N * b, *Quuig = a.Quui + mm * N * b,
           *Vxg = a.Vx + (size_t)n * N * b, *Vxxg = a.Vxx + nn * N * b;
    bool nolims = true;
    double limlo[m], limhi[m];
    if (LIMS) {
        nolims = a.lims[0] > a.lims[m];                             // :303
#pragma unroll
        for (int q = 0; q < m; ++q) { limlo[q] = a.lims[q]; limhi[q] = a.lims[q + m]; }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35

    struct Ops { double F[n * n], Fu[n * m], c[n], cu_[m], u_[LIMS ? m : 1], Cxx[n * n], Cxu[n * m], Cuu[m * m],
                 kx[n], ku[m], kxx[n * n], kxu[n * m], kuu[m * m], eta; };
    auto fetch = [&](int i, Ops &o) {
#pragma unroll
        for (int e = 0; e < n * n; ++e) { o.F[e] = fx[nn * i + e]; o.Cxx[e] = cxx[nn * i + e]; o.kxx[e] = cxxkl[nn * i + e]; }
#pragma unroll
        for (int e = 0; e < n * m; ++e) { o.Fu[e] = fu[nm * i + e]; o.Cxu[e] = cxu[nm * i + e]; o.kxu[e] = cxukl[nm * i + e]; }
#pragma unroll
        for (int e = 0; e < n; ++e) { o.c[e] = cx[(size_t)n * i + e]; o.kx[e] = cxkl[(size_t)n * i + e]; }
#pragma unroll
        for (int e = 0; e < m; ++e) {
            o.cu_[e] = cu[(size_t)m * i + e]; o.ku[e] = cukl[(size_t)m * i + e];
            if (LIMS) o.u_[e] = ug[(size_t)m * i + e];
        }
#pragma unroll
        for (int e = 0; e < m * m; ++e) { o.Cuu[e] = cuu[mm * i + e]; o.kuu[e] = cuukl[mm * i + e]; }
        o.eta = etag[a.eta_tv ? i : 0];
    };
    // terminal step (:280-283)
    double V[n * n], v[n], kprev[m];
    {
        const size_t tl = (size_t)(N - 1);
        const double etaN = etag[a.eta_tv ? N - 1 : 0];
        double Qn[m * m], Qni[m * m];
#pragma unroll
        for (int e = 0; e < n * n; ++e) V[e] = cxx[nn * tl + e];
#pragma unroll
        for (int e = 0; e < n; ++e) v[e] = cx[(size_t)n * tl + e];
#pragma unroll
        for (int e = 0; e < m * m; ++e) Qn[e] = cuu[mm * tl + e] / etaN + cuukl[mm * tl + e];
        inv_sym<m>(Qn, Qni);
#pragma unroll
        for (int q = 0; q < m; ++q) kprev[q] = 0.0;
        if (act) {
#pragma unroll
            for (int e = 0; e < n * n; ++e) Vxxg[nn * tl + e] = V[e];
#pragma unroll
            for (int e = 0; e < n; ++e) Vxg[(size_t)n * tl + e] = v[e];
#pragma unroll
            for (int e = 0; e < n * m; ++e) Kg[nm * tl + e] = 0.0;
#pragma unroll
            for (int e = 0; e < m; ++e) kg[(size_t)m * tl + e] = 0.0;
#pragma unroll
            for (int e = 0; e < m * m; ++e) { Quug[mm * tl + e] = Qn[e]; Quuig[mm * tl + e] = Qni[e]; }
        }
    }
    double dV0 = 0.0, dV1 = 0.0;
    int diverge = 0;

    auto step = [&](int i, const Ops &o) {
        const double *Fx = o.F, *Fu = o.Fu;
        const double et = o.eta;
        double Qu[m], Qx[n], fuV[m * n], fxV[n * n], Qux[m * n], Quu[m * m], Qxx[n * n];
#pragma unroll
        for (int q = 0; q < m; ++q) {                                   // Qu (:286)
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < n; ++l) s += Fu[l + n * q] * v[l];
            Qu[q] = o.cu_[q] + s;
        }
#pragma unroll
        for (int j = 0; j < n; ++j) {                                   // Qx (:287)
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < n; ++l) s += Fx[l + n * j] * v[l];
            Qx[j] = o.c[j] + s;
        }
#pragma unroll
        for (int c = 0; c < n; ++c) {                                   // fu'Vxx, fx'Vxx
#pragma unroll
            for (int q = 0; q < m; ++q) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < n; ++l) s += Fu[l + n * q] * V[l + n * c];
                fuV[q + m * c] = s;
            }
#pragma unroll
            for (int r = 0; r < n; ++r) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < n; ++l) s += Fx[l + n * r] * V[l + n * c];
                fxV[r + n * c] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < n; ++j)
#pragma unroll
            for (int q = 0; q < m; ++q) {                               // Qux (:288)
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < n; ++l) s += fuV[q + m * l] * Fx[l + n * j];
                Qux[q + m * j] = o.Cxu[j + n * q] + s;
            }
#pragma unroll
        for (int q2 = 0; q2 < m; ++q2)
#pragma unroll
            for (int q = 0; q < m; ++q) {                               // Quu (:289)
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < n; ++l) s += fuV[q + m * l] * Fu[l + n * q2];
                Quu[q + m * q2] = o.Cuu[q + m * q2] + s;
            }
#pragma unroll
        for (int c = 0; c < n; ++c)
#pragma unroll
            for (int r = 0; r < n; ++r) {                               // Qxx (:290)
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < n; ++l) s += fxV[r + n * l] * Fx[l + n * c];
                Qxx[r + n * c] = o.Cxx[r + n * c] + s;
            }
        // Q• <- Q•/η + c•kl  (:294-299), Quu = .5(Quu + Quu')  (:301)
#pragma unroll
        for (int q = 0; q < m; ++q) Qu[q] = Qu[q] / et + o.ku[q];
#pragma unroll
        for (int e = 0; e < n * m; ++e) Qux[e] = Qux[e] / et + o.kxu[e];
#pragma unroll
        for (int e = 0; e < m * m; ++e) Quu[e] = Quu[e] / et + o.kuu[e];
#pragma unroll
        for (int j = 0; j < n; ++j) Qx[j] = Qx[j] / et + o.kx[j];
#pragma unroll
        for (int e = 0; e < n * n; ++e) Qxx[e] = Qxx[e] / et + o.kxx[e];
        if constexpr (m == 2) { const double s = 0.5 * (Quu[1] + Quu[2]); Quu[1] = s; Quu[2] = s; }
        // ---- gains (:303-335), directly on the KL-augmented Quu (no λ)
        double R[m * m], ri[m], ki[m], Ki[m * n];
        unsigned clamped = 0u;
        int fail;
        if (!LIMS || nolims) {
            fail = chol_masked_ri<m>(m, Quu, 0u, R, ri);
#pragma unroll
            for (int q = 0; q < m; ++q) ki[q] = Qu[q];
            chol_solve_ri<m>(m, R, ri, ki);
#pragma unroll
            for (int q = 0; q < m; ++q) ki[q] = -ki[q];
        } else {
            double lo[m], up[m];
#pragma unroll
            for (int q = 0; q < m; ++q) { lo[q] = limlo[q] - o.u_[q]; up[q] = limhi[q] - o.u_[q]; }
            int iters;
            const int result = boxqp_dev_ri<m>(m, Quu, Qu, lo, up, kprev, qpo, ki, R, ri, clamped, iters);
            fail = (result < 1);
        }
        const bool alive = diverge == 0 && !fail;
        const bool failing = diverge == 0 && fail;
        if (failing) diverge = i + 1;
#pragma unroll
        for (int j = 0; j < n; ++j) {
            double col[m];
#pragma unroll
            for (int q = 0; q < m; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : Qux[q + m * j];
            chol_solve_ri<m>(m, R, ri, col);
#pragma unroll
            for (int q = 0; q < m; ++q) Ki[q + m * j] = ((clamped >> q) & 1u) ? 0.0 : -col[q];
        }
        // ---- value update (:338-341) with the KL-augmented Quu, Qux
        double Quuk[m], kQuuk = 0.0, kQu = 0.0;
#pragma unroll
        for (int q = 0; q < m; ++q) {
            double s = 0.0;
#pragma unroll
            for (int q2 = 0; q2 < m; ++q2) s += Quu[q + m * q2] * ki[q2];
            Quuk[q] = s;
        }
#pragma unroll
        for (int q = 0; q < m; ++q) { kQuuk += ki[q] * Quuk[q]; kQu += ki[q] * Qu[q]; }
        if (alive) { dV0 += kQu; dV1 += 0.5 * kQuuk; }
        double vn[n], Vn[n * n];
#pragma unroll
        for (int j = 0; j < n; ++j) {
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int q = 0; q < m; ++q) { s1 += Ki[q + m * j] * Quuk[q]; s2 += Ki[q + m * j] * Qu[q]; s3 += Qux[q + m * j] * ki[q]; }
            vn[j] = ((Qx[j] + s1) + s2) + s3;
        }
#pragma unroll
        for (int c = 0; c < n; ++c)
#pragma unroll
            for (int r = 0; r < n; ++r) {
                double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int q2 = 0; q2 < m; ++q2) {
                    double kq = 0.0;
#pragma unroll
                    for (int q = 0; q < m; ++q) kq += Ki[q + m * r] * Quu[q + m * q2];
                    s1 += kq * Ki[q2 + m * c];
                }
#pragma unroll
                for (int q = 0; q < m; ++q) { s2 += Ki[q + m * r] * Qux[q + m * c]; s3 += Qux[q + m * r] * Ki[q + m * c]; }
                Vn[r + n * c] = ((Qxx[r + n * c] + s1) + s2) + s3;
            }
#pragma unroll
        for (int c = 0; c < n; ++c)
#pragma unroll
            for (int r = 0; r < n; ++r) V[r + n * c] = .5 * (Vn[r + n * c] + Vn[c + n * r]);
#pragma unroll
        for (int j = 0; j < n; ++j) v[j] = vn[j];
#pragma unroll
        for (int q = 0; q < m; ++q) kprev[q] = ki[q];
        double Qi[m * m];
        inv_sym<m>(Quu, Qi);                                            // Quui[:,:,i] = inv(Quu[:,:,i])  (:346)
        // ---- stores (a diverged trajectory keeps writing; its range is zero-filled after the loop, Quu of the failing step stays)
        if (act) {
#pragma unroll
            for (int e = 0; e < n * m; ++e) Kg[nm * i + e] = Ki[e];
#pragma unroll
            for (int e = 0; e < m; ++e) kg[(size_t)m * i + e] = ki[e];
#pragma unroll
            for (int e = 0; e < m * m; ++e) { Quug[mm * i + e] = Quu[e]; Quuig[mm * i + e] = failing ? 0.0 : Qi[e]; }
#pragma unroll
            for (int e = 0; e < n; ++e) Vxg[(size_t)n * i + e] = v[e];
#pragma unroll
            for (int e = 0; e < n * n; ++e) Vxxg[nn * i + e] = V[e];
        }
    };
    Ops ring[DR];
#pragma unroll
    for (int d = 0; d < DR; ++d) { const int i = N - 2 - d; fetch(i >= 0 ? i : 0, ring[d]); }
    int i0 = N - 2;
    for (; i0 - (DR - 1) >= 0; i0 -= DR) {
#pragma unroll
        for (int d = 0; d < DR; ++d) {
            step(i0 - d, ring[d]);
            fetch(i0 - d - DR >= 0 ? i0 - d - DR : 0, ring[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < DR; ++d) {
        if (i0 - d >= 0) step(i0 - d, ring[d]);
    }
    if (diverge && act) {           // outputs earlier in time than the failing step are zero; Quu of the failing step itself stays
        const size_t ie = (size_t)diverge;
        for (size_t e = 0; e < nm * ie; ++e) Kg[e] = 0.0;
        for (size_t e = 0; e < (size_t)m * ie; ++e) kg[e] = 0.0;
        for (size_t e = 0; e < (size_t)n * ie; ++e) Vxg[e] = 0.0;
        for (size_t e = 0; e < nn * ie; ++e) Vxxg[e] = 0.0;
        for (size_t e = 0; e < mm * (ie - 1); ++e) Quug[e] = 0.0;
        for (size_t e = 0; e < mm * ie; ++e) Quuig[e] = 0.0;
    }
    if (act) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; a.diverge[b] = diverge; }
}

}   // namespace

// returns 1 if this shape is not handled here (caller falls back), 0 launched, <0 error
int ddp_launch_back_pass_gps_lane(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                                  const double *cxx, const double *cxu, const double *cuu, const double *fx,
                                  const double *fu, const ddp_kl_cost_terms *kl, const double *lims, const double *u,
                                  const int32_t *active, double *K, double *k, double *Quu, double *Quui, double *Vx,
                                  double *Vxx, double *dV, int32_t *diverge)
{
    if (!(d->n == 4 && (d->m == 1 || d->m == 2)) || d->N < 2 || !d->fx_tv || !d->cost_tv) return 1;
    GLArgs a;
    a.N = d->N; a.B = d->B; a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched; a.eta_tv = kl->eta_tv;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lims = lims; a.u = u;
    a.cxkl = kl->cx; a.cukl = kl->cu; a.cxxkl = kl->cxx; a.cxukl = kl->cxu; a.cuukl = kl->cuu; a.eta = kl->eta;
    a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Quui = Quui; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    const dim3 grid((unsigned)((d->B + DDP_WAVE - 1) / DDP_WAVE)), block(DDP_WAVE);
    if (d->m == 1) {
        if (d->has_lims) hipLaunchKernelGGL((back_pass_gps_lane_kernel<4, 1, true>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((back_pass_gps_lane_kernel<4, 1, false>), grid, block, 0, h->stream, a);
    } else {
        if (d->has_lims) hipLaunchKernelGGL((back_pass_gps_lane_kernel<4, 2, true>), grid, block, 0, h->stream, a);
        else hipLaunchKernelGGL((back_pass_gps_lane_kernel<4, 2, false>), grid, block, 0, h->stream, a);
    }
    DDP_HIP(hipGetLastError());
    return 0;
}
