// back_pass_mf2.hip — launcher of the large-state matrix-core backward pass, 32 < n <= 64, m <= 8 at run time (kernel: back_pass_mf2_kernel.h)
#include "back_pass_mf2_kernel.h"

// returns 1 if this shape is not handled here, 0 launched, <0 error
int ddp_launch_back_pass_mf2(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge, bool defer_64x8_lims)
{
    if (d->n <= 32 || d->n > mf2::NX || d->m < 1 || d->m > mf2::MX) return 1;
    const int nt = (d->n + 15) / 16;                 // 3 or 4 tiles of 16 states
    // the exact (64, 8) shape with a time-varying cost (SURVEY 8d's C4 layout) is 3-4 % faster on the round-5 kernel (8.42 vs 8.66-8.78 ms
    // at N = 256, B = 1 024: this kernel fetches its cost tiles with run-time strides at the top of every step)
    if (defer_64x8_lims && d->n == 64 && d->m == 8 && d->cost_tv && !d->has_lims) return 2;
    BPM2Args a;
    a.n = d->n; a.m = d->m; a.N = d->N; a.B = d->B;
    a.fx_tv = d->fx_tv; a.fx_batched = d->fx_batched; a.cost_tv = d->cost_tv; a.cost_batched = d->cost_batched;
    a.regType = d->regType; a.has_lims = d->has_lims;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.lims = lims;
    a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    a.sink = (double *)h->sink;
    DDP_CHECK(a.sink, "back_pass: the handle has no sink buffer");
    if (d->has_lims) {
        // lims[1,1] > lims[1,2] means "no limits" upstream (backward_pass.jl:31: the Cholesky branch, not a box-QP with infinite bounds,
        // whose projected-Newton iterations and extra exits would differ by rounding).  Two doubles come down once per call — a pass of
        // this shape takes milliseconds.
        DDP_CHECK(lims && u && h->h_pinned, "back_pass: has_lims needs lims and u");
        double *lh = (double *)h->h_pinned;
        DDP_HIP(hipMemcpyAsync(lh, lims, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        DDP_HIP(hipMemcpyAsync(lh + 1, lims + d->m, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        DDP_HIP(hipStreamSynchronize(h->stream));
        if (!(lh[0] > lh[1])) return (defer_64x8_lims && d->n == 64 && d->m == 8) ? 2 : ddp_bpm2_launch_lims(h, a, nt);
        a.has_lims = 0; a.lims = nullptr;
    }
    return nt == 4 ? mf2::launch<4, false>(h, a) : mf2::launch<3, false>(h, a);
}
