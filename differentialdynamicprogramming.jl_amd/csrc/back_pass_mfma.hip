// back_pass_mfma.hip — launcher of the n = 64, m = 8 matrix-core backward pass (kernel: back_pass_mfma_kernel.h)
#include "back_pass_mfma_kernel.h"


// returns 1 if this shape is not handled here, 0 launched, <0 error
int ddp_launch_back_pass_mfma(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                              const double *cxx, const double *cxu, const double *cuu, const double *fx,
                              const double *fu, const double *lambda, const double *lims, const double *u,
                              const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                              double *Vxx, double *dV, int32_t *diverge)
{
    if (d->n != n || d->m != m) return 1;
    BPMArgs a;
    a.N = d->N; a.B = d->B;
    a.fx_tv = d->fx_tv; a.fx_batched = d->fx_batched; a.cost_tv = d->cost_tv; a.cost_batched = d->cost_batched;
    a.regType = d->regType; a.has_lims = d->has_lims;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu; a.lambda = lambda; a.lims = lims;
    a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    if (d->has_lims) {
        // lims[1,1] > lims[1,2] means "no limits" upstream (backward_pass.jl:31: the Cholesky branch, not a box-QP with infinite bounds,
        // whose projected-Newton iterations and extra exits would differ by rounding).  Two doubles come down once per call — a pass of
        // this shape takes milliseconds.
        DDP_CHECK(lims && u && h->h_pinned, "back_pass: has_lims needs lims and u");
        double *lh = (double *)h->h_pinned;
        DDP_HIP(hipMemcpyAsync(lh, lims, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        DDP_HIP(hipMemcpyAsync(lh + 1, lims + m, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        DDP_HIP(hipStreamSynchronize(h->stream));
        if (!(lh[0] > lh[1])) return ddp_bpm_launch_lims(h, a);
        a.has_lims = 0; a.lims = nullptr;
    }
    return ddp_bpm_launch<false>(h, a);
}
