// Cost of ONE 8x8 box-QP solve on one wavefront: the lane-parallel form (csrc/boxqp_rows.h) against the every-lane-repeats-it
// form (csrc/boxqp_dev.h), in isolation (no other waves, everything in cache) — the reference point for the QP phase of
// back_pass_mfma_kernel<LIMS>.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../differentialdynamicprogramming.jl_amd/csrc boxqp_rows_bench.hip -o boxqp_rows_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "boxqp_rows.h"
void ddp_set_error(const char *, ...) {}

constexpr int M = 8;
__global__ __launch_bounds__(64) void bench_rows(const double *H, const double *g, const double *lo, const double *up, int reps, double *out, int *iters_out)
{
    const int lane = threadIdx.x, i = lane & 15;
    const bool in = i < M;
    const int qi = in ? i : 0;
    const QPOptsDev o = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};
    double acc = 0.0; int it = 0;
    for (int r = 0; r < reps; ++r) {
        bqr::Rows<M> q;
        for (int j = 0; j < M; ++j) { q.Hrow[j] = in ? H[qi + M * j] : 0.0; q.Hcol[j] = in ? H[j + M * qi] : 0.0; }
        double x; unsigned cl; int iters;
        const int res = bqr::boxqp_rows<M>(q, in ? g[qi] + 1e-9 * r : 0.0, in ? lo[qi] : 0.0, in ? up[qi] : 0.0, 0.0, o, i, x, cl, iters);
        acc += x + res; it += iters;
    }
    out[lane] = acc; if (lane == 0) *iters_out = it;
}
__global__ __launch_bounds__(64) void bench_serial(const double *H, const double *g, const double *lo, const double *up, int reps, double *out, int *iters_out)
{
    const QPOptsDev o = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};
    double acc = 0.0; int it = 0;
    for (int r = 0; r < reps; ++r) {
        double Hh[M * M], gg[M], l[M], u[M], x0[M], x[M], R[M * M], ri[M];
        for (int e = 0; e < M * M; ++e) Hh[e] = H[e];
        for (int j = 0; j < M; ++j) { gg[j] = g[j] + 1e-9 * r; l[j] = lo[j]; u[j] = up[j]; x0[j] = 0.0; }
        unsigned cl; int iters;
        const int res = boxqp_dev_ri<M>(M, Hh, gg, l, u, x0, o, x, R, ri, cl, iters);
        acc += x[threadIdx.x & 7] + res; it += iters;
    }
    out[threadIdx.x] = acc; if (threadIdx.x == 0) *iters_out = it;
}
int main()
{
    double hH[64], hg[8], hl[8], hu[8];
    double a[64];
    for (int e = 0; e < 64; ++e) a[e] = sin(1.0 + 0.7 * e);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { double s = (i == j) ? 0.5 : 0.0; for (int k = 0; k < 8; ++k) s += a[i + 8 * k] * a[j + 8 * k] / 8; hH[i + 8 * j] = s; }
    for (int i = 0; i < 8; ++i) { hg[i] = cos(2.0 + i); hl[i] = -0.3; hu[i] = 0.2 + 0.05 * i; }
    double *dH, *dg, *dl, *du, *dout; int *dit;
    (void)hipMalloc(&dH, sizeof hH); (void)hipMalloc(&dg, 64); (void)hipMalloc(&dl, 64); (void)hipMalloc(&du, 64); (void)hipMalloc(&dout, 512); (void)hipMalloc(&dit, 4);
    (void)hipMemcpy(dH, hH, sizeof hH, hipMemcpyHostToDevice); (void)hipMemcpy(dg, hg, 64, hipMemcpyHostToDevice);
    (void)hipMemcpy(dl, hl, 64, hipMemcpyHostToDevice); (void)hipMemcpy(du, hu, 64, hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int which = 0; which < 2; ++which) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int w = 0; w < 2; ++w) {
            (void)hipEventRecord(e0, 0);
            if (which == 0) bench_rows<<<1, 64>>>(dH, dg, dl, du, reps, dout, dit); else bench_serial<<<1, 64>>>(dH, dg, dl, du, reps, dout, dit);
            (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); int it; (void)hipMemcpy(&it, dit, 4, hipMemcpyDeviceToHost);
        double ho[64]; (void)hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
        printf("%s: %.2f us per QP (%.2f `iter` per QP), checksum %.12g\n", which == 0 ? "boxqp_rows (lane per coordinate)" : "boxqp_dev_ri (every lane repeats)", 1e3 * ms / reps, (double)it / reps, ho[0]);
    }
    return 0;
}
