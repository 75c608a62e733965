// The floor of ONE step of the pendulum rollout (n = 4, m = 1: csrc/forward_pass_dpp.hip, forward_pend_row_kernel — one 16-lane row per
// rollout; src/forward_pass.jl:17-24, src/system_pendcart.jl:83-89) on one wave of gfx950, with everything that is not on the dependency
// chain x̂_i -> u_i -> x̂_{i+1} taken away.  The pieces (DPP broadcasts, row sums, sin / cos, clamp) are the PRODUCTION ones (this file
// includes the kernel source); the step below restates the kernel's step lambda with its memory traffic as switches.
//   mode 0  dynamics only: θ, θ' broadcast, sin / cos (pend_math.h, 42 instructions), acceleration, Euler step
//   mode 1  + the control law u = ū + α k + K (x̂ - x) (row_shl, product, broadcast multiply-add, 4-term row sum), clamp, NaN test
//   mode 2  + the result store of the step (one 8-byte store per lane into the sink)
//   mode 3  + the fused cost (an LDS tile write per step, the tile summed and flushed every 16 steps)
//   mode 4  + the operand stream (one 8-byte load per lane and step, eight steps ahead)
// Prints ns and shader-clock ticks per step.  Production (C3: N = 600, B = 4 096 rollouts, limits, fused cost; 1 024 waves = one per
// SIMD): forward_pend_row_kernel 0.186 ms / 600 steps = 310 ns per step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../differentialdynamicprogramming.jl_amd/csrc pend_row_chain_floor.hip -o pend_row_chain_floor
#include <cstdarg>
#include "../../differentialdynamicprogramming.jl_amd/csrc/forward_pass_dpp.hip"

void ddp_set_error(const char *, ...) {}

namespace {
template <int MODE>
__global__ __launch_bounds__(64) void chain(int steps, double *out, long long *ticks, const double *stream, double *sink)
{
    constexpr int n = 4, G = 16, TS = 17, D = 8;
    constexpr bool POLICY = MODE >= 1, STORE = MODE >= 2, FUSE = MODE >= 3, FETCH = MODE >= 4;
    __shared__ double ctile[4 * 16 * TS];
    const int lane = threadIdx.x, grp = lane / G, j = lane % G;
    const bool inx = j < n, is1 = j == 1, is3 = j == 3, isu = j == n;
    const double alpha = 0.5, lo = -5.0, hi = 5.0, gl = 9.82 / 0.35, il = 1.0 / 0.35, l = 0.35, h = 0.01, dd = 0.99;
    double one = 1.0;
    asm volatile("" : "+v"(one));
    // lanes 0-3: K_i[j]; 4-7: x_i[j-4]; 8: ū_i; 9: k_i  (the operand stream of the production kernel)
    double ldc = j < 4 ? 0.1 * (j + 1) : (j < 8 ? (j == 4 ? 3.0 : 0.0) : (j == 8 ? 0.2 : (j == 9 ? -0.1 : 0.0)));
    double xh = inx ? (j == 0 ? 2.9 : 0.05 * j) : 0.0;
    double cw = inx ? 0.5 * (j == 0 ? 10.0 : 1.0) : (j == n ? 0.5 : 0.0), cg = (j == 0) ? 3.14159 : 0.0, cacc = 0.0;
    double *ct = &ctile[grp * 16 * TS], *ctw = ct + j;
    PendTrig trig;
    trig.init();
    dpp_fence(xh);
    const double *src = stream + lane;
    double ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = FETCH ? src[64 * d] : ldc;
    auto step = [&](int i, double ld) __attribute__((always_inline)) {
        double uu = row_bcast_all<8>(ld);
        if (POLICY) {
            const double xi = __builtin_amdgcn_update_dpp(0.0, ld, 0x104, 0xf, 0xf, true);
            double dxj = xh - xi;
            double pr = ld * dxj;
            fmac_bc<9>(uu, ld, alpha);
            dpp_fence(pr);
            double s1 = 0.0;
            RowSum<n>::run(uu, s1, pr, one);
            uu += s1;
            const bool nan = uu != uu;
            uu = fmin(fmax(uu, lo), hi);
            uu = nan ? 0.0 : uu;
        }
        const double v = isu ? uu : xh;
        if (STORE) sink[64 * (i & 7) + lane] = v;
        if (FUSE) {
            const double dv = v - cg;
            ctw[(i & 7) * TS] = (cw * dv) * dv;
        }
        const double x0v = row_bcast_all<0>(xh), x1v = row_bcast_all<1>(xh);
        double sn, cs;
        pend_sincos(trig, x0v, sn, cs);
        double ul = uu * il;
        ul = __builtin_fma(__builtin_fma(-ul, l, uu), il, ul);
        double acc = -gl * sn + ul * cs - dd * x1v;
        asm("" : "+v"(acc));
        const double nxt = __builtin_amdgcn_update_dpp(0.0, xh, 0xf9, 0xf, 0xf, true);
        const double inc = is1 ? acc : (is3 ? uu : nxt);
        xh = xh + h * inc;
        dpp_fence(xh);
    };
    const long long t0 = __builtin_readcyclecounter();
    for (int i0 = 0; i0 < steps; i0 += D) {
        if (FUSE) ctw = ct + j + (i0 & 8) * TS;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double ld = ring[d];
            asm volatile("" : "+v"(ld));
            step(i0 + d, ld);
            if (FETCH) ring[d] = src[64 * ((i0 + d + D) & 63)];
        }
        if (FUSE && (i0 & 8)) {
            wave_sync();
            double c = 0.0;
#pragma unroll
            for (int q = 0; q < n + 1; ++q) c += ct[j * TS + q];
            cacc += c;
            if (STORE) sink[512 + lane] = c;
            wave_sync();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[lane] = xh + cacc;
    if (lane == 0) ticks[0] = t1 - t0;
}

// The same step on `gridDim.x` work-groups of WPB waves at production's footprint: the operand stream of a wave is that of its
// trajectory (8 waves = the 8 step sizes share one), the results go to 40-byte rows per rollout and step (lanes 0-4), N = 600.
template <int MODE, int WPB>
__global__ __launch_bounds__(WPB * 64) void chain_mw(int steps, double *out, const double *stream, double *sink)
{
    constexpr int n = 4, G = 16, TS = 17, D = 8;
    constexpr bool POLICY = MODE >= 1, STORE = MODE >= 2, FUSE = MODE >= 3, FETCH = MODE >= 4;
    __shared__ double ctile[WPB * 4 * 16 * TS];
    const int lane = threadIdx.x % 64, wib = threadIdx.x / 64, grp = lane / G, j = lane % G;
    const long wave = (long)blockIdx.x * WPB + wib;
    const bool inx = j < n, is1 = j == 1, is3 = j == 3, isu = j == n;
    const double alpha = 0.5, lo = -5.0, hi = 5.0, gl = 9.82 / 0.35, il = 1.0 / 0.35, l = 0.35, h = 0.01, dd = 0.99;
    double one = 1.0;
    asm volatile("" : "+v"(one));
    double ldc = j < 4 ? 0.1 * (j + 1) : (j < 8 ? (j == 4 ? 3.0 : 0.0) : (j == 8 ? 0.2 : (j == 9 ? -0.1 : 0.0)));
    double xh = inx ? (j == 0 ? 2.9 : 0.05 * j) : 0.0;
    double cw = inx ? 0.5 * (j == 0 ? 10.0 : 1.0) : (j == n ? 0.5 : 0.0), cg = (j == 0) ? 3.14159 : 0.0, cacc = 0.0;
    double *ct = &ctile[(wib * 4 + grp) * 16 * TS], *ctw = ct + j;
    PendTrig trig;
    trig.init();
    dpp_fence(xh);
    const double *src = stream + (wave / 2) * (long)steps * 16 * 4 + lane;          // 4 trajectories per wave, 2 waves per 8 step sizes
    double *dst = sink + ((wave * 4 + grp) * (long)steps) * 5 + j;
    const bool st = j < 5;
    double ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = FETCH ? src[64 * d] : ldc;
    auto step = [&](int i, double ld) __attribute__((always_inline)) {
        double uu = row_bcast_all<8>(ld);
        if (POLICY) {
            const double xi = __builtin_amdgcn_update_dpp(0.0, ld, 0x104, 0xf, 0xf, true);
            double dxj = xh - xi;
            double pr = ld * dxj;
            fmac_bc<9>(uu, ld, alpha);
            dpp_fence(pr);
            double s1 = 0.0;
            RowSum<n>::run(uu, s1, pr, one);
            uu += s1;
            const bool nan = uu != uu;
            uu = fmin(fmax(uu, lo), hi);
            uu = nan ? 0.0 : uu;
        }
        const double v = isu ? uu : xh;
        if (STORE && st) dst[5 * i] = v;
        if (FUSE) {
            const double dv = v - cg;
            ctw[(i & 7) * TS] = (cw * dv) * dv;
        }
        const double x0v = row_bcast_all<0>(xh), x1v = row_bcast_all<1>(xh);
        double sn, cs;
        pend_sincos(trig, x0v, sn, cs);
        double ul = uu * il;
        ul = __builtin_fma(__builtin_fma(-ul, l, uu), il, ul);
        double acc = -gl * sn + ul * cs - dd * x1v;
        asm("" : "+v"(acc));
        const double nxt = __builtin_amdgcn_update_dpp(0.0, xh, 0xf9, 0xf, 0xf, true);
        const double inc = is1 ? acc : (is3 ? uu : nxt);
        xh = xh + h * inc;
        dpp_fence(xh);
    };
    for (int i0 = 0; i0 < steps; i0 += D) {
        if (FUSE) ctw = ct + j + (i0 & 8) * TS;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            double ld = ring[d];
            asm volatile("" : "+v"(ld));
            step(i0 + d, ld);
            if (FETCH) ring[d] = src[64 * min(i0 + d + D, steps - 1)];
        }
        if (FUSE && (i0 & 8)) {
            wave_sync();
            double c = 0.0;
#pragma unroll
            for (int q = 0; q < n + 1; ++q) c += ct[j * TS + q];
            cacc += c;
            wave_sync();
        }
    }
    out[wave * 64 + lane] = xh + cacc;
}

template <int MODE, int WPB> void run_mw(int waves)
{
    const int steps = 600, reps = 20;
    double *d, *stream, *sink;
    const size_t nstream = (size_t)(waves / 2 + 1) * steps * 64, nsink = (size_t)waves * 4 * steps * 5 + 64;
    (void)hipMalloc(&d, (size_t)waves * 64 * 8); (void)hipMalloc(&stream, nstream * 8); (void)hipMalloc(&sink, nsink * 8);
    (void)hipMemset(stream, 0, nstream * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) chain_mw<MODE, WPB><<<waves / WPB, WPB * 64>>>(steps, d, stream, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) chain_mw<MODE, WPB><<<waves / WPB, WPB * 64>>>(steps, d, stream, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  mode %d  %d waves per work-group  %5d waves (%5d rollouts): %7.1f us per launch  %6.1f ns / step\n", MODE, WPB, waves, waves * 4,
           ms * 1e3 / reps, ms * 1e6 / reps / steps);
    (void)hipFree(d); (void)hipFree(stream); (void)hipFree(sink);
}

template <int MODE> void run(const char *name)
{
    const int steps = 200000;
    double *d, *stream, *sink; long long *t, ht;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&t, 8); (void)hipMalloc(&stream, 64 * 64 * 8); (void)hipMalloc(&sink, 1024 * 8);
    {
        static double hs[64 * 64];
        for (int s = 0; s < 64; ++s) for (int lane = 0; lane < 64; ++lane) { const int j = lane % 16; hs[64 * s + lane] = j < 4 ? 0.1 * (j + 1) : (j < 8 ? (j == 4 ? 3.0 : 0.0) : (j == 8 ? 0.2 : (j == 9 ? -0.1 : 0.0))); }
        (void)hipMemcpy(stream, hs, sizeof hs, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    chain<MODE><<<1, 64>>>(2000, d, t, stream, sink); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); chain<MODE><<<1, 64>>>(steps, d, t, stream, sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
    double h0; (void)hipMemcpy(&h0, d, 8, hipMemcpyDeviceToHost);
    printf("%-78s %7.1f ns / step  %7.1f ticks / step   (check %.6g)\n", name, ms * 1e6 / steps, (double)ht / steps, h0);
    (void)hipFree(d); (void)hipFree(t); (void)hipFree(stream); (void)hipFree(sink);
}
}   // namespace

int main()
{
    run<0>("mode 0: broadcasts, sin / cos, acceleration, Euler step");
    run<1>("mode 1: + control law (row_shl, product, broadcast fma, 4-term row sum), clamp, NaN test");
    run<2>("mode 2: + result store (8 bytes per lane and step)");
    run<3>("mode 3: + fused cost (LDS tile write per step, sum + flush every 16 steps)");
    run<4>("mode 4: + operand stream (one load per lane and step, eight steps ahead)");
    printf("the same step on the whole device (N = 600, launch overhead included):\n");
    for (int waves : {256, 512, 640, 768, 1024, 2048}) {
        run_mw<1, 1>(waves); run_mw<2, 1>(waves); run_mw<3, 1>(waves); run_mw<4, 1>(waves); run_mw<4, 4>(waves);
    }
    printf("production forward_pend_row_kernel<POLICY, LIMS, FUSE> at C3 (N = 600, 4 096 rollouts, one wave per SIMD): 0.186 ms / 600 steps = 310 ns per step\n");
    return 0;
}
