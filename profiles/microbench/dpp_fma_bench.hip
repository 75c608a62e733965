// Microbenchmark (gfx950): issue cost of fp64 FMA forms used by the DPP-broadcast kernels.
//   hipcc --offload-arch=gfx950 -O3 dpp_fma_bench.hip -o dpp_fma_bench && ./dpp_fma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 20000
template <int MODE>
__global__ __launch_bounds__(64) void bench(double *out, double s)
{
    double acc[16], x = threadIdx.x * 1e-3 + s, y = 1.0 + 1e-9 * threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
            if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(x), "v"(y));
            if (MODE == 2) { double t; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(x)); acc[i] += t; }
            if (MODE == 3) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[0]) : "v"(x), "v"(y));     // dependent chain
            if (MODE == 4) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[0]) : "v"(x), "v"(y));
        }
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, int blocks)
{
    double *d; hipMalloc(&d, blocks * 64 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    bench<MODE><<<blocks, 64>>>(d, 0.5); hipDeviceSynchronize();
    hipEventRecord(a); bench<MODE><<<blocks, 64>>>(d, 0.5); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ns_per = ms * 1e6 / ((double)ITERS * 16);
    printf("%-28s blocks=%5d  %.3f ms  %.2f ns per wave-instruction (x2.4 GHz = %.1f cycles)\n", name, blocks, ms, ns_per, ns_per * 2.4);
    hipFree(d);
}
int main()
{
    for (int blocks : {1024, 4096}) {
        run<0>("v_fmac_f64 indep", blocks);
        run<1>("v_fmac_f64_dpp newbcast indep", blocks);
        run<2>("v_mov_b64_dpp + add", blocks);
        run<3>("v_fmac_f64 dependent", blocks);
        run<4>("v_fmac_f64_dpp dependent", blocks);
    }
    return 0;
}
