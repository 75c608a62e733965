// Microbenchmark (gfx950): issue cost of v_mfma_f64_16x16x4_f64 (independent accumulators / dependent chain).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 20000
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void bench(double *out, double s)
{
    double a = threadIdx.x * 1e-3 + s, b = 1.0 + 1e-9 * threadIdx.x;
    d4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
        }
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) t += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 64 + threadIdx.x] = t;
}
template <int MODE> void run(const char *name, int blocks)
{
    double *d; (void)hipMalloc(&d, blocks * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    bench<MODE><<<blocks, 64>>>(d, 0.5); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); bench<MODE><<<blocks, 64>>>(d, 0.5); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double ns = ms * 1e6 / ((double)ITERS * 4);
    printf("%-30s blocks=%5d  %.3f ms  %.2f ns per MFMA per wave  -> %.1f TFLOP/s fp64\n", name, blocks, ms, ns,
           (double)blocks * ITERS * 4 * 2048 / (ms * 1e-3) / 1e12);
    (void)hipFree(d);
}
int main()
{
    for (int blocks : {1024, 2048, 4096}) {
        run<0>("mfma_f64_16x16x4 4 indep acc", blocks);
        run<1>("mfma_f64_16x16x4 dependent", blocks);
    }
    return 0;
}
