// Do fp64 matrix instructions and fp64 vector instructions of ONE wave overlap?  Loop body: NM independent MFMAs (rotating accumulators)
// + NV independent v_fma_f64; cycles per iteration against the sum of the stand-alone costs.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KIND, int NM, int NV>
__global__ void k(int iters, double *out, long long *cyc)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999;
    double acc[4] = {0, 0, 0, 0};
    d4 big[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                if (KIND == 0 || KIND == 2) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[(q + r) & 3]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(big[(q + r) & 1]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (KIND < 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[q & 7]) : "v"(b), "v"(a));
                else { float f = (float)x[q & 7]; asm volatile("v_fma_f32 %0, %0, %0, %0\n v_mov_b32 %0, %0" : "+v"(f)); x[q & 7] = f; }    // KIND 2, 3: 32-bit vector work
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int q = 0; q < 8; ++q) s += x[q];
    out[threadIdx.x] = s + acc[0] + acc[1] + acc[2] + acc[3] + big[0].x + big[1].y;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
#define RUN(K, NM, NV) for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<K, NM, NV>), dim3(1), dim3(64), 0, 0, iters, out, cyc); hipDeviceSynchronize(); } \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-8s MFMAs %d + v_fma_f64 %d per group: %6.1f cycles per group\n", (K == 0 || K == 2) ? "4x4x4" : "16x16x4", NM, NV, (double)h / (iters * 4.0));
    RUN(0, 1, 0) RUN(0, 0, 2) RUN(0, 1, 2) RUN(0, 1, 4) RUN(0, 2, 0) RUN(0, 2, 4) RUN(0, 0, 4) RUN(0, 0, 8) RUN(0, 1, 8)
    RUN(1, 1, 0) RUN(1, 1, 4) RUN(1, 1, 8)
    printf("-- the same with 32-bit vector instructions (v_fma_f32 + v_mov_b32 per unit, plus conversions) beside the fp64 MFMA\n");
    RUN(2, 0, 4) RUN(2, 1, 0) RUN(2, 1, 4) RUN(3, 1, 0) RUN(3, 1, 4) RUN(3, 0, 4)
    return 0;
}
