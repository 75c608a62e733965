// Does a wave's fp64 VALU chain slow down when the other three SIMDs of its CU run fp64 MFMAs (the situation of the gain / boxQP
// wave of back_pass_mfma_kernel)?  One 256-thread work-group per CU: wave 0 runs a dependent v_fma_f64 chain (optionally with DPP
// broadcasts and s_nop fences like boxqp_rows.h), waves 1-3 idle / MFMA f64 16x16x4 / MFMA + LDS reads.
//   hipcc --offload-arch=gfx950 -O3 valu_beside_mfma.hip -o valu_beside_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int mode>
__global__ __launch_bounds__(256) void k(int iters, double *out, long long *cyc)
{
    __shared__ double sm[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 4096; e += 256) sm[e] = 1.0 + e * 1e-9;
    __syncthreads();
    if (wave == 0) {
        double x = 1.0 + lane * 1e-9, y = 0.999999, z = 1e-9;
        long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (mode & 8) asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(z));
                else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
            }
        }
        long long t1 = __builtin_readcyclecounter();
        out[blockIdx.x * 64 + lane] = x;
        if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    } else if (mode & 3) {
        d4 acc = {0, 0, 0, 0};
        double a = 1.0 + lane * 1e-9, b = 0.5;
        for (int i = 0; i < iters * 10; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if ((mode & 3) == 2) { a = sm[(lane * 17 + j * 64 + i) & 4095]; }
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
        }
        out[blockIdx.x * 64 + lane + 64 * 256 * wave] = acc.x + acc.y;
    }
}
int main()
{
    double *out; long long *cyc; long long h[256];
    hipMalloc(&out, 8 * 64 * 256 * 4); hipMalloc(&cyc, 8 * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode : {0, 1, 2, 8, 9, 10}) {
        float ms = 0;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, iters, out, cyc); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, iters, out, cyc); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, iters, out, cyc); break;
            case 8: hipLaunchKernelGGL(k<8>, dim3(256), dim3(256), 0, 0, iters, out, cyc); break;
            case 9: hipLaunchKernelGGL(k<9>, dim3(256), dim3(256), 0, 0, iters, out, cyc); break;
            default: hipLaunchKernelGGL(k<10>, dim3(256), dim3(256), 0, 0, iters, out, cyc); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(h, cyc, 8 * 256, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        printf("wave 0: %s chain, waves 1-3: %-22s  %6.2f counter ticks per instruction, kernel %.3f ms (%.2f ns per chain instruction if the chain is the critical path)\n",
               (mode & 8) ? "s_nop + v_fmac_f64_dpp" : "v_fma_f64", (mode & 3) == 0 ? "idle" : ((mode & 3) == 1 ? "MFMA f64 16x16x4" : "LDS read + MFMA"),
               s / 256 / (iters * 64.0), ms, 1e6 * ms / (iters * 64.0));
    }
    return 0;
}
