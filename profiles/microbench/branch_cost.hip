// Cost of a conditional branch for a lone wave (one wave per SIMD): not taken / taken forward, between dependent v_fma_f64.
//   hipcc --offload-arch=gfx950 -O3 branch_cost.hip -o branch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(int iters, int flag, double *out, long long *cyc)
{
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.999999, z = 1e-9;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
            if (MODE == 1) asm volatile("s_cmp_eq_u32 %3, 12345\n s_cbranch_scc1 1f\n v_fma_f64 %0, %0, %1, %2\n1:" : "+v"(x) : "v"(y), "v"(z), "s"(flag) : "scc");            // never taken
            if (MODE == 2) asm volatile("s_cmp_eq_u32 %3, 0\n s_cbranch_scc1 1f\n v_fma_f64 %0, %0, %1, %1\n1:\n v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z), "s"(flag) : "scc");   // always taken (skips one instruction)
            if (MODE == 3) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_cbranch_vccz 1f\n v_fma_f64 %0, %0, %1, %2\n1:" : "+v"(x) : "v"(y), "v"(z) : "vcc");                                      // VALU compare -> branch, not taken
            if (MODE == 4) asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_fma_f64 %0, %0, %1, %2\n1:\n s_or_b64 exec, exec, s[20:21]" : "+v"(x) : "v"(y), "v"(z) : "vcc", "s20", "s21");   // divergent-if idiom, body executed
            if (MODE == 5) asm volatile("s_cmp_eq_u32 %3, 12345\n s_cselect_b32 s20, 1, 0\n v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z), "s"(flag) : "scc", "s20");                  // the two SALU without a branch
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int iters = 1000;
    const char *names[] = {"v_fma_f64 (dependent)", "+ s_cmp, s_cbranch_scc1 never taken", "+ s_cmp, s_cbranch_scc1 always taken (skips 1 instr)",
                           "+ v_cmp, s_cbranch_vccz not taken", "+ v_cmp, s_and_saveexec, s_cbranch_execz (not taken), s_or exec", "+ s_cmp, s_cselect (no branch)"};
#define RUN(M) for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, iters, 0, out, cyc); hipDeviceSynchronize(); } \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-70s %7.2f cycles per group\n", names[M], (double)h / (iters * 32.0));
    RUN(0) RUN(1) RUN(2) RUN(5)
    return 0;
}
