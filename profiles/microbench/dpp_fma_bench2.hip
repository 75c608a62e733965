// Microbenchmark 2 (gfx950): does operand variety (register-file banking) slow fp64 FMA / DPP-FMA issue?
//   P1-like stream: w[r] += bcast_l(V[r]) * F[l], r inner (10 accumulators), l outer — exactly the kernel's stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4000
template <int L> __device__ __forceinline__ void fmac_bc(double &acc, double s0, double s1)
{ asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s0), "v"(s1), "n"(L)); }
__device__ __forceinline__ void fmac(double &acc, double s0, double s1)
{ asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(s0), "v"(s1)); }
template <int I, int N, class F> __device__ __forceinline__ void sfor(F &&f)
{ if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); } }
template <int MODE>
__global__ __launch_bounds__(64) void bench(double *out, double s)
{
    double V[10], F[10], w[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { V[i] = threadIdx.x * 1e-3 + i + s; F[i] = 1.0 + 1e-9 * (threadIdx.x + i); w[i] = i; }
    for (int it = 0; it < ITERS; ++it) {
        sfor<0, 10>([&](auto lc) { constexpr int l = decltype(lc)::value;
            sfor<0, 10>([&](auto rc) { constexpr int r = decltype(rc)::value;
                if (MODE == 0) fmac_bc<l>(w[r], V[r], F[l]);
                if (MODE == 1) fmac(w[r], V[r], F[l]);
                if (MODE == 2) fmac_bc<l>(w[r], V[0], F[0]);       // same sources every time
                if (MODE == 3) fmac_bc<3>(w[r], V[r], F[l]);       // fixed broadcast lane
            }); });
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) t += w[i];
    out[blockIdx.x * 64 + threadIdx.x] = t;
}
template <int MODE> void run(const char *name, int blocks)
{
    double *d; (void)hipMalloc(&d, blocks * 64 * 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    bench<MODE><<<blocks, 64>>>(d, 0.5); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); bench<MODE><<<blocks, 64>>>(d, 0.5); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-44s blocks=%5d  %.3f ms  %.2f ns per wave-instruction\n", name, blocks, ms, ms * 1e6 / ((double)ITERS * 100));
    (void)hipFree(d);
}
int main()
{
    for (int blocks : {1024, 2048, 4096}) {
        run<0>("fmac_dpp w[r]+=bcast_l(V[r])*F[l]", blocks);
        run<1>("fmac     w[r]+=V[r]*F[l]", blocks);
        run<2>("fmac_dpp same sources", blocks);
        run<3>("fmac_dpp fixed lane, varying regs", blocks);
    }
    return 0;
}
