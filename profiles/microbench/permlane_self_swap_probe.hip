// Is `v_permlane16_swap_b32 v, v` (the SAME register as both operands) an in-register exchange of the 16-lane rows 0<->1, 2<->3?
// (two distinct registers: odd rows of vdst are swapped with even rows of vsrc — permlane_swap_probe.hip)
//   hipcc --offload-arch=gfx950 -O3 permlane_self_swap_probe.hip -o permlane_self_swap_probe && ./permlane_self_swap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out, long long *cyc)
{
    unsigned v = threadIdx.x;
    asm volatile("v_permlane16_swap_b32 %0, %0" : "+v"(v));
    out[threadIdx.x] = v;
    // dependent-chain latency of the self swap
    unsigned w = threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("v_permlane16_swap_b32 %0, %0" : "+v"(w));
    long long t1 = __builtin_amdgcn_s_memtime();
    out[64 + threadIdx.x] = w;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    unsigned *d, h[128];
    long long *c, hc;
    hipMalloc(&d, sizeof(h)); hipMalloc(&c, 8);
    k<<<1, 64>>>(d, c);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) { const unsigned want = l ^ 16; if (h[l] != want) ok = 0; }
    printf("self swap = exchange of rows 0<->1, 2<->3: %s\n", ok ? "YES" : "NO");
    for (int l = 0; l < 64; l += 16) printf("  lanes %2d..: %u %u %u ...\n", l, h[l], h[l + 1], h[l + 2]);
    int ok2 = 1;
    for (int l = 0; l < 64; ++l) if (h[64 + l] != (unsigned)l) ok2 = 0;
    printf("64 dependent self swaps: identity %s, %.1f s_memtime ticks each\n", ok2 ? "yes" : "NO", hc / 64.0);
    return 0;
}
