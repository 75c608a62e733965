// Probe of the gfx950 row-swap instructions used by back_pass_mx.hip: prints which lanes end up where.
//   hipcc --offload-arch=gfx950 -O3 permlane_swap_probe.hip -o permlane_swap_probe && ./permlane_swap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o)
{
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    u2 s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y; o[128 + threadIdx.x] = s.x; o[192 + threadIdx.x] = s.y;
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[4] = {"permlane32_swap.x", "permlane32_swap.y", "permlane16_swap.x", "permlane16_swap.y"};
    for (int q = 0; q < 4; ++q) {
        printf("%s (a = lane, b = 100+lane), first lane of each 16-row:", nm[q]);
        for (int r = 0; r < 4; ++r) printf(" %u", h[64 * q + 16 * r]);
        printf("\n");
    }
    return 0;
}
