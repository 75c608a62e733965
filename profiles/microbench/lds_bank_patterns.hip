// Microbenchmark (gfx950): cycles per ds_read_b64 / ds_write_b64 of one wave for the lane -> address patterns the n = 64 matrix-core
// backward kernel uses (address in doubles = sa * (lane & 15) + sb * (lane >> 4)), against the conflict-free pattern `lane`.
// Round 5's PMC run reported SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 0.50 for that kernel; this says WHICH of its patterns conflict.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bank lds_bank_patterns.hip && /tmp/lds_bank
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
#define UNR 16
__global__ __launch_bounds__(256) void bench(double *out, long long *cyc, int sa, int sb, int write, int waves)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 16384; e += blockDim.x) lds[e] = e * 1e-3;
    __syncthreads();
    if (wv >= waves) return;
    const int base = (sa * (lane & 15) + sb * (lane >> 4)) & 8191;
    double acc = 0.0;
    const double *p = lds + base + 2048 * (wv & 3);
    double *q = lds + base + 2048 * (wv & 3);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (write) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) { q[(u & 3)] = acc + u; asm volatile("" ::: "memory"); }
        } else {
            // 16 independent reads in flight, ONE wait: the issue rate of the LDS pipe, not its latency
            double v[UNR];
            const unsigned ad = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)p;
#pragma unroll
            for (int u = 0; u < UNR; ++u) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[u]) : "v"(ad), "n"((u & 3) * 8));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (it == ITERS - 1) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) acc += v[u];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    double *d; long long *c, hc[4];
    (void)hipMalloc(&d, 256 * 8 * 4); (void)hipMalloc(&c, 4 * 8 * 4);
    (void)hipFuncSetAttribute((const void *)bench, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    struct P { const char *name; int sa, sb; } pats[] = {
        {"lane (l15 + 16 l4): ideal", 1, 16}, {"Vs A read   l15 + 65 l4", 1, 65}, {"Fs k-major  66 l15 + l4", 66, 1}, {"WT read     l15 + 80 l4", 1, 80},
        {"Ks / Ys     10 l15 + l4", 10, 1}, {"bad         64 l15 + l4", 64, 1}, {"            65 l15 + l4", 65, 1}, {"            68 l15 + l4", 68, 1},
        {"            72 l15 + l4", 72, 1}, {"            34 l15 + l4", 34, 1}, {"Fs tile     66 l15 + 4 l4", 66, 4}, {"Vs mirror   65 l15 + l4", 65, 1},
        {"Vs qp       l15 + 260 l4", 1, 260}, {"PT          l15 + 9 l4", 1, 9}, {"            9 l15 + l4 (K 9)", 9, 1}, {"            12 l15 + l4", 12, 1},
        {"            l15 + 66 l4", 1, 66}, {"            l15 + 68 l4", 1, 68}, {"            2 l15 + l4 ... (b64 pairs)", 2, 1}, {"            17 l15 + l4", 17, 1}};
    for (int write = 0; write < 2; ++write)
        for (int waves = 1; waves <= 4; waves += 3)
            for (auto &p : pats) {
                bench<<<1, 256, 131072>>>(d, c, p.sa, p.sb, write, waves);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(hc, c, sizeof hc, hipMemcpyDeviceToHost);
                printf("%s waves=%d  %-40s %.2f ticks per instruction (wave 0)\n", write ? "ds_write_b64" : "ds_read_b64 ", waves, p.name,
                       (double)hc[0] / ((double)ITERS * UNR));
            }
    return 0;
}
