// Single-wave latency probe for the instruction kinds on the dependent chain of back_pass_mx.hip (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 chain_latency.hip -o chain_latency && ./chain_latency
// Prints shader-clock cycles per operation for DEPENDENT chains (latency) and independent streams (issue rate),
// measured with s_memtime by one wavefront on an otherwise idle GPU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}
#define REP 64
#define PIN() asm volatile("" : "+v"(x), "+v"(y), "+v"(u), "+v"(acc) : "s"(t0))

__global__ void probe(double *out, unsigned long long *cyc, double seed)
{
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3, y = 1.0 + seed, z = 0.5;
    unsigned long long t0, t1;
    int q = 0;
    unsigned u = (unsigned)lane;
    d4 acc = d4{x, x, x, x};
    // 0: dependent v_fma_f64
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) x = fma(x, y, z);
    asm volatile("" : "+v"(x));
    t1 = now(); cyc[q++] = t1 - t0;
    // 1: 8 independent fma chains
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = x + j;
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP / 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fma(a[j], y, z);
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(a[j]));
    t1 = now(); cyc[q++] = t1 - t0;
#pragma unroll
    for (int j = 0; j < 8; ++j) x += a[j];
    // 2: dependent MFMA accumulate chain
    acc = d4{x, x, x, x};
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
    asm volatile("" : "+v"(acc));
    t1 = now(); cyc[q++] = t1 - t0;
    // 3: MFMA whose B operand is the previous MFMA's result (GEMM1 -> GEMM2 pattern)
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, acc.x, d4{0, 0, 0, 0}, 0, 0, 0);
    asm volatile("" : "+v"(acc));
    t1 = now(); cyc[q++] = t1 - t0;
    // 4: MFMA -> one VALU op on the result -> MFMA
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0); acc.x = acc.x + acc.y; }
    asm volatile("" : "+v"(acc));
    t1 = now(); cyc[q++] = t1 - t0;
    x += acc.x + acc.y + acc.z + acc.w;
    // 5: dependent v_rcp_f64
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rcp(x);
    asm volatile("" : "+v"(x));
    t1 = now(); cyc[q++] = t1 - t0;
    // 6: dependent DPP row broadcast (v_mov_b64_dpp)
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_update_dpp(0.0, x, 0x150 + 5, 0xf, 0xf, true);
    asm volatile("" : "+v"(x));
    t1 = now(); cyc[q++] = t1 - t0;
    // 7: dependent permlane32_swap + permlane16_swap on one dword
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP / 2; ++i) {
        u2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        u2v s = __builtin_amdgcn_permlane16_swap(r.y, r.y, false, false);
        u = s.x ^ s.y;
    }
    asm volatile("" : "+v"(u));
    t1 = now(); cyc[q++] = t1 - t0;
    // 8: dependent v_cndmask_b32 pair (f64 select)
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) { x = (lane & 1) ? x : y; asm volatile("" : "+v"(x)); y = (lane & 2) ? y : x; asm volatile("" : "+v"(y)); }
    t1 = now(); cyc[q++] = t1 - t0;
    // 9: dependent v_add_f64
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP; ++i) x = x + y;
    asm volatile("" : "+v"(x));
    t1 = now(); cyc[q++] = t1 - t0;
    // 10: LDS write -> read round trip (wave-private)
    __shared__ double sh[128];
    t0 = now(); PIN();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) { sh[lane] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = sh[lane ^ 1] + 1.0; }
    asm volatile("" : "+v"(x));
    t1 = now(); cyc[q++] = t1 - t0;
    // 12: shader clock vs the 100 MHz real-time counter over a long dependent chain
    {
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t0 = now(); PIN();
        for (int i = 0; i < 20000; ++i) { x = fma(x, y, z); asm volatile("" : "+v"(x)); }
        t1 = now();
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cyc[12] = t1 - t0; cyc[13] = r1 - r0;
    }
    // 14: dependent MFMA chain with 12 independent v_fma_f64 between consecutive MFMAs (do they hide under the MFMA?)
    {
        double bq[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) bq[j] = x + j;
        t0 = now(); PIN();
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 12; ++j) bq[j] = fma(bq[j], y, z);
        }
        asm volatile("" : "+v"(acc));
#pragma unroll
        for (int j = 0; j < 12; ++j) asm volatile("" : "+v"(bq[j]));
        t1 = now(); cyc[14] = t1 - t0;
#pragma unroll
        for (int j = 0; j < 12; ++j) x += bq[j];
    }
    // 15: one 64-lane global_store_dwordx2 (per-lane 64-bit addresses, 32-byte pieces) per 8 dependent v_fma_f64
    {
        double *dst = out + 64 + (lane & 3) + 10 * (lane >> 2);
        t0 = now(); PIN();
#pragma unroll
        for (int i = 0; i < REP / 4; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x = fma(x, y, z);
            dst[160 * i] = x;
        }
        asm volatile("" : "+v"(x));
        t1 = now(); cyc[15] = t1 - t0;
    }
    // 11: empty timer
    t0 = now(); t1 = now(); cyc[q++] = t1 - t0;
    out[lane] = x + y + (double)u;
}

int main()
{
    double *o; unsigned long long *c, h[16];
    double *obig;
    (void)hipMalloc(&o, (64 + 160 * 20 + 200) * sizeof(double)); (void)hipMalloc(&c, sizeof(h));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o, c, 0.25);
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[12] = {"dependent v_fma_f64", "8 independent v_fma_f64 chains", "dependent MFMA f64 16x16x4 (accumulate)",
                          "MFMA -> MFMA through the B operand", "MFMA -> v_add_f64 -> MFMA", "dependent v_rcp_f64",
                          "dependent v_mov_b64_dpp (row_newbcast)", "dependent permlane32_swap + permlane16_swap (pair)",
                          "dependent f64 select (2 x v_cndmask_b32), x2", "dependent v_add_f64", "LDS write -> hand-off -> read", "timer overhead"};
    const int cnt[12] = {REP, REP, REP, REP, REP, REP, REP, REP / 2, REP, REP, REP / 4, 1};
    for (int q = 0; q < 12; ++q) printf("%-52s %8llu cycles total, %7.1f per op\n", nm[q], h[q], (double)(h[q] - h[11]) / cnt[q]);
    printf("MFMA + 12 independent v_fma_f64: %.1f cycles per group (MFMA alone 63, 12 fma alone ~50)\n", (double)(h[14] - h[11]) / (REP / 4));
    printf("8 dependent v_fma_f64 + 1 global_store_dwordx2: %.1f cycles per group (8 fma alone ~41)\n", (double)(h[15] - h[11]) / (REP / 4));
    printf("s_memtime ticks %llu over %llu ticks of the 100 MHz counter -> s_memtime runs at %.1f MHz\n", h[12], h[13], 100.0 * (double)h[12] / (double)h[13]);
    return 0;
}
