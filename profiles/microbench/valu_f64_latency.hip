// Dependent-chain latency and independent issue rate of the fp64 vector instructions the m = 1 boxQP fast path is made of,
// one wave per SIMD (the regime of back_pass_q4*).  hipcc --offload-arch=gfx950 -O3 valu_f64_latency.hip -o valu_f64_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 256
#define CHAIN(NAME, ...)                                                                            \
    __global__ void NAME(double *out, long long *cyc, double a, double b)                           \
    {                                                                                               \
        double x = a + threadIdx.x, y = b, z = a * 0.5, w = b + 1.0;                                \
        long long t0 = __builtin_readcyclecounter();                                                \
        _Pragma("unroll") for (int i = 0; i < REP; ++i) { __VA_ARGS__; }                                   \
        long long t1 = __builtin_readcyclecounter();                                                \
        out[threadIdx.x] = x + y + z + w;                                                           \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                     \
    }
CHAIN(k_fma_dep, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z)))
CHAIN(k_fma_ind, asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b)))
CHAIN(k_mul_dep, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_add_dep, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_max_dep, asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(y)))
CHAIN(k_rcp_dep, asm volatile("v_rcp_f64 %0, %0" : "+v"(x)))
CHAIN(k_rcp_ind, asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)))
CHAIN(k_cmp_sel, { float f = (float)x, g = (float)y, p = (float)z, q = (float)w; asm volatile("v_cmp_lt_f64 vcc, %4, %5\n v_cndmask_b32 %0, %2, %3, vcc\n v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %3, vcc" : "+v"(f) : "v"(g), "v"(p), "v"(q), "v"(x), "v"(y) : "vcc"); x = f; })
CHAIN(k_cmp_sand_sel, { float f = (float)x, g = (float)y, p = (float)z, q = (float)w; asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %2, %3, vcc\n v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %2, %3, vcc" : "+v"(f) : "v"(g), "v"(p), "v"(q) : "vcc"); x = f; })
CHAIN(k_salu_dep, asm volatile("s_and_b64 vcc, vcc, exec\n s_or_b64 vcc, vcc, exec" ::: "vcc"))
CHAIN(k_mov32_dep, { float f = (float)x; asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0" : "+v"(f)); x = f; })
CHAIN(k_fma32_dep, { float f = (float)x; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); x = f; })
CHAIN(k_mfma_dep, asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z)))
CHAIN(k_mfma_fma, asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n s_nop 3\n v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z)))
CHAIN(k_dsw_dsr, { __shared__ double sm[64]; sm[threadIdx.x] = x; x = sm[threadIdx.x ^ 1]; })

int main()
{
    double *out; long long *cyc, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
#define RUN(K, PER, WHAT) for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.9999999); hipDeviceSynchronize(); } \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-16s %7.2f cycles per %s\n", #K, (double)h / (REP * PER), WHAT);
    RUN(k_fma_dep, 1, "dependent v_fma_f64");
    RUN(k_fma_ind, 4, "independent v_fma_f64");
    RUN(k_mul_dep, 1, "dependent v_mul_f64");
    RUN(k_add_dep, 1, "dependent v_add_f64");
    RUN(k_max_dep, 1, "dependent v_max_f64");
    RUN(k_rcp_dep, 1, "dependent v_rcp_f64");
    RUN(k_rcp_ind, 4, "independent v_rcp_f64");
    RUN(k_cmp_sel, 2, "v_cmp -> v_cndmask round trip (+ cvt overhead / 2)");
    RUN(k_cmp_sand_sel, 2, "v_cmp -> s_and -> v_cndmask round trip (+ cvt overhead / 2)");
    RUN(k_salu_dep, 2, "dependent s_and_b64");
    RUN(k_mov32_dep, 4, "dependent v_mov_b32 (+ cvt overhead / 4)");
    RUN(k_fma32_dep, 1, "cvt + v_fma_f32 + cvt");
    RUN(k_mfma_dep, 1, "dependent v_mfma_f64_4x4x4 (C operand)");
    RUN(k_mfma_fma, 1, "v_mfma_f64_4x4x4 -> v_fma_f64 -> back");
    RUN(k_dsw_dsr, 1, "ds_write_b64 -> ds_read_b64 round trip");
    return 0;
}
