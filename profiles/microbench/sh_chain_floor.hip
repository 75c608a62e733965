// The floor of ONE step of the shared-LTI matrix chain (csrc/back_pass_sh.hip, sh_chain; src/backward_pass.jl:239-247, :41-42, :64-72) on one
// wave of gfx950, with everything that is not on the dependency chain taken away:
//   mode 0  the seven v_mfma_f64_16x16x4 in their dependency pattern (W = V F: 3 through the accumulator; G = F'W + H: 3, the first
//           through its B operand; V = G + K'Y: 1 through accumulator, A and B) — operands trivially derived from the previous product
//   mode 1  + the 2x2 gain solve as the kernel has it (row spread by v_permlane16_swap, three DPP broadcasts, det, v_rcp_f64 + two
//           Newton steps, numerators, selects, the B operand) between G and the update product
//   mode 2  + the step record (three LDS stores) and the divergence test
// Prints ns and shader-clock ticks (s_memtime) per step.  The production step of round 5 is ~360 ns: what is above mode 2 is the LDS
// symmetrisation every 8th step and the hand-over to the builder wave; what is below mode 0 needs another formulation (fewer than seven
// dependent fp64 matrix instructions per step), not a better schedule.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 sh_chain_floor.hip -o sh_chain_floor && ./sh_chain_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ double row_bcast(double x) { return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, true); }
__device__ __forceinline__ double rcp_nr(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ void spread_pair(double z, double &q0, double &q1)
{
    const unsigned lo = (unsigned)__double2loint(z), hi = (unsigned)__double2hiint(z);
    const u2v e = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const u2v f = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    q0 = __hiloint2double((int)f.x, (int)e.x);
    q1 = __hiloint2double((int)f.y, (int)e.y);
}

template <int MODE>
__global__ __launch_bounds__(64) void chain(int steps, double *out, long long *ticks, double lam)
{
    __shared__ double rec[8 * 160];
    const int lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
    const bool odd = l4 & 1, hi2 = l4 >= 2;
    // a contraction-like operand set so that the values stay finite: F ~ 0.25 I, H ~ I
    double F[3], S[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) { F[s] = (l4 + 4 * s == l15) ? 0.25 : 1e-3; S[s] = (l4 + 4 * s == l15) ? 1.0 : 0.0; }
    const d4 Hc = d4{l4 == l15 ? 1.0 : 0.0, l4 + 4 == l15 ? 1.0 : 0.0, l4 + 8 == l15 ? 1.0 : 0.0, (l15 == 10 + (l4 & 1)) ? 1.0 : 0.01};
    const d4 zero4 = d4{0, 0, 0, 0};
    const double chl = -0.5 * lam;
    int diverge = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < steps; ++i) {
        d4 w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[0], F[0], zero4, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[1], F[1], w, 0, 0, 0);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(S[2], F[2], w, 0, 0, 0);
        d4 g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], w.x, Hc, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], w.y, g, 0, 0, 0);
        g = __builtin_amdgcn_mfma_f64_16x16x4f64(F[2], w.z, g, 0, 0, 0);
        double Aop, Bop;
        const double Z = g.w + 0.0;
        if (MODE == 0) {
            Aop = Z; Bop = Z * 1e-3;
        } else {
            double Q0, Q1;
            spread_pair(Z, Q0, Q1);
            const double F00 = row_bcast<10>(Q0) + lam, F01 = row_bcast<11>(Q0), F11 = row_bcast<11>(Q1) + lam;
            const double det = fma(F00, F11, -(F01 * F01));
            const double y = rcp_nr(det);
            const double n0 = fma(F11, Q0, -(F01 * Q1)), n1 = fma(F00, Q1, -(F01 * Q0));
            const double Ksel = -((odd ? n1 : n0) * y);
            Bop = fma(chl, Ksel, 0.5 * Z);
            Aop = Ksel;
            if (MODE == 2) {
                const bool bad = (__builtin_amdgcn_ballot_w64(!(F00 > 0.0)) | __builtin_amdgcn_ballot_w64(!(det > 0.0))) != 0;
                const int bd = bad ? i + 1 : 0;
                diverge = diverge ? diverge : bd;
                rec[(i & 7) * 160 + 124 + lane % 24] = Z;
            }
        }
        const d4 v = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop, Bop, g, 0, 0, 0);
        S[0] = v.x; S[1] = v.y; S[2] = v.z;
        if (MODE == 2) {
            rec[(i & 7) * 160 + l4 + 10 * (l15 % 10)] = S[0];
            rec[(i & 7) * 160 + l4 + 4 + 10 * (l15 % 10)] = S[1];
            rec[(i & 7) * 160 + 100 + lane % 20] = hi2 ? Aop : S[2];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[lane] = S[0] + S[1] + S[2] + diverge + rec[lane];
    if (lane == 0) ticks[0] = t1 - t0;
}

template <int MODE>
void run(const char *name)
{
    double *out; long long *tk, h;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&tk, 8);
    const int steps = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    chain<MODE><<<1, 64>>>(steps, out, tk, 1.0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); chain<MODE><<<1, 64>>>(steps, out, tk, 1.0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.1f ns per step  %7.1f s_memtime ticks per step\n", name, ms * 1e6 / steps, (double)h / steps);
    (void)hipFree(out); (void)hipFree(tk);
}

int main()
{
    run<0>("mode 0: seven dependent fp64 MFMA 16x16x4 (3 + 3 + 1)");
    run<1>("mode 1: + the 2x2 gain solve between G and the update");
    run<2>("mode 2: + step record (LDS) and divergence test");
    return 0;
}
