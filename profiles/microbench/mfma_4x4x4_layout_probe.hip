// Probe (gfx950): operand layout of v_mfma_f64_4x4x4_4b, asked because a 16-lane row sum by two such MFMAs would have
// replaced the 20 DPP broadcast-FMAs of K·dx in the rollout kernel.  Result: the four "blocks" are INTERLEAVED —
//   block b = (lane/4)%4;  A[i][k]: i = lane%4, k = lane/16;  B[k][j]: k = lane/16, j = lane%4;  D[i][j]: lane = 16 i + 4 b + j
// — so the contraction index k runs ACROSS the four 16-lane rows of a wave, not inside one: it cannot sum a DPP row.
// (Two dependent 4x4x4 MFMAs cost ~94 cycles.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
template <int MODE>
__device__ __forceinline__ double block_sum(double v)
{
    const double one = 1.0;
    double d1 = (MODE & 1) ? __builtin_amdgcn_mfma_f64_4x4x4f64(one, v, 0.0, 0, 0, 0) : __builtin_amdgcn_mfma_f64_4x4x4f64(v, one, 0.0, 0, 0, 0);
    return (MODE & 2) ? __builtin_amdgcn_mfma_f64_4x4x4f64(d1, one, 0.0, 0, 0, 0) : __builtin_amdgcn_mfma_f64_4x4x4f64(one, d1, 0.0, 0, 0, 0);
}
template <int MODE> __global__ void probe(double *out, int l0)
{
    const int l = threadIdx.x;
    out[l] = block_sum<MODE>(l0 < 0 ? 1.0 + l * 0.37 : (l == l0 ? 1.0 : 0.0));
}
__global__ void single(double *out, int l0, int asA)
{
    const int l = threadIdx.x; const double v = (l == l0) ? 1.0 : 0.0;
    out[l] = asA ? __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0) : __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, v, 0.0, 0, 0, 0);
}
int main()
{
    double *d; (void)hipMalloc(&d, 64 * 8); double h[64];
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 0) probe<0><<<1, 64>>>(d, -1); if (mode == 1) probe<1><<<1, 64>>>(d, -1); if (mode == 2) probe<2><<<1, 64>>>(d, -1); if (mode == 3) probe<3><<<1, 64>>>(d, -1);
        (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int b = 0; b < 4; ++b) { double s = 0; for (int l = 16 * b; l < 16 * b + 16; ++l) s += 1.0 + l * 0.37; for (int l = 16 * b; l < 16 * b + 16; ++l) worst = fmax(worst, fabs(h[l] - s) / s); }
        printf("mode %d: worst deviation %.2e   lane0 %.3f lane17 %.3f (block sums %.3f %.3f)\n", mode, worst, h[0], h[17], 16 + 0.37 * 120, 16 + 0.37 * (120 + 256));
    }
    for (int asA = 1; asA >= 0; --asA) for (int l0 : {0, 1, 4, 5, 16, 21}) {
        single<<<1, 64>>>(d, l0, asA); (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%s e_%d -> nonzero D lanes:", asA ? "A" : "B", l0); for (int l = 0; l < 64; ++l) if (h[l] != 0) printf(" %d", l); printf("\n");
    }
    return 0;
}
