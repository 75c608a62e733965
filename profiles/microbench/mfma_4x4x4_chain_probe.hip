// Probe (gfx950): the algebra back_pass_q4.hip rests on.  With X[r][c] of block b held by lane 16 r + 4 b + c ("layout L",
// the D layout of v_mfma_f64_4x4x4_4b found by mfma_4x4x4_layout_probe.hip), a register used as the A operand is read
// TRANSPOSED (A[i][k] = lane 16 k + 4 b + i) and as the B operand as it is, so
//     mfma(X, Y, C) = X'·Y + C      per block, result again in layout L,
// products chain without moving data between lanes, mfma(X, I) = X' is an exact transposition, and a register that is zero
// outside rows 0 and 2 gives the rank-2 update  x0·y0' + x2·y2'.  Also times a dependent chain of such MFMAs.
//   hipcc --offload-arch=gfx950 -O3 mfma_4x4x4_chain_probe.hip -o mfma_4x4x4_chain_probe && ./mfma_4x4x4_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

__device__ __forceinline__ double mm(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

__global__ void probe(const double *X, const double *Y, const double *C, double *out)
{   // X, Y, C: [4 blocks][4][4] row-major; out[0]: X'Y + C, out[1]: X', out[2]: rank-2 of rows 0,2
    const int l = threadIdx.x, r = l / 16, b = (l / 4) % 4, c = l % 4;
    const double x = X[16 * b + 4 * r + c], y = Y[16 * b + 4 * r + c], cc = C[16 * b + 4 * r + c];
    out[64 * 0 + 16 * b + 4 * r + c] = mm(x, y, cc);
    out[64 * 1 + 16 * b + 4 * r + c] = mm(x, (r == c) ? 1.0 : 0.0, 0.0);
    const double xm = (r == 0 || r == 2) ? x : 0.0, ym = (r == 0 || r == 2) ? y : 0.0;
    out[64 * 2 + 16 * b + 4 * r + c] = mm(xm, ym, cc);
}

__global__ void chain(double *out, int reps, long long *cycles)
{
    double v = 1.0 + 1e-3 * threadIdx.x, f = 0.25;
    const long long t0 = clock64();
    for (int i = 0; i < reps; ++i) { v = mm(v, f, 0.0); v = mm(f, v, 0.0); }
    const long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

int main()
{
    double hX[64], hY[64], hC[64], hO[192];
    for (int i = 0; i < 64; ++i) { hX[i] = sin(1.0 + i); hY[i] = cos(2.0 + 0.7 * i); hC[i] = 0.01 * i; }
    double *dX, *dY, *dC, *dO; long long *dcy;
    (void)hipMalloc(&dX, sizeof hX); (void)hipMalloc(&dY, sizeof hY); (void)hipMalloc(&dC, sizeof hC); (void)hipMalloc(&dO, sizeof hO);
    (void)hipMalloc(&dcy, 8);
    (void)hipMemcpy(dX, hX, sizeof hX, hipMemcpyHostToDevice); (void)hipMemcpy(dY, hY, sizeof hY, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dX, dY, dC, dO);
    (void)hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = hC[16 * b + 4 * i + j], s2 = s;
                for (int k = 0; k < 4; ++k) {
                    s += hX[16 * b + 4 * k + i] * hY[16 * b + 4 * k + j];
                    if (k == 0 || k == 2) s2 += hX[16 * b + 4 * k + i] * hY[16 * b + 4 * k + j];
                }
                e0 = fmax(e0, fabs(hO[16 * b + 4 * i + j] - s));
                e1 = fmax(e1, fabs(hO[64 + 16 * b + 4 * i + j] - hX[16 * b + 4 * j + i]));
                e2 = fmax(e2, fabs(hO[128 + 16 * b + 4 * i + j] - s2));
            }
    printf("mfma(X,Y,C) - (X'Y+C): %.2e   mfma(X,I) - X': %.2e (must be 0)   rank-2 rows 0,2: %.2e\n", e0, e1, e2);
    const int reps = 10000;
    chain<<<1, 64>>>(dO, reps, dcy); (void)hipDeviceSynchronize();
    hipEvent_t e0_, e1_; (void)hipEventCreate(&e0_); (void)hipEventCreate(&e1_);
    (void)hipEventRecord(e0_, 0); chain<<<1, 64>>>(dO, reps, dcy); (void)hipEventRecord(e1_, 0); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0_, e1_);
    long long cy; (void)hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost);
    printf("dependent v_mfma_f64_4x4x4_4b chain: %.1f ns per MFMA (%.1f clock64 ticks)\n", 1e6 * ms / (2.0 * reps), (double)cy / (2.0 * reps));
    return (e0 < 1e-13 && e1 == 0.0 && e2 < 1e-13) ? 0 : 1;
}
