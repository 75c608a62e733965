// How accurate is v_rcp_f64 on gfx950, and after one / two Newton steps y <- y + y(1 - x y)?  (The chain of sh_back_kernel / mx2 pays two
// steps = 4 dependent fp64 instructions per Riccati step for 1 / det(QuuF).)  Relative error against 1/x in long double on the host.
//   hipcc --offload-arch=gfx950 -O3 rcp_f64_accuracy.hip -o rcp_f64_accuracy && ./rcp_f64_accuracy
// MI355X: v_rcp_f64 4.6e-08 (2^-24.4), one Newton step 2.2e-15 (2^-48.7), two 1.1e-16 (2^-53.0): the second step is what makes 1/det
// correctly rounded; dropping it would save ~2 % of a chain step at 2e-15 per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__global__ void k(const double *x, double *y0, double *y1, double *y2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i];
    double y = __builtin_amdgcn_rcp(a);
    y0[i] = y;
    double e = fma(-a, y, 1.0); y = fma(y, e, y);
    y1[i] = y;
    e = fma(-a, y, 1.0); y = fma(y, e, y);
    y2[i] = y;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> hx(n), h0(n), h1(n), h2(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hx[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
    double *x, *y0, *y1, *y2;
    hipMalloc(&x, n * 8); hipMalloc(&y0, n * 8); hipMalloc(&y1, n * 8); hipMalloc(&y2, n * 8);
    hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, y0, y1, y2, n);
    hipMemcpy(h0.data(), y0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), y1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), y2, n * 8, hipMemcpyDeviceToHost);
    long double w0 = 0, w1 = 0, w2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double r = 1.0L / (long double)hx[i];
        w0 = fmaxl(w0, fabsl(((long double)h0[i] - r) / r)); w1 = fmaxl(w1, fabsl(((long double)h1[i] - r) / r)); w2 = fmaxl(w2, fabsl(((long double)h2[i] - r) / r));
    }
    printf("max relative error over %d arguments: v_rcp_f64 %.3Le (2^%.1Lf), one Newton step %.3Le (2^%.1Lf), two %.3Le (2^%.1Lf)\n", n, w0, log2l(w0), w1, log2l(w1), w2, log2l(w2));
    return 0;
}
