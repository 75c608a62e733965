// Why the pendulum rollout (csrc/forward_pass_dpp.hip, forward_pend_row_kernel) stops scaling at ~1.65 TB/s: the memory side of its step
// alone, no arithmetic.  One 16-lane row per rollout, four rollouts per wave, R rollouts in lock step over N steps; per rollout and step
// K_i (32 B), x_i (32 B), u_i (8 B), k_i (8 B) are read and x_i (32 B), u_i (8 B) written: 120 B, the arrays [., N, R] as the library has them.
//   mode 0  production's pattern: ONE 8-byte load per lane and step (lanes 0-3 K, 4-7 x, 8 u, 9 k), eight steps in flight per lane, one
//           8-byte store per lane and step (lanes 0-4)
//   mode 1  the same bytes in 512-byte runs: every 16 steps a lane fetches 32 B of K and of x, 8 B of u and of k (16 steps of the stream),
//           one chunk ahead; stores likewise
// Prints GB/s of the 120 B per rollout-step.   hipcc --offload-arch=gfx950 -O3 -o narrow_streams narrow_streams.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void streams(int N, int R, const double *K, const double *x, const double *u, const double *k, double *xn, double *un, double *sink)
{
    const int lane = threadIdx.x, j = lane & 15;
    const long r = (long)blockIdx.x * 4 + (lane >> 4);
    if (r >= R) return;
    double acc = 0.0;
    if (MODE == 0) {
        constexpr int D = 8;
        const double *src = j < 4 ? K + 4l * N * r + j : (j < 8 ? x + 4l * N * r + (j - 4) : (j == 9 ? k + (long)N * r : u + (long)N * r));
        const int ss = j < 8 ? 4 : (j < 10 ? 1 : 0);
        double *dst = j < 4 ? xn + 4l * N * r + j : (j == 4 ? un + (long)N * r : sink + lane);
        const int ds = j < 4 ? 4 : (j == 4 ? 1 : 0);
        double ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ring[d] = src[(long)ss * d];
        for (int i0 = 0; i0 < N; i0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int i = i0 + d;
                const double v = ring[d];
                acc += v;
                dst[(long)ds * i] = v;
                const int in = i + D < N ? i + D : N - 1;
                ring[d] = src[(long)ss * in];
            }
        }
    } else {
        const d4 *K4 = (const d4 *)(K + 4l * N * r) + j, *x4 = (const d4 *)(x + 4l * N * r) + j;      // step 16 c + j of chunk c
        const double *u1 = u + (long)N * r + j, *k1 = k + (long)N * r + j;
        d4 *xn4 = (d4 *)(xn + 4l * N * r) + j;
        double *un1 = un + (long)N * r + j;
        d4 a = K4[0], b = x4[0];
        double c = u1[0], e = k1[0];
        for (int ch = 0; ch < N / 16; ++ch) {
            const int nx = ch + 1 < N / 16 ? ch + 1 : ch;
            const d4 a2 = K4[16 * nx], b2 = x4[16 * nx];
            const double c2 = u1[16 * nx], e2 = k1[16 * nx];
            acc += a.x + a.w + e;
            xn4[16 * ch] = b + a;
            un1[16 * ch] = c;
            a = a2; b = b2; c = c2; e = e2;
        }
    }
    if (acc == 123.456) sink[0] = acc;
}
template <int MODE> void run(int R)
{
    const int N = 592, reps = 20;
    double *K, *x, *u, *k, *xn, *un, *sink;
    (void)hipMalloc(&K, 32l * N * R); (void)hipMalloc(&x, 32l * N * R); (void)hipMalloc(&u, 8l * N * R); (void)hipMalloc(&k, 8l * N * R);
    (void)hipMalloc(&xn, 32l * N * R); (void)hipMalloc(&un, 8l * N * R); (void)hipMalloc(&sink, 4096);
    (void)hipMemset(K, 0, 32l * N * R); (void)hipMemset(x, 0, 32l * N * R); (void)hipMemset(u, 0, 8l * N * R); (void)hipMemset(k, 0, 8l * N * R);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) streams<MODE><<<R / 4, 64>>>(N, R, K, x, u, k, xn, un, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) streams<MODE><<<R / 4, 64>>>(N, R, K, x, u, k, xn, un, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 120.0 * N * R;
    printf("mode %d  %5d rollouts x %d steps: %7.1f us per launch  %6.1f ns per step  %7.1f GB/s\n", MODE, R, N, ms * 1e3 / reps, ms * 1e6 / reps / N, bytes / (ms * 1e-3 / reps) * 1e-9);
    (void)hipFree(K); (void)hipFree(x); (void)hipFree(u); (void)hipFree(k); (void)hipFree(xn); (void)hipFree(un); (void)hipFree(sink);
}
int main()
{
    for (int R : {1024, 2048, 3072, 4096, 8192, 16384}) { run<0>(R); run<1>(R); }
    return 0;
}
