// df.hip — derivatives along the trajectory for the registered problem families: the `df` closure that
// STEP 1 of the reference's iteration calls (src/iLQG.jl:225-229).
//   LQ        cx = Q x, cu = R u                                   (src/demo_linear.jl:35-41)
//   pendcart  cx = Q (x - goal), cu = R u, and the ZoH discretisation of the continuous Jacobian by
//             exp of the 5x5 block matrix [fxc*h fuc*h; 0]          (src/system_pendcart.jl:112-116,137-154)
// One lane per (time step, trajectory): both are embarrassingly parallel, HBM-bound streams
// (pendcart adds ~1.5 kflop of 5x5 Padé arithmetic per element, still far below the fp64 ridge).
#include <stdlib.h>
#include "ddp_internal.h"

namespace {

// ---- dense matrix exponential of a D x D matrix held in registers --------------------------------
// Higham (2005) scaling & squaring with Padé 3/5/7/9/13 — the algorithm behind Julia's
// exp(::Matrix{Float64}) (without gebal balancing, which only changes rounding).
template <int D>
struct Mat {
    double a[D * D];
    __device__ double &operator()(int r, int c) { return a[r + D * c]; }
    __device__ double operator()(int r, int c) const { return a[r + D * c]; }
};

template <int D>
__device__ __forceinline__ void mmul(const Mat<D> &A, const Mat<D> &B, Mat<D> &C)
{
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += A(i, k) * B(k, j);
            C(i, j) = s;
        }
}

// solve A X = X in place (LU, partial pivoting; rows are swapped by value so indexing stays static)
template <int D>
__device__ __forceinline__ void gesv(Mat<D> &A, Mat<D> &X)
{
#pragma unroll
    for (int c = 0; c < D; ++c) {
        int p = c;
        double best = fabs(A(c, c));
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            const double v = fabs(A(r, c));
            if (v > best) { best = v; p = r; }
        }
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            if (p == r) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    double t = A(c, j); A(c, j) = A(r, j); A(r, j) = t;
                    t = X(c, j); X(c, j) = X(r, j); X(r, j) = t;
                }
            }
        }
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            const double f = A(r, c) / A(c, c);
            A(r, c) = 0.0;
#pragma unroll
            for (int j = c + 1; j < D; ++j) A(r, j) -= f * A(c, j);
#pragma unroll
            for (int j = 0; j < D; ++j) X(r, j) -= f * X(c, j);
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int r = D - 1; r >= 0; --r) {
            double s = X(r, j);
#pragma unroll
            for (int c = r + 1; c < D; ++c) s -= A(r, c) * X(c, j);
            X(r, j) = s / A(r, r);
        }
}

template <int D>
__device__ void expm_dev(const Mat<D> &Ain, Mat<D> &E)
{
    double nA = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) s += fabs(Ain(i, j));
        nA = fmax(nA, s);
    }
    Mat<D> A = Ain, A2, U, V, T;
    if (nA <= 2.1) {
        const double C9[10] = {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.};
        const double C7[8] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
        const double C5[6] = {30240., 15120., 3360., 420., 30., 1.};
        const double C3[4] = {120., 60., 12., 1.};
        double C[10];
        int nc;
#pragma unroll
        for (int i = 0; i < 10; ++i) C[i] = 0.0;
        if (nA > 0.95) { nc = 10;
#pragma unroll
            for (int i = 0; i < 10; ++i) C[i] = C9[i]; }
        else if (nA > 0.25) { nc = 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) C[i] = C7[i]; }
        else if (nA > 0.015) { nc = 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) C[i] = C5[i]; }
        else { nc = 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) C[i] = C3[i]; }
        mmul<D>(A, A, A2);
        Mat<D> P;
#pragma unroll
        for (int i = 0; i < D * D; ++i) { P.a[i] = 0.0; U.a[i] = 0.0; V.a[i] = 0.0; }
#pragma unroll
        for (int i = 0; i < D; ++i) { P(i, i) = 1.0; U(i, i) = C[1]; V(i, i) = C[0]; }
#pragma unroll
        for (int kk = 1; kk <= 4; ++kk) {                       // static indices keep C[] in registers
            if (kk <= nc / 2 - 1) {
                mmul<D>(P, A2, T);
                P = T;
#pragma unroll
                for (int i = 0; i < D * D; ++i) { U.a[i] += C[2 * kk + 1] * P.a[i]; V.a[i] += C[2 * kk] * P.a[i]; }
            }
        }
        mmul<D>(A, U, T);
        U = T;
#pragma unroll
        for (int i = 0; i < D * D; ++i) { E.a[i] = V.a[i] + U.a[i]; T.a[i] = V.a[i] - U.a[i]; }
        gesv<D>(T, E);
    } else {
        const double CC[14] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800.,
                               129060195264000., 10559470521600., 670442572800., 33522128640., 1323241920.,
                               40840800., 960960., 16380., 182., 1.};
        const double s = log2(nA / 5.4);
        int si = 0;
        if (s > 0) {
            si = (int)ceil(s);
            const double sc = ldexp(1.0, si);
#pragma unroll
            for (int i = 0; i < D * D; ++i) A.a[i] /= sc;
        }
        Mat<D> A4, A6, P;
        mmul<D>(A, A, A2); mmul<D>(A2, A2, A4); mmul<D>(A2, A4, A6);
#pragma unroll
        for (int i = 0; i < D * D; ++i) P.a[i] = CC[13] * A6.a[i] + CC[11] * A4.a[i] + CC[9] * A2.a[i];
        mmul<D>(A6, P, T);
#pragma unroll
        for (int i = 0; i < D * D; ++i) T.a[i] += CC[7] * A6.a[i] + CC[5] * A4.a[i] + CC[3] * A2.a[i];
#pragma unroll
        for (int i = 0; i < D; ++i) T(i, i) += CC[1];
        mmul<D>(A, T, U);
#pragma unroll
        for (int i = 0; i < D * D; ++i) P.a[i] = CC[12] * A6.a[i] + CC[10] * A4.a[i] + CC[8] * A2.a[i];
        mmul<D>(A6, P, V);
#pragma unroll
        for (int i = 0; i < D * D; ++i) V.a[i] += CC[6] * A6.a[i] + CC[4] * A4.a[i] + CC[2] * A2.a[i];
#pragma unroll
        for (int i = 0; i < D; ++i) V(i, i) += CC[0];
#pragma unroll
        for (int i = 0; i < D * D; ++i) { E.a[i] = V.a[i] + U.a[i]; T.a[i] = V.a[i] - U.a[i]; }
        gesv<D>(T, E);
        for (int q = 0; q < si; ++q) { mmul<D>(E, E, T); E = T; }
    }
}

// ---- exp of the ZoH block matrix [fxc*h fuc*h; 0] of a 4-state system: the LAST ROW IS ZERO, and so is the last row of every power.
// Same operations in the same order as expm_dev<5> on the rows 0..3 (the terms that drop out are products with exact zeros), Padé 3/5/7/9
// only (norm <= 2.1; the caller sends larger norms to the dense routine).  Five 4x5 matrices live instead of seven 5x5 ones, every product
// in place: 230 registers and no scratch against 256 + 168 bytes of scratch, i.e. two waves per SIMD instead of one.
struct M45 {
    double a[20];
    __device__ double &operator()(int r, int c) { return a[r + 4 * c]; }
    __device__ double operator()(int r, int c) const { return a[r + 4 * c]; }
};
__device__ __forceinline__ void expm5_zlast_small(const M45 &A, double nA, M45 &X)
{
    const double C9[10] = {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.};
    const double C7[8] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
    const double C5[6] = {30240., 15120., 3360., 420., 30., 1.};
    const double C3[4] = {120., 60., 12., 1.};
    double C[10];
    int nc;
#pragma unroll
    for (int i = 0; i < 10; ++i) C[i] = 0.0;
    if (nA > 0.95) { nc = 10;
#pragma unroll
        for (int i = 0; i < 10; ++i) C[i] = C9[i]; }
    else if (nA > 0.25) { nc = 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) C[i] = C7[i]; }
    else if (nA > 0.015) { nc = 6;
#pragma unroll
        for (int i = 0; i < 6; ++i) C[i] = C5[i]; }
    else { nc = 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) C[i] = C3[i]; }
    M45 A2, P, U, V;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += A(i, k) * A(k, j);          // (k = 4: A(4, j) = 0)
            A2(i, j) = s;
        }
#pragma unroll
    for (int i = 0; i < 20; ++i) { P.a[i] = 0.0; U.a[i] = 0.0; V.a[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { P(i, i) = 1.0; U(i, i) = C[1]; V(i, i) = C[0]; }
    const double u44 = C[1], v44 = C[0];                                  // the (4,4) entries: the powers add nothing to the last row
#pragma unroll
    for (int kk = 1; kk <= 4; ++kk) {
        if (kk <= nc / 2 - 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                 // P <- P A2, row by row in place
                double t[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s += P(i, k) * A2(k, j);
                    t[j] = s;
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) P(i, j) = t[j];
            }
#pragma unroll
            for (int i = 0; i < 20; ++i) { U.a[i] += C[2 * kk + 1] * P.a[i]; V.a[i] += C[2 * kk] * P.a[i]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {                                         // U <- A U, column by column in place
        double t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += A(i, k) * U(k, j);
            if (j == 4) s += A(i, 4) * u44;                               // U(4, 4)
            t[i] = s;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) U(i, j) = t[i];
    }
#pragma unroll
    for (int i = 0; i < 20; ++i) { const double u_ = U.a[i], v_ = V.a[i]; V.a[i] = v_ + u_; U.a[i] = v_ - u_; }     // V: right-hand side E, U: the matrix T
    // solve T X = E by LU with partial pivoting; row 4 of both is [0 0 0 0 v44]: never a pivot candidate, untouched by the elimination
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int pr = c;
        double best = fabs(U(c, c));
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const double v = fabs(U(r, c));
            if (v > best) { best = v; pr = r; }
        }
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            if (pr == r) {
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    double t = U(c, j); U(c, j) = U(r, j); U(r, j) = t;
                    t = V(c, j); V(c, j) = V(r, j); V(r, j) = t;
                }
            }
        }
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const double f = U(r, c) / U(c, c);
            U(r, c) = 0.0;
#pragma unroll
            for (int j = c + 1; j < 5; ++j) U(r, j) -= f * U(c, j);
#pragma unroll
            for (int j = 0; j < 5; ++j) V(r, j) -= f * V(c, j);
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const double x4 = (j == 4) ? v44 / v44 : 0.0 / v44;               // X(4, j)
#pragma unroll
        for (int r = 3; r >= 0; --r) {
            double s = V(r, j);
#pragma unroll
            for (int c = r + 1; c < 4; ++c) s -= U(r, c) * X(c, j);
            if (j == 4) s -= U(r, 4) * x4;                                // (j < 4: X(4, j) = 0)
            X(r, j) = s / U(r, r);
        }
    }
}

__global__ void df_lq_kernel(int n, int m, int N, int B, const double *Q, const double *R, const double *x,
                             const double *u, const int32_t *active, double *cx, double *cu)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;     // column index over (time, batch)
    if (t >= (long)N * B) return;
    const int b = (int)(t / N);
    if (active && active[b] == 0) return;
    const double *xc = x + (size_t)n * t, *uc = u + (size_t)m * t;
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += Q[i + n * j] * xc[j];
        cx[(size_t)n * t + i] = s;
    }
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int j = 0; j < m; ++j) {
            double uj = uc[j];
            if (uj != uj) uj = 0.0;                                   // u[isnan.(u)] .= 0
            s += R[i + m * j] * uj;
        }
        cu[(size_t)m * t + i] = s;
    }
}

// cx = Q x, cu = R u for n <= 64: Q and R live in LDS, a work-group walks tiles of 256/n trajectory columns — the x tile is
// read and the cx tile written as one contiguous 2 KB piece, thread (i, c) forms row i of column c (the one-thread-per-column
// kernel above reads and writes with a stride of n doubles between lanes: 8 ms for the 64 x 262 144 columns of BASELINE
// config 4, as long as the whole backward pass).  Same summation order as df_lq_kernel.
__global__ __launch_bounds__(256) void df_lq_tiled_kernel(int n, int m, int N, long cols, const double *Q, const double *R, const double *x,
                                                           const double *u, const int32_t *active, double *cx, double *cu)
{
    extern __shared__ double sm[];
    double *Qs = sm, *Rs = sm + n * n, *xs = Rs + m * m;
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += 256) Qs[e] = Q[e];
    for (int e = tid; e < m * m; e += 256) Rs[e] = R[e];
    const int cpb = 256 / n, i = tid % n, c = tid / n;
    const long ntiles = (cols + cpb - 1) / cpb;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long col = tile * cpb + c;
        const bool valid = c < cpb && col < cols;
        __syncthreads();                                           // Qs/Rs ready; previous tile's xs consumed
        if (valid) xs[tid] = x[(size_t)n * col + i];
        __syncthreads();
        if (valid && !(active && active[(int)(col / N)] == 0)) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += Qs[i + n * j] * xs[j + n * c];
            cx[(size_t)n * col + i] = s;
        }
    }
    const long nu = (long)m * cols;
    for (long e = (long)blockIdx.x * 256 + tid; e < nu; e += (long)gridDim.x * 256) {
        const long col = e / m;
        const int q = (int)(e % m);
        if (active && active[(int)(col / N)] == 0) continue;
        double s = 0.0;
        for (int j = 0; j < m; ++j) {
            double uj = u[(size_t)m * col + j];
            if (uj != uj) uj = 0.0;                                   // u[isnan.(u)] .= 0
            s += Rs[q + m * j] * uj;
        }
        cu[e] = s;
    }
}

// MODE 0: everything by the dense routine (the handle has no flag word).  MODE 1: cx, cu, and the exponential by expm5_zlast_small;
// an element whose norm asks for Padé 13 (> 2.1: controls in the hundreds — a diverging rollout) only raises *flag.  MODE 2: a fixed
// small grid that leaves at once unless the flag is up, then walks all elements and does those by the dense routine.
template <int MODE>
__global__ __launch_bounds__(64) void df_pendcart_kernel(int N, int B, double g, double l, double h, double d,
                                                         double g0, double g1, double g2, double g3, const double *Q,
                                                         const double *R, const double *x, const double *u,
                                                         const int32_t *active, double *cx, double *cu, double *fx,
                                                         double *fu, int32_t *flag, int v2)
{
    const long total = (long)N * B;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = MODE == 2 ? (long)gridDim.x * blockDim.x : total;
    if (MODE == 2 && *(volatile int32_t *)flag == 0) return;
    for (; t < total; t += stride) {
        const int b = (int)(t / N);
        if (active && active[b] == 0) continue;
        const double goal[4] = {g0, g1, g2, g3};
        double xv[4], dx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { xv[i] = x[4 * t + i]; dx[i] = xv[i] - goal[i]; }
        double uu = u[t];
        if (uu != uu) uu = 0.0;
        if (MODE != 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) s += Q[i + 4 * j] * dx[j];
                cx[4 * t + i] = s;                                        // system_pendcart.jl:113
            }
            cu[t] = R[0] * uu;                                            // :114
        }
        const double sn = sin(xv[0]), cs = cos(xv[0]);
        const double m01 = 1.0 * h, m10 = (-g / l * cs - uu / l * sn) * h, m11 = (-d) * h, m23 = 1.0 * h, m14 = (cs / l) * h, m34 = 1.0 * h;     // fxc*h, fuc*h (:130-148)
        // 1-norm of the block matrix: the largest column sum (columns 0, 1, 3, 4; column 2 is zero)
        const double nA = fmax(fmax(0.0 + fabs(m10), fabs(m01) + fabs(m11)), fmax(fabs(m23), fabs(m14) + fabs(m34)));
        if (MODE == 1) {
            if (!(nA <= 2.1)) { *flag = 1; continue; }
            M45 A, X;
#pragma unroll
            for (int i = 0; i < 20; ++i) A.a[i] = 0.0;
            A(0, 1) = m01; A(1, 0) = m10; A(1, 1) = m11; A(2, 3) = m23; A(1, 4) = m14; A(3, 4) = m34;
            expm5_zlast_small(A, nA, X);
            if (v2) {                                                     // 16-byte stores (the launcher has checked the alignment)
                typedef double d2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int e = 0; e < 20; e += 2) {
                    const d2v v = {X.a[e], X.a[e + 1]};
                    *(d2v *)((e < 16 ? fx + 16 * t : fu + 4 * t - 16) + e) = v;  // X is column-major 4 x 5: entries 0..15 = fx (:149), 16..19 = fu (:150)
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) fx[16 * t + r + 4 * c] = X(r, c);
#pragma unroll
                for (int r = 0; r < 4; ++r) fu[4 * t + r] = X(r, 4);
            }
        } else {
            if (MODE == 2 && nA <= 2.1) continue;
            Mat<5> M, E;
#pragma unroll
            for (int i = 0; i < 25; ++i) M.a[i] = 0.0;
            M(0, 1) = m01; M(1, 0) = m10; M(1, 1) = m11; M(2, 3) = m23; M(1, 4) = m14; M(3, 4) = m34;
            expm_dev<5>(M, E);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) fx[16 * t + r + 4 * c] = E(r, c);    // :149
#pragma unroll
            for (int r = 0; r < 4; ++r) fu[4 * t + r] = E(r, 4);                  // :150
        }
    }
}

}   // namespace

extern "C" {

int ddp_df_f64_dev(ddp_handle h, const ddp_problem *p, const double *x, const double *u, const int32_t *active,
                   double *cx, double *cu, double *fx, double *fu)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && p && x && u && cx && cu, "df: null argument");
    const long cols = (long)p->N * p->B;
    if (p->kind == DDP_PROBLEM_LQ) {
        if (p->n <= 64 && p->m <= 16) {
            const int cpb = 256 / p->n;
            const long ntiles = (cols + cpb - 1) / cpb;
            const dim3 grid((unsigned)(ntiles < 8192 ? ntiles : 8192)), block(256);
            const size_t shmem = ((size_t)p->n * p->n + (size_t)p->m * p->m + 256) * sizeof(double);
            hipLaunchKernelGGL(df_lq_tiled_kernel, grid, block, shmem, h->stream, p->n, p->m, p->N, cols, p->Q, p->R, x, u, active, cx, cu);
        } else {
            const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
            hipLaunchKernelGGL(df_lq_kernel, grid, block, 0, h->stream, p->n, p->m, p->N, p->B, p->Q, p->R, x, u, active, cx, cu);
        }
    } else if (p->kind == DDP_PROBLEM_PENDCART) {
        DDP_CHECK(p->n == 4 && p->m == 1, "df: pendcart needs n=4, m=1");
        DDP_CHECK(fx && fu, "df: pendcart needs fx and fu outputs");
        const dim3 grid((unsigned)((cols + 63) / 64)), block(64);
        const char *de = ddp_env(h, ENV_DF_DENSE);                     // 1: the dense expm for every element (A/B timing, cross-check in the tests)
        int32_t *flag = h->sink ? (int32_t *)((char *)h->sink + 4096) : nullptr;     // the word behind the 4 KB the masked lanes may write
        const int v2 = ((((uintptr_t)fx | (uintptr_t)fu) & 15) == 0) ? 1 : 0;
#define DDP_DFP(MODE_, GRID_) hipLaunchKernelGGL(df_pendcart_kernel<MODE_>, GRID_, block, 0, h->stream, p->N, p->B, p->g, p->l, p->h, p->d, p->goal[0], \
                                                 p->goal[1], p->goal[2], p->goal[3], p->Q, p->R, x, u, active, cx, cu, fx, fu, flag, v2)
        if (!flag || (de && de[0] == '1')) DDP_DFP(0, grid);
        else {
            DDP_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), h->stream));
            DDP_DFP(1, grid);
            DDP_DFP(2, dim3(1024));
        }
#undef DDP_DFP
    } else {
        DDP_CHECK(false, "df: unknown problem kind %d", p->kind);
    }
    DDP_HIP(hipGetLastError());
    return 0;
}

int ddp_df_f64(ddp_handle h, const ddp_problem *p, const double *x, const double *u, double *cx, double *cu,
               double *fx, double *fu)
{
    DDP_CHECK(h && p && x && u && cx && cu, "df: null argument");
    const size_t n = p->n, m = p->m, N = p->N, B = p->B;
    const bool pend = p->kind == DDP_PROBLEM_PENDCART;
    const size_t bytes = ((n + m) * N * B * 2 + n * n + m * m + (pend ? (n * n + n * m) * N * B : 0)) * 8 + 16 * 256;
    void *base;
    int rc = ddp_scratch(h, bytes, &base);
    if (rc) return rc;
    char *pp = (char *)base;
    auto take = [&](size_t cnt) { double *r = (double *)pp; pp += ((cnt * 8 + 255) & ~(size_t)255); return r; };
    double *dx = take(n * N * B), *du = take(m * N * B), *dcx = take(n * N * B), *dcu = take(m * N * B), *dQ = take(n * n),
           *dR = take(m * m), *dfx = pend ? take(n * n * N * B) : nullptr, *dfu = pend ? take(n * m * N * B) : nullptr;
    DDP_HIP(hipMemcpyAsync(dx, x, n * N * B * 8, hipMemcpyHostToDevice, h->stream));
    DDP_HIP(hipMemcpyAsync(du, u, m * N * B * 8, hipMemcpyHostToDevice, h->stream));
    DDP_HIP(hipMemcpyAsync(dQ, p->Q, n * n * 8, hipMemcpyHostToDevice, h->stream));
    DDP_HIP(hipMemcpyAsync(dR, p->R, m * m * 8, hipMemcpyHostToDevice, h->stream));
    ddp_problem pd = *p;
    pd.Q = dQ; pd.R = dR;
    rc = ddp_df_f64_dev(h, &pd, dx, du, nullptr, dcx, dcu, dfx, dfu);
    if (rc) return rc;
    DDP_HIP(hipMemcpyAsync(cx, dcx, n * N * B * 8, hipMemcpyDeviceToHost, h->stream));
    DDP_HIP(hipMemcpyAsync(cu, dcu, m * N * B * 8, hipMemcpyDeviceToHost, h->stream));
    if (pend && fx) DDP_HIP(hipMemcpyAsync(fx, dfx, n * n * N * B * 8, hipMemcpyDeviceToHost, h->stream));
    if (pend && fu) DDP_HIP(hipMemcpyAsync(fu, dfu, n * m * N * B * 8, hipMemcpyDeviceToHost, h->stream));
    DDP_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

}   // extern "C"
