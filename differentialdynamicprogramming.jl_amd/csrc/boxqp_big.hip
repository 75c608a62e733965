// boxqp_big.hip — boxQP(H, g, lower, upper, x0) for 8 < m <= DDP_QP_MAX_M (src/boxQP.jl:29-188; upstream's own large case is
// demoQP, m = 500, boxQP.jl:190-199): ONE 256-THREAD WORK-GROUP PER PROBLEM.  The control flow is the reference's, statement by
// statement, executed uniformly by all threads (every tested scalar is a work-group-wide value); what is parallel is the linear
// algebra inside an iteration:
//   grad = g + H·x, g + H·(x∘clamped)     one row per thread, j ascending (the reference's order); H[i + m·j]: coalesced
//   value = x'g + ((½x')·H)·x              a wave per column (lanes over rows, coalesced), wave sums, then a fixed-order sum
//   cholesky(H[free,free]).U               row by row (the order of LAPACK's unblocked dpotrf-U):
//                                          R[k,c] = (A[k,c] − Σ_{p<k} R[p,k]·R[p,c]) / R[k,k], one column c per thread, p ascending;
//                                          the factor is kept ROW-MAJOR in the caller's Hfree while it is worked on (row p of R is
//                                          then contiguous over the threads' columns) and transposed in place at the end
//   Hfree \ (Hfree' \ rhs)                 forward substitution in its column-oriented (axpy) form — same order of subtractions as
//                                          the row form —, backward substitution by one work-group-wide dot product per row
// No workspace besides the outputs and the LDS (10 vectors of m doubles).  Small m stays with the one-lane-per-problem kernel of
// capi.hip (m <= 8: the sizes the backward pass calls it with).
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

constexpr int QT = 256;

struct QPBig {
    int m;
    const double *H, *g, *lo, *up, *x0;
    QPOptsDev o;
    double *x, *Hfree;
    int32_t *result;
    uint8_t *free_out;
};

// sum of one value per thread, the same bits in every thread: wave sums by shuffles, then the four wave sums in order
__device__ __forceinline__ double block_sum(double v, double *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();                                            // red[] of an earlier call has been read
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(QT) void boxqp_big_kernel(QPBig a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int m = a.m, t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t mm = (size_t)m * m;
    const double *H = a.H + mm * t, *g = a.g + (size_t)m * t;
    double *R = a.Hfree + mm * t;                               // row-major while it is worked on: R[p][c] at p·m + c
    double *x = sm, *grad = sm + m, *gc = sm + 2 * m, *search = sm + 3 * m, *xc = sm + 4 * m, *lo = sm + 5 * m, *up = sm + 6 * m,
           *gs = sm + 7 * m, *col = sm + 8 * m, *rhs = sm + 9 * m, *red = sm + 10 * m;             // red: 8 doubles
    int *idx = (int *)(red + 8), *clamped = idx + m, *flags = clamped + m;                          // flags: 4 ints
    const QPOptsDev o = a.o;

    // (½x')·H·x with x'g: a wave per column, lanes over the rows
    auto qp_value = [&](const double *xv) -> double {
        double part = 0.0;
        for (int i = tid; i < m; i += QT) part += xv[i] * gs[i];
        const double xg = block_sum(part, red);
        double q = 0.0;
        for (int j = wave; j < m; j += QT / 64) {
            double tj = 0.0;
            for (int i = lane; i < m; i += 64) tj += (0.5 * xv[i]) * H[i + (size_t)m * j];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) tj += __shfl_xor(tj, off, 64);
            q += tj * xv[j];                                    // the same in every lane of the wave
        }
        return xg + block_sum(lane == 0 ? q : 0.0, red);
    };

    // Σ_j H[i,j]·v[j], j ascending, sixteen loads in flight
    auto row_dot = [&](int i, const double *v) -> double {
        const double *Hi = H + i;
        double s = 0.0;
        int j = 0;
        for (; j + 16 <= m; j += 16) {
            double r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) r[u] = Hi[(size_t)(j + u) * m];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += r[u] * v[j + u];
        }
        for (; j < m; ++j) s += Hi[(size_t)j * m] * v[j];
        return s;
    };

    for (int i = tid; i < m; i += QT) {
        lo[i] = a.lo[(size_t)m * t + i]; up[i] = a.up[(size_t)m * t + i]; gs[i] = g[i];
        x[i] = ddp_clamp(a.x0[(size_t)m * t + i], lo[i], up[i]);                                    // :58
        clamped[i] = 0; idx[i] = i;
        a.free_out[(size_t)m * t + i] = 1;                                                          // :47-48
    }
    for (size_t e = tid; e < mm; e += QT) R[e] = 0.0;                                               // :54
    __syncthreads();
    int result = 0, iter = 1, nfree = m;
    bool thrown = false;
    double oldvalue = 0.0, value = qp_value(x);                                                     // :63

    while (iter <= o.maxIter) {                                                                     // :71
        if (result != 0) break;
        if (iter > 1 && (oldvalue - value) < o.minRelImprove * fabs(oldvalue)) { result = 4; break; }   // :78-81
        oldvalue = value;
        // ---- grad = g + H·x (:85), clamped / free sets (:88-95)
        if (tid == 0) { flags[0] = 1; flags[1] = 0; }           // all clamped, set changed
        __syncthreads();
        for (int i = tid; i < m; i += QT) {
            const double gr = gs[i] + row_dot(i, x);
            grad[i] = gr;
            const int c = ((x[i] == lo[i]) && (gr > 0)) || ((x[i] == up[i]) && (gr < 0));
            if (c != clamped[i]) flags[1] = 1;
            if (!c) flags[0] = 0;
            clamped[i] = c;
            a.free_out[(size_t)m * t + i] = (uint8_t)!c;
        }
        __syncthreads();
        const bool all_clamped = flags[0] != 0, changed = flags[1] != 0;
        // the free coordinates in order: wave 0, 64 at a time (ballot + prefix count)
        if (wave == 0) {
            int base = 0;
            for (int i0 = 0; i0 < m; i0 += 64) {
                const int i = i0 + lane;
                const bool fr = i < m && !clamped[i];
                const unsigned long long bal = __ballot(fr);
                if (fr) idx[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
                base += __popcll(bal);
            }
            if (lane == 0) flags[2] = base;
        }
        __syncthreads();
        nfree = flags[2];
        if (all_clamped) { result = 6; break; }                                                     // :98-101
        // ---- factorize if the clamped set has changed (:104-117)
        if (iter == 1 || changed) {
            for (size_t e = tid; e < mm; e += QT) R[e] = 0.0;
            if (tid == 0) flags[3] = 0;
            __syncthreads();
            for (int k = 0; k < nfree; ++k) {
                for (int p = tid; p < k; p += QT) col[p] = R[(size_t)p * m + k];
                __syncthreads();
                const int ik = idx[k];
                constexpr int NQ = (DDP_QP_MAX_M + QT - 1) / QT;
                double sv[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c = k + tid + q * QT;
                    if (c < nfree) {
                        double s = H[ik + (size_t)m * idx[c]];  // upper triangle of H[free,free] (Hermitian view, DESIGN Q21)
                        // sixteen loads in flight, the subtractions in the order p = 0, 1, ... (one load at a time is an L2 round trip
                        // per multiply-add: 15 ms per factorisation at m = 500 instead of ~1)
                        const double *Rc = R + c;
                        int p = 0;
                        for (; p + 16 <= k; p += 16) {
                            double r[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) r[u] = Rc[(size_t)(p + u) * m];
#pragma unroll
                            for (int u = 0; u < 16; ++u) s -= col[p + u] * r[u];
                        }
                        for (; p < k; ++p) s -= col[p] * Rc[(size_t)p * m];
                        sv[q] = s;
                        if (c == k) {
                            if (!(s > 0.0)) flags[3] = 1;       // dpotrf: ajj <= 0 or NaN
                            red[4] = sqrt(s);
                        }
                    }
                }
                __syncthreads();
                if (flags[3]) break;
                const double piv = red[4];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c = k + tid + q * QT;
                    if (c < nfree) R[(size_t)k * m + c] = (c == k) ? piv : sv[q] / piv;
                }
                __syncthreads();
            }
            if (flags[3]) { result = 0; thrown = true; break; }                                     // PosDefException (caller: :48-52)
        }
        // ---- gradient norm over the free set (:120-124)
        {
            double part = 0.0;
            for (int q = tid; q < nfree; q += QT) part += grad[idx[q]] * grad[idx[q]];
            if (sqrt(block_sum(part, red)) < o.minGrad) { result = 5; break; }
        }
        // ---- search direction (:127-129): grad_clamped = g + H·(x∘clamped); search[free] = −Hfree\(Hfree'\grad_clamped[free]) − x[free]
        for (int i = tid; i < m; i += QT) xc[i] = clamped[i] ? x[i] : 0.0;
        __syncthreads();
        for (int i = tid; i < m; i += QT) {
            gc[i] = gs[i] + row_dot(i, xc);
        }
        __syncthreads();
        for (int q = tid; q < nfree; q += QT) rhs[q] = gc[idx[q]];
        __syncthreads();
        for (int k = 0; k < nfree; ++k) {                       // R'y = b: y_k, then b_i −= R[k,i]·y_k for i > k (k ascending per b_i)
            const double yk = rhs[k] / R[(size_t)k * m + k];
            __syncthreads();
            if (tid == 0) rhs[k] = yk;
            for (int i = k + 1 + tid; i < nfree; i += QT) rhs[i] -= R[(size_t)k * m + i] * yk;
            __syncthreads();
        }
        for (int i = nfree - 1; i >= 0; --i) {                  // R x = y: x_i = (y_i − Σ_{k>i} R[i,k]·x_k) / R[i,i]
            double part = 0.0;
            for (int k = i + 1 + tid; k < nfree; k += QT) part += R[(size_t)i * m + k] * rhs[k];
            const double s = block_sum(part, red);
            if (tid == 0) rhs[i] = (rhs[i] - s) / R[(size_t)i * m + i];
            __syncthreads();
        }
        for (int i = tid; i < m; i += QT) search[i] = 0.0;
        __syncthreads();
        for (int q = tid; q < nfree; q += QT) search[idx[q]] = -rhs[q] - x[idx[q]];
        __syncthreads();
        double sdotg;
        {
            double part = 0.0;
            for (int i = tid; i < m; i += QT) part += search[i] * grad[i];
            sdotg = block_sum(part, red);                                                           // :132
        }
        if (sdotg >= 0) break;                                                                      // :133-135
        // ---- Armijo line search (:138-151)
        double step = 1.0, vc;
        for (int i = tid; i < m; i += QT) xc[i] = ddp_clamp(x[i] + step * search[i], lo[i], up[i]);
        __syncthreads();
        vc = qp_value(xc);
        while ((vc - oldvalue) / (step * sdotg) < o.Armijo) {
            step = step * o.stepDec;
            __syncthreads();
            for (int i = tid; i < m; i += QT) xc[i] = ddp_clamp(x[i] + step * search[i], lo[i], up[i]);
            __syncthreads();
            vc = qp_value(xc);
            if (step < o.minStep) { result = 2; break; }
        }
        __syncthreads();
        for (int i = tid; i < m; i += QT) x[i] = xc[i];                                             // :161-163
        __syncthreads();
        value = vc;
        iter += 1;
    }
    if (!thrown && iter == o.maxIter) result = 1;                                                   // :167-169
    __syncthreads();
    for (int i = tid; i < m; i += QT) a.x[(size_t)m * t + i] = x[i];
    if (tid == 0) a.result[t] = result;
    // the factor to column-major (upper triangle): swap (p, c) with (c, p), the latter is zero
    __threadfence_block();
    __syncthreads();
    for (size_t e = tid; e < mm; e += QT) {
        const int p = (int)(e / m), c = (int)(e % m);
        if (p < c) { const double v = R[(size_t)p * m + c]; R[(size_t)p * m + c] = 0.0; R[(size_t)c * m + p] = v; }
    }
    (void)nfree;
}

}   // namespace

// m > 8: one work-group per problem
int ddp_launch_boxqp_big(ddp_handle h, int m, int count, const double *H, const double *g, const double *lower, const double *upper,
                         const double *x0, const QPOptsDev &o, double *x, int32_t *result, double *Hfree, uint8_t *free_out)
{
    QPBig a;
    a.m = m; a.H = H; a.g = g; a.lo = lower; a.up = upper; a.x0 = x0; a.o = o; a.x = x; a.Hfree = Hfree; a.result = result; a.free_out = free_out;
    const size_t shmem = (size_t)(10 * m + 8) * sizeof(double) + (size_t)(2 * m + 4) * sizeof(int);
    // per device and cheap: set on every launch (a process-wide "done" flag left the second GPU of a process at the 64 KB default)
    DDP_HIP(hipFuncSetAttribute((const void *)boxqp_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(boxqp_big_kernel, dim3(count), dim3(QT), shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
}
