// back_pass.hip — batched iLQG backward pass for gfx950 (MI355X).
//
// Replaces  back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u)  of the reference:
//   src/backward_pass.jl:217-252  (LTI dynamics, time-invariant cost)       FXTV=0 CTV=0
//   src/backward_pass.jl:162-177  (LTV dynamics, time-invariant cost)       FXTV=1 CTV=0
//   src/backward_pass.jl:179-215  (LTV dynamics, time-varying cost)         FXTV=1 CTV=1
//   shared tail @end_backward_pass src/backward_pass.jl:28-79 (Cholesky or boxQP, gains, value update)
//
// Mapping: ONE 64-lane wavefront per trajectory (work-group = 1 wave, so the four per-step
// hand-offs below are wave-level, not CU-level barriers).  The time loop is a strict dependency
// chain (Vxx_{i+1} -> Vxx_i); Vxx/Vx, the stacked Jacobian F=[fx fu] and the step's small
// intermediates live in LDS, the Qxx entries stay in registers between the expansion and the
// value update.  Per step:
//   P1  W = Vxx·F                (n·(n+m) dot products of length n, lanes over elements)
//       q = [cx;cu] + F'Vx        (Qx;Qu)
//   P2  upper triangle of Qxx = cxx + fx'W_x  (kept in registers), Qux/Quu (+ regularised
//       variants) = rows n..n+m of F'W  (+ λ·F_u'F for regType 2)
//   P3  every lane: Cholesky of QuuF (or boxQP, all lanes redundantly => wave-uniform control
//       flow); lane c<n solves column c of K, lane n solves k and accumulates dV
//   P4  Vxx_i = sym(Qxx + K'T + Qux'K), T = Quu·K + Qux;  Vx_i;  stores of K,k,Vx,Vxx,Quu
// Global traffic per step: reads cx_i,cu_i (+u_i with limits, + fx_i,fu_i / cxx_i,cxu_i,cuu_i when
// time-varying), writes K_i,k_i,Vx_i,Vxx_i,Quu_i — the algorithmic bytes of SURVEY.md §8(d).
// cx/cu/u are prefetched in chunks of TC time steps, time-varying operands one step ahead.
#include <stdlib.h>
#include "ddp_internal.h"
#include "boxqp_dev.h"

namespace {

struct BPArgs {
    int n, m, N, B;
    int fx_batched, cost_batched, regType;
    const double *cx, *cu, *cxx, *cxu, *cuu, *fx, *fu, *lambda, *lims, *u;
    const int32_t *active;
    double *K, *k, *Quu, *Vx, *Vxx, *dV;
    int32_t *diverge;
    // KL-augmented variant (back_pass_gps, backward_pass.jl:259-350): Q• <- Q•/η + c•kl, no λ, Quui = inv(Quu)
    const double *cxkl, *cukl, *cxxkl, *cxukl, *cuukl, *eta;
    int eta_tv;
    double *Quui;
};

constexpr int TC = 8;   // time steps per cx/cu/u prefetch chunk

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// LDS carve-up (doubles).  Identical on host (size computation) and device.
struct BPLds {
    int Fs, Vs, vs, Ws, Qs, cxxs, cxus, cuus, Quxs, Quxrs, Quus, QuuFs, Ks, ks, Ts, Quuks, cbuf, total;
    __host__ __device__ BPLds(int n, int m, bool lims)
    {
        const int p = n + m;
        int o = 0;
        Fs = o; o += n * p;
        Vs = o; o += n * n;
        vs = o; o += n;
        Ws = o; o += n * p;
        Qs = o; o += p;
        cxxs = o; o += n * n;
        cxus = o; o += n * m;
        cuus = o; o += m * m;
        Quxs = o; o += m * n;
        Quxrs = o; o += m * n;
        Quus = o; o += m * m;
        QuuFs = o; o += m * m;
        Ks = o; o += m * n;
        ks = o; o += m;
        Ts = o; o += m * n;
        Quuks = o; o += m;
        o = (o + 1) & ~1;
        cbuf = o; o += 2 * TC * (n + m + (lims ? m : 0));   // double-buffered cx|cu|u chunks
        total = (o + 1) & ~1;
    }
};

// inverse of the leading m x m block of H (column-major, leading dimension MM) by Gauss-Jordan elimination with partial
// pivoting — what `inv(Quu[:,:,i])` computes (backward_pass.jl:283,346) up to rounding order; out is m x m, ld m
template <int MM>
__device__ void inv_small(int m, const double (&H)[MM * MM], double *out)
{
    double A[MM * MM], X[MM * MM];
#pragma unroll
    for (int c = 0; c < MM; ++c)
#pragma unroll
        for (int r = 0; r < MM; ++r) { A[r + MM * c] = (r < m && c < m) ? H[r + MM * c] : (r == c ? 1.0 : 0.0); X[r + MM * c] = (r == c) ? 1.0 : 0.0; }
#pragma unroll
    for (int c = 0; c < MM; ++c) {
        if (c < m) {
            int pr = c; double best = fabs(A[c + MM * c]);
#pragma unroll
            for (int r = 0; r < MM; ++r) { const double v = fabs(A[r + MM * c]); if (r > c && r < m && v > best) { best = v; pr = r; } }
#pragma unroll
            for (int r = 0; r < MM; ++r) {
                if (r > c && r == pr) {                              // swap rows c and pr
#pragma unroll
                    for (int j = 0; j < MM; ++j) {
                        double t = A[c + MM * j]; A[c + MM * j] = A[r + MM * j]; A[r + MM * j] = t;
                        t = X[c + MM * j]; X[c + MM * j] = X[r + MM * j]; X[r + MM * j] = t;
                    }
                }
            }
            const double piv = 1.0 / A[c + MM * c];
#pragma unroll
            for (int j = 0; j < MM; ++j) { A[c + MM * j] *= piv; X[c + MM * j] *= piv; }
#pragma unroll
            for (int r = 0; r < MM; ++r) {
                if (r != c && r < m) {
                    const double f = A[r + MM * c];
#pragma unroll
                    for (int j = 0; j < MM; ++j) { A[r + MM * j] -= f * A[c + MM * j]; X[r + MM * j] -= f * X[c + MM * j]; }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MM; ++c)
#pragma unroll
        for (int r = 0; r < MM; ++r)
            if (r < m && c < m) out[r + m * c] = X[r + MM * c];
}

template <int NS, int MS, bool FXTV, bool CTV, bool LIMS, bool GPS = false>
__global__ __launch_bounds__(DDP_WAVE) void back_pass_kernel(BPArgs a)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (a.active && a.active[b] == 0) return;
    const int n = NS ? NS : a.n, m = MS ? MS : a.m, N = a.N, p = n + m;
    constexpr int NMAX = NS ? NS : DDP_MAX_N_GENERIC;
    constexpr int MM = MS ? MS : DDP_MAX_M;
    constexpr int PMAX = NMAX + MM;
    constexpr int R1 = cdiv(NMAX * PMAX, DDP_WAVE);                 // W elements per lane
    constexpr int RT = cdiv(NMAX * (NMAX + 1) / 2, DDP_WAVE);       // Qxx upper-triangle elements per lane
    constexpr int RU = cdiv(MM * PMAX, DDP_WAVE);                   // u-row elements per lane
    constexpr int RN = cdiv(NMAX * NMAX, DDP_WAVE);
    constexpr int RC = cdiv(TC * (NMAX + 2 * MM), DDP_WAVE);        // chunk elements per lane

    extern __shared__ double lds[];
    const BPLds L(n, m, LIMS);
    double *Fs = lds + L.Fs, *Vs = lds + L.Vs, *vs = lds + L.vs, *Ws = lds + L.Ws, *Qs = lds + L.Qs,
           *cxxs = lds + L.cxxs, *cxus = lds + L.cxus, *cuus = lds + L.cuus, *Quxs = lds + L.Quxs,
           *Quxrs = lds + L.Quxrs, *Quus = lds + L.Quus, *QuuFs = lds + L.QuuFs, *Ks = lds + L.Ks,
           *ks = lds + L.ks, *Ts = lds + L.Ts, *Quuks = lds + L.Quuks, *cbuf = lds + L.cbuf;

    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const double *cx = a.cx + (size_t)n * N * b, *cu = a.cu + (size_t)m * N * b;
    const double *ug = LIMS ? a.u + (size_t)m * N * b : nullptr;
    const double *fx = a.fx + (a.fx_batched ? nn * (FXTV ? N : 1) * b : 0);
    const double *fu = a.fu + (a.fx_batched ? nm * (FXTV ? N : 1) * b : 0);
    const double *cxx = a.cxx + (a.cost_batched ? nn * (CTV ? N : 1) * b : 0);
    const double *cxu = a.cxu + (a.cost_batched ? nm * (CTV ? N : 1) * b : 0);
    const double *cuu = a.cuu + (a.cost_batched ? mm * (CTV ? N : 1) * b : 0);
    double *Kg = a.K + nm * N * b, *kg = a.k + (size_t)m * N * b, *Quug = a.Quu + mm * N * b,
           *Vxg = a.Vx + (size_t)n * N * b, *Vxxg = a.Vxx + nn * N * b;
    const double lam = GPS ? 0.0 : a.lambda[b];
    const int regType = GPS ? 0 : a.regType;
    const double *cxkl = GPS ? a.cxkl + (size_t)n * N * b : nullptr, *cukl = GPS ? a.cukl + (size_t)m * N * b : nullptr,
                 *cxxkl = GPS ? a.cxxkl + nn * N * b : nullptr, *cxukl = GPS ? a.cxukl + nm * N * b : nullptr,
                 *cuukl = GPS ? a.cuukl + mm * N * b : nullptr, *etag = GPS ? a.eta + (a.eta_tv ? (size_t)N * b : b) : nullptr;
    double *Quuig = GPS ? a.Quui + mm * N * b : nullptr;
    bool nolims = true;
    double limlo[MM], limhi[MM];
    if (LIMS) {
        nolims = a.lims[0] > a.lims[m];                             // backward_pass.jl:31
#pragma unroll
        for (int q = 0; q < MM; ++q) {
            limlo[q] = (q < m) ? a.lims[q] : 0.0;
            limhi[q] = (q < m) ? a.lims[q + m] : 0.0;
        }
    }
    const QPOptsDev qpo = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};       // boxQP.jl:30-35

    // ---- per-lane element assignments (loop invariant)
    int w_v[R1], w_f[R1];                 // P1: LDS offsets of the V column and F column of W element
    for (int r = 0; r < R1; ++r) {
        const int e = lane + DDP_WAVE * r;
        w_v[r] = (e % n) * n;
        w_f[r] = (e / n) * n;
    }
    int t_i[RT], t_j[RT];                 // P2/P4: (i <= j) of the Qxx / Vxx upper-triangle element
    const int ntri = n * (n + 1) / 2;
    for (int r = 0; r < RT; ++r) {
        const int e = lane + DDP_WAVE * r;
        int j = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (j * (j + 1) / 2 > e) --j;
        while ((j + 1) * (j + 2) / 2 <= e) ++j;
        t_j[r] = j;
        t_i[r] = e - j * (j + 1) / 2;
    }
    const int chunk_len = n + m + (LIMS ? m : 0);     // doubles per time step in a chunk

    // chunk loader: chunk c holds time steps [c*TC, c*TC+TC) of cx | cu | u, time-major per stream
    auto chunk_elem = [&](int c, int e) -> double {   // e in [0, TC*chunk_len)
        const int t0 = c * TC;
        if (e < TC * n) {
            const int t = t0 + e / n;
            return t < N ? cx[(size_t)t0 * n + e] : 0.0;
        }
        e -= TC * n;
        if (e < TC * m) {
            const int t = t0 + e / m;
            return t < N ? cu[(size_t)t0 * m + e] : 0.0;
        }
        e -= TC * m;
        const int t = t0 + e / m;
        return (LIMS && t < N) ? ug[(size_t)t0 * m + e] : 0.0;
    };

    // ---- terminal step (backward_pass.jl:234-236 / :197-199) and loop-invariant operands
    for (int e = lane; e < n * n; e += DDP_WAVE) {
        const double v = cxx[(CTV ? nn * (N - 1) : 0) + e];
        Vs[e] = v;
        Vxxg[nn * (N - 1) + e] = v;
        if (!CTV) cxxs[e] = v;
    }
    for (int e = lane; e < n; e += DDP_WAVE) {
        const double v = cx[(size_t)n * (N - 1) + e];
        vs[e] = v;
        Vxg[(size_t)n * (N - 1) + e] = v;
    }
    for (int e = lane; e < m * m; e += DDP_WAVE) {
        double v = cuu[(CTV ? mm * (N - 1) : 0) + e];
        if (!CTV) cuus[e] = v;
        if (GPS) { v = v / etag[a.eta_tv ? N - 1 : 0] + cuukl[mm * (N - 1) + e]; QuuFs[e] = v; }   // :282
        Quug[mm * (N - 1) + e] = v;
    }
    if (GPS) {                                                      // Quui[:,:,N] = inv(Quu[:,:,N])  (:283)
        wave_sync();
        if (lane == 0) {
            double Hn[MM * MM];
#pragma unroll
            for (int c2 = 0; c2 < MM; ++c2)
#pragma unroll
                for (int r2 = 0; r2 < MM; ++r2) Hn[r2 + MM * c2] = (r2 < m && c2 < m) ? QuuFs[r2 + m * c2] : 0.0;
            inv_small<MM>(m, Hn, Quuig + mm * (N - 1));
        }
    }
    for (int e = lane; e < m * n; e += DDP_WAVE) {
        Kg[nm * (N - 1) + e] = 0.0;
        if (!CTV) cxus[e] = cxu[e];
    }
    for (int e = lane; e < m; e += DDP_WAVE) { kg[(size_t)m * (N - 1) + e] = 0.0; ks[e] = 0.0; }
    if (!FXTV) {
        for (int e = lane; e < n * n; e += DDP_WAVE) Fs[e] = fx[e];
        for (int e = lane; e < n * m; e += DDP_WAVE) Fs[n * n + e] = fu[e];
    }
    double dV0 = 0.0, dV1 = 0.0;
    if (N < 2) {
        if (lane == 0) { a.dV[2 * b] = 0.0; a.dV[2 * b + 1] = 0.0; a.diverge[b] = 0; }
        return;
    }
    // first step's operands straight to LDS, next chunk / next step into registers
    double pfc[RC];
    {
        const int c0 = (N - 2) / TC;
        for (int e = lane; e < TC * chunk_len; e += DDP_WAVE) cbuf[(c0 & 1) * TC * chunk_len + e] = chunk_elem(c0, e);
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const int e = lane + DDP_WAVE * r;
            pfc[r] = (c0 > 0 && e < TC * chunk_len) ? chunk_elem(c0 - 1, e) : 0.0;
        }
    }
    double pfF[FXTV ? R1 : 1], pfxx[CTV ? RN : 1], pfxu[CTV ? RU : 1], pfuu[CTV ? 1 : 1];
    if (FXTV) {
        const int i0 = N - 2;
        for (int e = lane; e < n * n; e += DDP_WAVE) Fs[e] = fx[nn * i0 + e];
        for (int e = lane; e < n * m; e += DDP_WAVE) Fs[n * n + e] = fu[nm * i0 + e];
    }
    if (CTV) {
        const int i0 = N - 2;
        for (int e = lane; e < n * n; e += DDP_WAVE) cxxs[e] = cxx[nn * i0 + e];
        for (int e = lane; e < n * m; e += DDP_WAVE) cxus[e] = cxu[nm * i0 + e];
        for (int e = lane; e < m * m; e += DDP_WAVE) cuus[e] = cuu[mm * i0 + e];
    }
    wave_sync();

    int diverge = 0;
    for (int i = N - 2; i >= 0; --i) {
        const int cc = i / TC;
        const double *cb = cbuf + (cc & 1) * TC * chunk_len;
        const double *cxi = cb + (i - cc * TC) * n;
        const double *cui = cb + TC * n + (i - cc * TC) * m;
        const double *ui = cb + TC * (n + m) + (i - cc * TC) * m;

        // ---- issue next step's time-varying operands (land while this step computes)
        if (FXTV && i > 0) {
#pragma unroll
            for (int r = 0; r < R1; ++r) {
                const int e = lane + DDP_WAVE * r;
                if (e < n * p) pfF[r] = (e < n * n) ? fx[nn * (i - 1) + e] : fu[nm * (i - 1) + (e - n * n)];
            }
        }
        if (CTV && i > 0) {
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const int e = lane + DDP_WAVE * r;
                if (e < n * n) pfxx[r] = cxx[nn * (i - 1) + e];
            }
#pragma unroll
            for (int r = 0; r < RU; ++r) {
                const int e = lane + DDP_WAVE * r;
                if (e < n * m) pfxu[r] = cxu[nm * (i - 1) + e];
            }
            if (lane < m * m) pfuu[0] = cuu[mm * (i - 1) + lane];
        }

        // ================= P1: W = Vxx·F,  Qs = [cx;cu] + F'Vx ==================================
#pragma unroll
        for (int r = 0; r < R1; ++r) {
            const int e = lane + DDP_WAVE * r;
            if (e < n * p) {
                const double *vc = Vs + w_v[r], *fc = Fs + w_f[r];
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < NMAX; ++l)
                    if (l < n) s += vc[l] * fc[l];
                Ws[e] = s;
            }
        }
        {
            const int e = DDP_WAVE - 1 - lane;        // high lanes: idle in the last W round
            for (int j = e; j < p; j += DDP_WAVE) {
                const double *fc = Fs + j * n;
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < NMAX; ++l)
                    if (l < n) s += fc[l] * vs[l];
                double qv = (j < n ? cxi[j] : cui[j - n]) + s;   // backward_pass.jl:240-241
                if (GPS) qv = qv / etag[a.eta_tv ? i : 0] + (j < n ? cxkl[(size_t)n * i + j] : cukl[(size_t)m * i + (j - n)]);   // :295,298
                Qs[j] = qv;
            }
        }
        wave_sync();

        // ================= P2: Qxx (registers), Qux, Quu and regularised variants ================
        double qxx[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int e = lane + DDP_WAVE * r;
            qxx[r] = 0.0;
            if (e < ntri) {
                const double *fc = Fs + t_i[r] * n, *wc = Ws + t_j[r] * n;
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < NMAX; ++l)
                    if (l < n) s += fc[l] * wc[l];
                qxx[r] = cxxs[t_i[r] + n * t_j[r]] + s;          // backward_pass.jl:244
                if (GPS)                                         // :299; only the symmetric part of cxxkl survives :341
                    qxx[r] = qxx[r] / etag[a.eta_tv ? i : 0] + 0.5 * (cxxkl[nn * i + t_i[r] + n * t_j[r]] + cxxkl[nn * i + t_j[r] + n * t_i[r]]);
            }
        }
#pragma unroll
        for (int r = 0; r < RU; ++r) {
            const int e = (DDP_WAVE - 1 - lane) + DDP_WAVE * r;
            if (e < m * p) {
                const int q = e % m, j = e / m;
                const double *fc = Fs + (n + q) * n, *wc = Ws + j * n, *fj = Fs + j * n;
                double s = 0.0, sr = 0.0;
#pragma unroll
                for (int l = 0; l < NMAX; ++l)
                    if (l < n) { s += fc[l] * wc[l]; sr += fc[l] * fj[l]; }
                if (GPS) {                                       // Q• <- Q•/η + c•kl, no λ  (:296-297)
                    const double et = etag[a.eta_tv ? i : 0];
                    if (j < n) {
                        const double v = (cxus[j + n * q] + s) / et + cxukl[nm * i + q + m * j];
                        Quxs[q + m * j] = v; Quxrs[q + m * j] = v;
                    } else {
                        const int bb = j - n;
                        Quus[q + m * bb] = (cuus[q + m * bb] + s) / et + cuukl[mm * i + q + m * bb];
                    }
                } else if (j < n) {                              // Qux, Qux_reg  (:242,246)
                    const double c = cxus[j + n * q];
                    Quxs[q + m * j] = c + s;
                    Quxrs[q + m * j] = c + (regType == 2 ? s + lam * sr : s);
                } else {                                         // Quu, QuuF     (:243,247)
                    const int bb = j - n;
                    const double c = cuus[q + m * bb];
                    Quus[q + m * bb] = c + s;
                    QuuFs[q + m * bb] = c + (regType == 2 ? s + lam * sr : s) + ((regType == 1 && q == bb) ? lam : 0.0);
                }
            }
        }
        wave_sync();
        if (GPS) {                                               // Quu = .5(Quu + Quu')  (:301); it is also the matrix factorised
            double sv = 0.0;
            if (lane < m * m) sv = 0.5 * (Quus[lane] + Quus[(lane / m) + m * (lane % m)]);
            wave_sync();
            if (lane < m * m) { Quus[lane] = sv; QuuFs[lane] = sv; }
            wave_sync();
        }

        // ================= P3: gains (backward_pass.jl:30-62) =====================================
        double H[MM * MM], R[MM * MM], kk[MM], ri[MM];
        unsigned clamped = 0u;
#pragma unroll
        for (int c2 = 0; c2 < MM; ++c2)
#pragma unroll
            for (int r2 = 0; r2 < MM; ++r2) H[r2 + MM * c2] = (r2 < m && c2 < m) ? QuuFs[r2 + m * c2] : 0.0;
        int fail;
        if (!LIMS || nolims) {
            fail = chol_masked_ri<MM>(m, H, 0u, R, ri);                // cholesky(Hermitian(QuuF)), :35
#pragma unroll
            for (int q = 0; q < MM; ++q) kk[q] = (q < m) ? Qs[n + q] : 0.0;
            chol_solve_ri<MM>(m, R, ri, kk);
#pragma unroll
            for (int q = 0; q < MM; ++q) kk[q] = -kk[q];         // k_i = -(R\Qu), :41
        } else {
            double g[MM], lo[MM], up[MM], x0[MM];
#pragma unroll
            for (int q = 0; q < MM; ++q) {
                const double uq = (q < m) ? ui[q] : 0.0;
                g[q] = (q < m) ? Qs[n + q] : 0.0;
                lo[q] = limlo[q] - uq;                           // :45-46
                up[q] = limhi[q] - uq;
                x0[q] = (q < m) ? ks[q] : 0.0;                   // k[:,min(i+1,N-1)], :49 (Q9)
            }
            int iters;
            const int result = boxqp_dev_ri<MM>(m, H, g, lo, up, x0, qpo, kk, R, ri, clamped, iters);
            fail = (result < 1);                                 // :53
        }
        if (fail) {                                              // wave-uniform: diverge = i (:37-38,54-55)
            diverge = i + 1;
            // Quu[:,:,i] was already assigned by the reference before the failure
            for (int e = lane; e < m * m; e += DDP_WAVE) Quug[mm * i + e] = Quus[e];
            for (size_t e = lane; e < nm * (i + 1); e += DDP_WAVE) Kg[e] = 0.0;
            for (size_t e = lane; e < (size_t)m * (i + 1); e += DDP_WAVE) kg[e] = 0.0;
            for (size_t e = lane; e < (size_t)n * (i + 1); e += DDP_WAVE) Vxg[e] = 0.0;
            for (size_t e = lane; e < nn * (i + 1); e += DDP_WAVE) Vxxg[e] = 0.0;
            for (size_t e = lane; e < mm * i; e += DDP_WAVE) Quug[e] = 0.0;
            break;
        }
        if (lane < n) {                                          // K_i column `lane`
            double col[MM];
#pragma unroll
            for (int q = 0; q < MM; ++q) col[q] = (q < m && !((clamped >> q) & 1u)) ? Quxrs[q + m * lane] : 0.0;
            chol_solve<MM>(m, R, col);                           // :42 / :59  (IEEE divisions: with the reciprocal-pivot solve the
                                                                 //  GPS + limits instantiation, 550 spilled VGPRs, faults on gfx950)
#pragma unroll
            for (int q = 0; q < MM; ++q) col[q] = ((clamped >> q) & 1u) ? 0.0 : -col[q];
#pragma unroll
            for (int q = 0; q < MM; ++q) {
                if (q < m) {
                    double t = Quxs[q + m * lane];               // T = Quu·K + Qux
#pragma unroll
                    for (int q2 = 0; q2 < MM; ++q2)
                        if (q2 < m) t += Quus[q + m * q2] * col[q2];
                    Ks[q + m * lane] = col[q];
                    Ts[q + m * lane] = t;
                }
            }
        } else if (lane == n) {                                  // k_i, Quu·k, dV (:64-68)
            double kQu = 0.0, kQuuk = 0.0;
#pragma unroll
            for (int q = 0; q < MM; ++q) {
                if (q < m) {
                    double t = 0.0;
#pragma unroll
                    for (int q2 = 0; q2 < MM; ++q2)
                        if (q2 < m) t += Quus[q + m * q2] * kk[q2];
                    Quuks[q] = t;
                    ks[q] = kk[q];
                    kQu += kk[q] * Qs[n + q];
                    kQuuk += kk[q] * t;
                }
            }
            dV0 += kQu;
            dV1 += 0.5 * kQuuk;
        }
        if (GPS && lane == DDP_WAVE - 1) {                       // Quui[:,:,i] = inv(Quu[:,:,i])  (:346)
            double Hq[MM * MM];
#pragma unroll
            for (int c2 = 0; c2 < MM; ++c2)
#pragma unroll
                for (int r2 = 0; r2 < MM; ++r2) Hq[r2 + MM * c2] = (r2 < m && c2 < m) ? Quus[r2 + m * c2] : 0.0;
            inv_small<MM>(m, Hq, Quuig + mm * i);
        }
        wave_sync();

        // ================= P4: value update (:69-76), stores, operand hand-over ====================
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int e = lane + DDP_WAVE * r;
            if (e < ntri) {
                const int ii = t_i[r], jj = t_j[r];
                double mij = qxx[r], mji = qxx[r];
#pragma unroll
                for (int q = 0; q < MM; ++q) {
                    if (q < m) {
                        const double Ki = Ks[q + m * ii], Kj = Ks[q + m * jj];
                        mij += Ki * Ts[q + m * jj] + Quxs[q + m * ii] * Kj;
                        mji += Kj * Ts[q + m * ii] + Quxs[q + m * jj] * Ki;
                    }
                }
                const double v = (mij + mji) / 2;                // :71-72
                Vs[ii + n * jj] = v;
                Vs[jj + n * ii] = v;
                Vxxg[nn * i + ii + n * jj] = v;
                Vxxg[nn * i + jj + n * ii] = v;
            }
        }
        for (int j = DDP_WAVE - 1 - lane; j < n; j += DDP_WAVE) {   // Vx_i (:69)
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int q = 0; q < MM; ++q) {
                if (q < m) {
                    s1 += Ks[q + m * j] * Quuks[q];
                    s2 += Ks[q + m * j] * Qs[n + q];
                    s3 += Quxs[q + m * j] * ks[q];
                }
            }
            const double v = ((Qs[j] + s1) + s2) + s3;
            vs[j] = v;
            Vxg[(size_t)n * i + j] = v;
        }
        for (int e = lane; e < m * n; e += DDP_WAVE) Kg[nm * i + e] = Ks[e];      // :75-76
        if (lane < m) kg[(size_t)m * i + lane] = ks[lane];
        if (lane < m * m) Quug[mm * i + lane] = Quus[lane];
        // hand the prefetched operands of step i-1 to LDS (their last readers were P1/P2)
        if (FXTV && i > 0) {
#pragma unroll
            for (int r = 0; r < R1; ++r) {
                const int e = lane + DDP_WAVE * r;
                if (e < n * p) Fs[e] = pfF[r];
            }
        }
        if (CTV && i > 0) {
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const int e = lane + DDP_WAVE * r;
                if (e < n * n) cxxs[e] = pfxx[r];
            }
#pragma unroll
            for (int r = 0; r < RU; ++r) {
                const int e = lane + DDP_WAVE * r;
                if (e < n * m) cxus[e] = pfxu[r];
            }
            if (lane < m * m) cuus[lane] = pfuu[0];
        }
        if (i > 0 && i == cc * TC) {                             // leaving chunk cc: publish cc-1, fetch cc-2
            double *nb = cbuf + ((cc - 1) & 1) * TC * chunk_len;
#pragma unroll
            for (int r = 0; r < RC; ++r) {
                const int e = lane + DDP_WAVE * r;
                if (e < TC * chunk_len) nb[e] = pfc[r];
            }
            if (cc >= 2) {
#pragma unroll
                for (int r = 0; r < RC; ++r) {
                    const int e = lane + DDP_WAVE * r;
                    if (e < TC * chunk_len) pfc[r] = chunk_elem(cc - 2, e);
                }
            }
        }
        wave_sync();
    }
    if (lane == n) { a.dV[2 * b] = dV0; a.dV[2 * b + 1] = dV1; }
    if (lane == 0) a.diverge[b] = diverge;
}

template <int NS, int MS>
int launch_nm(ddp_handle h, const ddp_bp_desc *d, const BPArgs &a)
{
    const BPLds L(d->n, d->m, d->has_lims != 0);
    const size_t shmem = (size_t)L.total * sizeof(double);
    const dim3 grid(d->B), block(DDP_WAVE);
    const int key = (d->fx_tv ? 4 : 0) | (d->cost_tv ? 2 : 0) | (d->has_lims ? 1 : 0);
#define DDP_BP_CASE(K_, FX_, C_, L_)                                                                 \
    case K_:                                                                                         \
        hipLaunchKernelGGL((back_pass_kernel<NS, MS, FX_, C_, L_>), grid, block, shmem, h->stream, a); \
        break;
    switch (key) {
        DDP_BP_CASE(0, false, false, false)
#ifndef DDP_FAST_BUILD
        DDP_BP_CASE(1, false, false, true)
        DDP_BP_CASE(2, false, true, false)
        DDP_BP_CASE(3, false, true, true)
        DDP_BP_CASE(4, true, false, false)
        DDP_BP_CASE(5, true, false, true)
        DDP_BP_CASE(6, true, true, false)
        DDP_BP_CASE(7, true, true, true)
#endif
    }
#undef DDP_BP_CASE
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // namespace

// back_pass_gps (backward_pass.jl:259-350): run-time sizes n <= 32, m <= 8, all cost/dynamics arrays 3-D as the
// reference's method signature requires.  Quui must be zero-filled by the caller (steps before a failure stay zero).
int ddp_launch_back_pass_gps(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const ddp_kl_cost_terms *kl, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Quui, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge)
{
    DDP_DEVICE(h);
#ifndef DDP_FAST_BUILD
    DDP_CHECK(d->n >= 1 && d->m >= 1 && d->N >= 1 && d->B >= 1, "back_pass_gps: bad sizes n=%d m=%d N=%d B=%d", d->n, d->m, d->N, d->B);
    DDP_CHECK(d->fx_tv && d->cost_tv, "back_pass_gps: needs time-varying (3-D) fx/fu and cxx/cxu/cuu like backward_pass.jl:259");
    DDP_CHECK(!d->has_lims || (lims && u), "back_pass_gps: has_lims needs lims and u");
    DDP_CHECK(d->n <= DDP_MAX_N_GENERIC && d->m <= DDP_MAX_M, "back_pass_gps: n=%d m=%d has no kernel (n <= %d, m <= %d)", d->n, d->m,
              DDP_MAX_N_GENERIC, DDP_MAX_M);
    DDP_CHECK(kl && kl->cx && kl->cu && kl->cxx && kl->cxu && kl->cuu && kl->eta, "back_pass_gps: incomplete kl_cost_terms");
    BPArgs a;
    a.n = d->n; a.m = d->m; a.N = d->N; a.B = d->B;
    a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched; a.regType = 1;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu;
    a.lambda = nullptr; a.lims = lims; a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    a.cxkl = kl->cx; a.cukl = kl->cu; a.cxxkl = kl->cxx; a.cxukl = kl->cxu; a.cuukl = kl->cuu; a.eta = kl->eta; a.eta_tv = kl->eta_tv;
    a.Quui = Quui;
    const BPLds L(d->n, d->m, d->has_lims != 0);
    const size_t shmem = (size_t)L.total * sizeof(double);
    const dim3 grid(d->B), block(DDP_WAVE);
    if (d->has_lims) hipLaunchKernelGGL((back_pass_kernel<0, 0, true, true, true, true>), grid, block, shmem, h->stream, a);
    else hipLaunchKernelGGL((back_pass_kernel<0, 0, true, true, false, true>), grid, block, shmem, h->stream, a);
    DDP_HIP(hipGetLastError());
    return 0;
#else
    DDP_CHECK(false, "back_pass_gps: not in DDP_FAST_BUILD");
#endif
}

// ---- odd sizes above the run-time-sized kernel's range (33 <= n <= 63 odd, or an odd m with n > 32): the 256-thread kernel
// works on 2x2 register blocks, so the problem is embedded in the next even sizes — an extra state that stays zero and costs
// nothing, an extra control with cuu = 1, zero gradient and no coupling (its gain is exactly zero) — and the results cut back.
// Sums gain exact zeros only, so the outputs are those of the unpadded problem.  Compatibility path: operands and results take a
// trip through a pad buffer owned by the handle.
namespace {
// dst[rp x cp x cnt] <- src[r x c x cnt] in the top-left corner, zeros elsewhere, ones on the diagonal from index `one_from` on
__global__ __launch_bounds__(256) void pad2d_kernel(const double *src, double *dst, int r, int c, int rp, int cp, long cnt, int one_from)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rp * cp * cnt) return;
    const int i = (int)(e % rp), j = (int)((e / rp) % cp);
    const long t = e / ((long)rp * cp);
    dst[e] = (i < r && j < c) ? src[i + (long)r * (j + (long)c * t)] : ((i == j && i >= one_from) ? 1.0 : 0.0);
}
// dst[r x c x cnt] <- top-left corner of src[rp x cp x cnt]; `per` consecutive slices belong to one trajectory (active mask)
__global__ __launch_bounds__(256) void unpad2d_kernel(const double *src, double *dst, int r, int c, int rp, int cp, long cnt, long per,
                                                      const int32_t *active)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)r * c * cnt) return;
    const int i = (int)(e % r), j = (int)((e / r) % c);
    const long t = e / ((long)r * c);
    if (active && active[t / per] == 0) return;
    dst[e] = src[i + (long)rp * (j + (long)cp * t)];
}
}   // namespace

// the embedded problem's work space in bytes (operands and results at the padded sizes)
static size_t padded_bytes(const ddp_bp_desc *d, int np_, int mp)
{
    const size_t N = d->N, B = d->B, cf = (d->fx_tv ? N : 1) * (d->fx_batched ? B : 1), cc = (d->cost_tv ? N : 1) * (d->cost_batched ? B : 1), NB = N * B;
    return 8 * ((size_t)np_ * NB * 2 + (size_t)mp * NB * 3 + ((size_t)np_ * np_ + (size_t)np_ * mp + (size_t)mp * mp) * cc + ((size_t)np_ * np_ + (size_t)np_ * mp) * cf +
                ((size_t)mp * np_ + (size_t)mp * mp + (size_t)np_ * np_) * NB) + 16 * 256;
}

// (np_, mp): the sizes the problem is embedded in — the next even ones for back_pass_big_kernel, or (64, 8) for the matrix-core kernel
static int launch_back_pass_padded(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu, const double *cxx, const double *cxu,
                                   const double *cuu, const double *fx, const double *fu, const double *lambda, const double *lims,
                                   const double *u, const int32_t *active, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                                   double *dV, int32_t *diverge, int np_, int mp, bool to_mfma)
{
    const int n = d->n, m = d->m;
    const long N = d->N, B = d->B;
    const long cf = (d->fx_tv ? N : 1) * (d->fx_batched ? B : 1), cc = (d->cost_tv ? N : 1) * (d->cost_batched ? B : 1), NB = N * B;
    auto al = [](size_t b_) { return (b_ + 255) & ~(size_t)255; };
    const size_t s_cx = al((size_t)np_ * NB * 8), s_cu = al((size_t)mp * NB * 8), s_cxx = al((size_t)np_ * np_ * cc * 8),
                 s_cxu = al((size_t)np_ * mp * cc * 8), s_cuu = al((size_t)mp * mp * cc * 8), s_fx = al((size_t)np_ * np_ * cf * 8),
                 s_fu = al((size_t)np_ * mp * cf * 8), s_l = al((size_t)mp * 2 * 8), s_K = al((size_t)mp * np_ * NB * 8),
                 s_Quu = al((size_t)mp * mp * NB * 8), s_Vxx = al((size_t)np_ * np_ * NB * 8);
    const size_t bytes = s_cx + 2 * s_cu + s_cxx + s_cxu + s_cuu + s_fx + s_fu + s_l + s_K + s_cu + s_Quu + s_cx + s_Vxx;
    if (bytes > h->pad_bytes) {
        DDP_HIP(hipStreamSynchronize(h->stream));
        if (h->pad) DDP_HIP(hipFree(h->pad));
        h->pad = nullptr; h->pad_bytes = 0;
        DDP_HIP(hipMalloc(&h->pad, bytes));
        h->pad_bytes = bytes;
    }
    char *q = (char *)h->pad;
    auto tk = [&](size_t b_) { double *r_ = (double *)q; q += b_; return r_; };
    double *pcx = tk(s_cx), *pcu = tk(s_cu), *pu = tk(s_cu), *pcxx = tk(s_cxx), *pcxu = tk(s_cxu), *pcuu = tk(s_cuu), *pfx = tk(s_fx),
           *pfu = tk(s_fu), *pl = tk(s_l), *pK = tk(s_K), *pk = tk(s_cu), *pQuu = tk(s_Quu), *pVx = tk(s_cx), *pVxx = tk(s_Vxx);
    hipStream_t st = h->stream;
    auto pad = [&](const double *src, double *dst, int r, int c, int rp, int cp, long cnt, int one_from) {
        const long tot = (long)rp * cp * cnt;
        hipLaunchKernelGGL(pad2d_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, dst, r, c, rp, cp, cnt, one_from);
    };
    pad(cx, pcx, n, 1, np_, 1, NB, 1 << 30); pad(cu, pcu, m, 1, mp, 1, NB, 1 << 30);
    if (u) pad(u, pu, m, 1, mp, 1, NB, 1 << 30);
    pad(cxx, pcxx, n, n, np_, np_, cc, 1 << 30); pad(cxu, pcxu, n, m, np_, mp, cc, 1 << 30); pad(cuu, pcuu, m, m, mp, mp, cc, m);
    pad(fx, pfx, n, n, np_, np_, cf, 1 << 30); pad(fu, pfu, n, m, np_, mp, cf, 1 << 30);
    if (lims) {                                                 // the extra control is free inside [-1, 1] (it stays at 0)
        double hl[2 * DDP_MAX_M];
        DDP_HIP(hipMemcpyAsync(hl, lims, (size_t)m * 2 * 8, hipMemcpyDeviceToHost, st));
        DDP_HIP(hipStreamSynchronize(st));
        double hp[2 * DDP_MAX_M + 4];
        for (int q2 = 0; q2 < mp; ++q2) { hp[q2] = q2 < m ? hl[q2] : -1.0; hp[q2 + mp] = q2 < m ? hl[q2 + m] : 1.0; }
        DDP_HIP(hipMemcpyAsync(pl, hp, (size_t)mp * 2 * 8, hipMemcpyHostToDevice, st));
        DDP_HIP(hipStreamSynchronize(st));                      // hp lives on this stack frame
    }
    ddp_bp_desc dp = *d;
    dp.n = np_; dp.m = mp;
    const int rc = to_mfma ? ddp_launch_back_pass_mfma(h, &dp, pcx, pcu, pcxx, pcxu, pcuu, pfx, pfu, lambda, lims ? pl : nullptr, u ? pu : nullptr, active,
                                                       pK, pk, pQuu, pVx, pVxx, dV, diverge)
                           : ddp_launch_back_pass_big(h, &dp, pcx, pcu, pcxx, pcxu, pcuu, pfx, pfu, lambda, lims ? pl : nullptr, u ? pu : nullptr, active,
                                                      pK, pk, pQuu, pVx, pVxx, dV, diverge);
    if (rc) return rc < 0 ? rc : -1;
    auto unpad = [&](const double *src, double *dst, int r, int c, int rp, int cp) {
        const long tot = (long)r * c * NB;
        hipLaunchKernelGGL(unpad2d_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, dst, r, c, rp, cp, NB, N, active);
    };
    unpad(pK, K, m, n, mp, np_); unpad(pk, k, m, 1, mp, 1); unpad(pQuu, Quu, m, m, mp, mp); unpad(pVx, Vx, n, 1, np_, 1);
    unpad(pVxx, Vxx, n, n, np_, np_);
    DDP_HIP(hipGetLastError());
    return 0;
}

// back_pass_dppw.hip: the row kernel with a write-back wave per chain wave (n = 10, m = 2, LTI, no limits, large batches); 1 = not applicable
int ddp_launch_back_pass_dppw(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                              const double *cxx, const double *cxu, const double *cuu, const double *fx,
                              const double *fu, const double *lambda, const int32_t *active, double *K,
                              double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge);

constexpr int MXG_LIMS_MAX_B = 2048;             // (measured cross-over with the row kernels: profiles/r05_lims_sweep.sh)

static int launch_back_pass_inner(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                         const double *cxx, const double *cxu, const double *cuu, const double *fx,
                         const double *fu, const double *lambda, const double *lims, const double *u,
                         const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                         double *Vxx, double *dV, int32_t *diverge)
{
    // Kernel choice.  Several implementations of the same arithmetic exist:
    //   x (mx)  one 16x16 fp64 MFMA tile per trajectory, one wave each (back_pass_mx.hip; n=10, m=2, no limits): shortest
    //           dependent chain per time step, best while the batch gives a SIMD only one or two waves (B=1024: 0.55 ms
    //           against 0.90 ms for the 64-lane vector kernel it replaced); measured cross-over with `dpp` between B=4096 and B=6144;
    //   dpp     16 lanes per trajectory (back_pass_dpp.hip): fewest instructions per trajectory-step, best once the
    //           batch gives every SIMD a few wavefronts; also the kernel for control limits;
    //   general 64 lanes per trajectory, any n <= 32 / m <= 8 / limits (this file).
    //   row     the same row kernel compiled for PADDED sizes (back_pass_row.hip): any n <= 14, m <= 4, n + m <= 15 that has no exact instantiation;
    //   tile    the mx kernel with run-time sizes n <= 10, m <= 2 (back_pass_mx.hip, RT), small and medium batches without limits;
    //   mid     one wave per trajectory with LDS operands (back_pass_mid.hip): n <= 32, m <= 8;
    //   wtile   the tile kernel for n <= 12, m <= 4 (back_pass_mxg.hip), same batches;
    // DDP_BACKPASS=x|q|general|dpp|row|tile|wtile|mid|big forces one (A/B timing, tests of every code path).
    const char *force_env = ddp_env(h, ENV_BACKPASS);          // read per call so tests can switch paths
    const char force = force_env ? force_env[0] : 0;
    if (force == 'x' || (force == 0 && d->B < 5120)) {
        // two waves per trajectory (chain + write-back, back_pass_mx2.hip) while a CU's four SIMDs hold one trajectory each;
        // DDP_MX2=0 / 1 forces the one-wave / two-wave kernel
        const char *mx2_env = ddp_env(h, ENV_MX2);
        if (mx2_env ? mx2_env[0] == '1' : d->B <= 1024) {
            const int r2 = ddp_launch_back_pass_mx2(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, active, K, k, Quu, Vx, Vxx, dV, diverge);
            if (r2 <= 0) { h->last_kernel[0] = "back_pass_mx2_kernel"; return r2; }
        }
        const int rc = ddp_launch_back_pass_mx(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_mx_kernel"; return rc; }
    }
    if (force == 'q' || force == 0) {                             // n = 4, m = 1: one trajectory per 4x4x4 MFMA block
        const int rc = ddp_launch_back_pass_q4(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_q4"; return rc; }
    }
    if (force == 0 || force == 'd') {                             // machine-filling batches of the LTI shape: row kernel + write-back waves
        const int rw = ddp_launch_back_pass_dppw(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rw <= 0) { h->last_kernel[0] = "back_pass_dppw_kernel"; return rw; }
    }
    // control limits at small and medium batches (any n <= 12, m <= 4): one WAVE per trajectory with the box-QP as a wave-uniform solve
    // (back_pass_mxg.hip) instead of 16 lanes per trajectory with a divergent one — at B = 1 024 the row kernels leave three quarters of
    // the SIMDs without a wave (n=10, m=2, N=1000 with limits: 2.8 ms there; profiles/r05_lims_sweep.sh)
    // (measured cross-over, profiles/r05_lims_threshold.txt: a wave-uniform QP is bound by its own latency, so the time doubles with a
    // second wave on a SIMD; the row kernels hold their time up to B = 4 096 — n=6, m=2: 1.53 vs 2.31 ms at B = 1 024, 3.0 vs 2.45 at 1 536)
    const int mxg_lims_max = d->m == 1 ? 512 : ((d->n > 8 || d->m >= 3 || d->n <= 4) ? MXG_LIMS_MAX_B : 1024);   // (m = 1: the row kernels' straight-line QP — 0.73 vs 0.81 ms at n=3, B = 1 024)
    if (d->has_lims && (force == 'w' || (force == 0 && d->B <= mxg_lims_max))) {
        const int rc = ddp_launch_back_pass_mxg(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_mxg_kernel"; return rc; }
    }
    if (force != 'g' && force != 'b' && force != 'r' && force != 't' && force != 'w' && force != 'm') {      // (a forced family either runs or falls through to the general kernel)
        const int rc = ddp_launch_back_pass_dpp(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_dpp_kernel"; return rc; }
    }
    // n <= 10, m <= 2 without limits: the fp64 tile kernel with run-time sizes while a SIMD holds one to three waves — its step is half the
    // row kernel's (0.49 vs 0.99 ms at n=6, m=2, N=1000, B=1024) but costs seven 16x16x4 products whatever n is, so the row kernel
    // wins once the matrix pipe is the bound (profiles/r05_tile_vs_row.txt: B=3072 1.33 vs 1.47 ms at n=6, 1.21 vs 0.79 at n=3)
    if (force == 't' || (force == 0 && (d->B <= 1024 || (d->B <= 3072 && d->n >= 5)))) {
        const int rc = ddp_launch_back_pass_mxr(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_mx_kernel<RT>"; return rc; }
    }
    // the same for n <= 12, m <= 4 (back_pass_mxg.hip; profiles/r05_wtile_vs_row.txt: ahead of the row kernel up to B = 4096 — 0.83 vs 1.16 ms at
    // n=12, m=3, N=500, B=2048; 1.15 vs 1.55 at n=8, m=4, B=4096 — level with it there for n <= 4)
    if (force == 'w' || (force == 0 && !d->has_lims && (d->B <= 3072 || (d->B <= 4096 && d->n >= 5)))) {      // (with limits: the block above, up to MXG_LIMS_MAX_B)
        const int rc = ddp_launch_back_pass_mxg(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_mxg_kernel"; return rc; }
    }
    if (force == 0 || force == 'r') {                             // every other shape a 16-lane row holds: the row kernel compiled for padded sizes
        const int rc = ddp_launch_back_pass_row(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_row_kernel"; return rc; }
    }
    if (force == 0 || force == 'm') {                             // 14 < n <= 32 (or m > 4): one wave per trajectory on the matrix cores, LDS operands
        const int rc = ddp_launch_back_pass_mid(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_mid_kernel"; return rc; }
    }
    h->last_kernel[0] = "back_pass_kernel";
    BPArgs a = {};
    a.n = d->n; a.m = d->m; a.N = d->N; a.B = d->B;
    a.fx_batched = d->fx_batched; a.cost_batched = d->cost_batched; a.regType = d->regType;
    a.cx = cx; a.cu = cu; a.cxx = cxx; a.cxu = cxu; a.cuu = cuu; a.fx = fx; a.fu = fu;
    a.lambda = lambda; a.lims = lims; a.u = u; a.active = active;
    a.K = K; a.k = k; a.Quu = Quu; a.Vx = Vx; a.Vxx = Vxx; a.dV = dV; a.diverge = diverge;
    if (d->n == 10 && d->m == 2) return launch_nm<10, 2>(h, d, a);
#ifndef DDP_FAST_BUILD
    if (d->n == 4 && d->m == 1) return launch_nm<4, 1>(h, d, a);
    if (d->n == 6 && d->m == 3) return launch_nm<6, 3>(h, d, a);
    // 32 < n <= 64, m <= 8: the fp64 matrix-core kernel with run-time sizes (back_pass_mf2_kernel.h; round 5 embedded 32 < n < 64 in the
    // (64, 8) problem through padded COPIES of every operand and result — up to 48 GB of scratch on the handle, half of the time in the
    // copy kernels).  DDP_BACKPASS=old: the round-5 kernel of the exact (64, 8) shape (A/B timing).
    if (force == 'o') {
        const int rc = ddp_launch_back_pass_mfma(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_mfma_kernel"; return rc; }
    }
    if (force != 'b' && force != 'g') {
        // (rc 2: the exact (64, 8) shape with a time-varying cost, or with REAL control limits — lims[1,1] <= lims[1,2], which the launcher has just looked at — stays
        // on the round-5 kernel by default: its gain wave runs the 8 x 8 box-QP in 16 400 ticks per step against 21 200 in the
        // run-time-sized kernel, 10.8 vs 13.3 ms at C4 with limits; DDP_BACKPASS=new forces the new one)
        const int rc = ddp_launch_back_pass_mf2(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge, force != 'n');
        if (rc <= 0) { h->last_kernel[0] = "back_pass_mf2_kernel"; return rc; }
        if (rc == 2) {
            const int ro = ddp_launch_back_pass_mfma(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
            if (ro <= 0) { h->last_kernel[0] = "back_pass_mfma_kernel"; return ro; }
        }
    }
    if (d->n > DDP_MAX_N_GENERIC || force == 'b') {               // large states: 256-thread work-group per trajectory
        const int rc = ddp_launch_back_pass_big(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
        if (rc <= 0) { h->last_kernel[0] = "back_pass_big_kernel"; return rc; }
    }
    if (d->n > DDP_MAX_N_GENERIC && d->n <= 64 && ((d->n | d->m) & 1))      // odd n or m: embed in the next even sizes
        return launch_back_pass_padded(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge, d->n + (d->n & 1), d->m + (d->m & 1), false);
    DDP_CHECK(d->n <= DDP_MAX_N_GENERIC, "back_pass: n=%d m=%d has no kernel (n <= %d with m <= %d, or n <= 64)", d->n, d->m, DDP_MAX_N_GENERIC, DDP_MAX_M);
    return launch_nm<0, 0>(h, d, a);
#else
    DDP_CHECK(false, "back_pass: DDP_FAST_BUILD only has the (10,2) LTI kernel");
#endif
}

int ddp_launch_back_pass(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                         const double *cxx, const double *cxu, const double *cuu, const double *fx,
                         const double *fu, const double *lambda, const double *lims, const double *u,
                         const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                         double *Vxx, double *dV, int32_t *diverge)
{
    DDP_DEVICE(h);
    DDP_CHECK(d->n >= 1 && d->m >= 1 && d->N >= 1 && d->B >= 1, "back_pass: bad sizes n=%d m=%d N=%d B=%d", d->n, d->m, d->N, d->B);
    DDP_CHECK(d->regType == 1 || d->regType == 2, "back_pass: regType must be 1 or 2 (got %d)", d->regType);
    DDP_CHECK(!d->has_lims || (lims && u), "back_pass: has_lims needs lims and u");
    DDP_CHECK(d->m <= DDP_MAX_M, "back_pass: m=%d exceeds DDP_MAX_M=%d", d->m, DDP_MAX_M);
    const char *force_env = ddp_env(h, ENV_BACKPASS);
    const char force = force_env ? force_env[0] : 0;
    // Operands shared by the batch (the reference's LTI method with ONE fx, fu, cxx, cxu, cuu): the matrix recursion once per distinct
    // λ (back_pass_sh.hip).  What it leaves out (λ values that occur once, more distinct values than it has groups) comes back as the
    // activity mask of the per-trajectory kernels, which are launched behind it and exit at once when there is nothing for them.
    // Measured (profiles/ab_sh.py): B = 1 024 0.42 ms (= the per-trajectory kernel: both wait for one 999-step matrix chain),
    // 2 048 0.45 vs 0.97 ms, 32 768 6.7 vs 9.4 ms.  DDP_SH_MIN_B moves the threshold (tests run it at B = 6).
    const char *sh_env = ddp_env(h, ENV_SH_MIN_B);
    const int sh_min = sh_env ? atoi(sh_env) : 1024;
    if ((force == 0 || force == 's') && d->B >= sh_min) {
        const int32_t *fb = nullptr;
        const int rs = ddp_launch_back_pass_sh(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, active, K, k, Quu, Vx, Vxx, dV, diverge, &fb);
        if (rs < 0) return rs;
        if (rs == 0) {
            const int rc = launch_back_pass_inner(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, fb, K, k, Quu, Vx, Vxx, dV, diverge);
            h->last_kernel[0] = "sh_back_kernel";
            return rc;
        }
    }
    return launch_back_pass_inner(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
}
