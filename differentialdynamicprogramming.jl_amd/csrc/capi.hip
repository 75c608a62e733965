// capi.hip — handle, memory/event helpers, host-pointer flavours and the standalone batched boxQP
// of the C ABI declared in include/ddp_amd.h.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <vector>
#include "ddp_internal.h"
#include "boxqp_dev.h"

static thread_local char g_err[1024] = "";

void ddp_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" {

const char *ddp_last_error(void) { return g_err; }
const char *ddp_version(void) { return "ddp_amd 0.3.0 (gfx950, fp64)"; }

static const char *const ddp_env_names[ENV_COUNT] = {"DDP_BACKPASS", "DDP_SH_MIN_B", "DDP_MX2", "DDP_DPPW", "DDP_DPPW_EXP", "DDP_MX_LDS", "DDP_Q4_EXP", "DDP_Q4_SINGLE", "DDP_Q4_LDS", "DDP_GPS_Q4", "DDP_GPS_Q4L", "DDP_DF_DENSE", "DDP_FORWARD", "DDP_FORWARD64", "DDP_FORWARD_FAST", "DDP_FORWARD_FUSE", "DDP_FORWARD_LANE", "DDP_FORWARD_PEND", "DDP_FORWARD_PIPE", "DDP_ILQG_COMPACT", "DDP_ILQG_LSGROUPS", "DDP_TEST_COMPACT_ALLOC_FAIL", "DDP_GPS_LANE", "DDP_FCOV_Q4", "DDP_FCOV_Q4L", "DDP_KL_LDS", "DDP_TEST_SH_ABORT", "DDP_SH_NT_MAX_B", "DDP_MXG_COAL", "DDP_FORWARD_MID", "DDP_PEND_CHUNK"};

int ddp_reload_env(ddp_handle h)
{
    if (!h) { ddp_set_error("null handle"); return -1; }
    for (int i = 0; i < ENV_COUNT; ++i) {
        const char *v = getenv(ddp_env_names[i]);
        h->envset[i] = v != nullptr;
        if (v) { strncpy(h->envv[i], v, sizeof h->envv[i] - 1); h->envv[i][sizeof h->envv[i] - 1] = 0; }
    }
    // the cost_diag verdicts are keyed by device addresses, which allocators recycle: a reload forgets them (so does ddp_free of a
    // cached address) — the next call with cost_diag = 1 looks at Q, R again
    for (auto &e : h->diag_cache) { e.Q = e.R = nullptr; e.n = e.m = e.ok = 0; }
    h->diag_next = 0;
    return 0;
}

const char *ddp_last_kernel(ddp_handle h, int which)
{
    if (!h || which < 0 || which > 1 || !h->last_kernel[which]) return "";
    return h->last_kernel[which];
}

int ddp_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

static int create_impl(int device, void *ext_stream, bool adopt, ddp_handle *out)
{
    DDP_CHECK(out, "ddp_create: out is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    DDP_CHECK(e == hipSuccess && c > 0, "ddp_create: no HIP device available (%s) — libddp_amd has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    DDP_CHECK(device >= 0 && device < c, "ddp_create: device %d out of range [0,%d)", device, c);
    DDP_HIP(hipSetDevice(device));
    ddp_handle h = new ddp_handle_s();
    h->device = device;
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    h->pad = nullptr; h->pad_bytes = 0; h->sink = nullptr; h->sh = nullptr; h->sh_bytes = 0; h->sh_timeouts = 0; h->sh_attr = false; h->ncu = 0; h->sched_aux = nullptr; h->diag_next = 0; h->diag_skip = 0; for (auto &e : h->diag_cache) { e.Q = e.R = nullptr; e.n = e.m = e.ok = 0; } h->last_kernel[0] = h->last_kernel[1] = nullptr; ddp_reload_env(h);
    h->h_pinned = nullptr;
    h->timing = nullptr; h->timing_cap = 0; h->tev_ok = false;
    h->owns_stream = !adopt;
    if (adopt) {
        h->stream = (hipStream_t)ext_stream;
    } else if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        ddp_set_error("ddp_create: hipStreamCreate failed");
        return -2;
    }
    if (hipHostMalloc((void **)&h->h_pinned, 256) != hipSuccess) h->h_pinned = nullptr;
    // + a flag word behind it (df.hip) + 1 KB that stays ZERO (offset 4352: loads of padded rows / columns are pointed at it, back_pass_mf2)
    if (hipMalloc(&h->sink, 4096 + 256 + 1024) != hipSuccess) h->sink = nullptr;
    else if (hipMemset(h->sink, 0, 4096 + 256 + 1024) != hipSuccess) { hipFree(h->sink); h->sink = nullptr; }
    *out = h;
    return 0;
}

int ddp_create(int device, ddp_handle *out) { return create_impl(device, nullptr, false, out); }
int ddp_create_with_stream(int device, void *hip_stream, ddp_handle *out) { return create_impl(device, hip_stream, true, out); }

int ddp_destroy(ddp_handle h)
{
    if (!h) return 0;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (h->scratch) hipFree(h->scratch);
    if (h->h_pinned) hipHostFree(h->h_pinned);
    if (h->sink) hipFree(h->sink);
    if (h->pad) hipFree(h->pad);
    if (h->sh) hipFree(h->sh);
    if (h->sched_aux) { hipStreamDestroy(h->sched_aux); hipEventDestroy(h->sched_ev[0]); hipEventDestroy(h->sched_ev[1]); }
    if (h->tev_ok) for (int e = 0; e < 4; ++e) hipEventDestroy(h->tev[e]);
    if (h->owns_stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int ddp_sync(ddp_handle h)
{
    DDP_DEVICE(h);
    DDP_CHECK(h, "ddp_sync: null handle");
    DDP_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

void *ddp_stream(ddp_handle h) { return h ? (void *)h->stream : nullptr; }

int ddp_malloc(ddp_handle h, size_t bytes, void **dptr)
{
    DDP_CHECK(h && dptr, "ddp_malloc: null argument");
    DDP_HIP(hipSetDevice(h->device));
    DDP_HIP(hipMalloc(dptr, bytes ? bytes : 8));
    return 0;
}
int ddp_free(ddp_handle h, void *dptr)
{
    DDP_DEVICE(h);
    DDP_CHECK(h, "ddp_free: null handle");
    if (dptr) {
        DDP_HIP(hipStreamSynchronize(h->stream));
        // a cost_diag verdict about a Q or R inside this allocation dies with it (the next allocation may get the address back)
        hipDeviceptr_t base = nullptr; size_t sz = 0;
        if (hipMemGetAddressRange(&base, &sz, (hipDeviceptr_t)dptr) != hipSuccess) { (void)hipGetLastError(); base = (hipDeviceptr_t)dptr; sz = 1; }
        const char *lo = (const char *)base, *hi = lo + sz;
        for (auto &e : h->diag_cache) {
            const char *q = (const char *)e.Q, *r = (const char *)e.R;
            if ((q && q >= lo && q < hi) || (r && r >= lo && r < hi)) { e.Q = e.R = nullptr; e.n = e.m = e.ok = 0; }
        }
        DDP_HIP(hipFree(dptr));
    }
    return 0;
}
// ---- page-locked host memory for results (ddp_amd.h).  Process-wide cache of freed blocks, keyed by their (2 MB-rounded) size: a host
// that calls the same entry point again and again gets the same blocks back, so neither the pinning nor the first touch of fresh pages
// is paid per call.  DDP_PINNED_CACHE_MB bounds the bytes kept in the cache (default 2048, read once).
namespace {
struct PinnedPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;      // size -> block
    std::map<void *, size_t> live;                  // block -> size
    size_t cached = 0;
    size_t cap()
    {
        // bytes of FREED blocks kept for reuse: DDP_PINNED_CACHE_MB, clamped to [0, 64 GB] (a negative or garbled value used to shift into
        // an enormous cap); default 2 048 — enough for the result arrays of a C2 pass (1.2 GB).  Blocks in use are the caller's arrays and
        // are not bounded here: every result of >= 1 MB that the ctypes / Julia hosts return lives in page-locked memory until collected.
        static const size_t cap_bytes = [] {
            const char *e = getenv("DDP_PINNED_CACHE_MB");
            long mb = 2048;
            if (e && *e) { char *end = nullptr; const long v = strtol(e, &end, 10); if (end != e && v >= 0) mb = v < 65536 ? v : 65536; }
            return (size_t)mb << 20;
        }();
        return cap_bytes;
    }
};
PinnedPool &pinned_pool() { static PinnedPool p; return p; }
}   // namespace

int ddp_host_alloc(size_t bytes, void **hptr)
{
    DDP_CHECK(hptr, "ddp_host_alloc: null argument");
    const size_t sz = ((bytes ? bytes : 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    PinnedPool &P = pinned_pool();
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.free_blocks.find(sz);
        if (it != P.free_blocks.end()) {
            *hptr = it->second;
            P.free_blocks.erase(it);
            P.cached -= sz;
            P.live[*hptr] = sz;
            return 0;
        }
    }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, sz, hipHostMallocPortable);
    if (e != hipSuccess) {                                      // make room: drop the cache and try once more
        (void)hipGetLastError();
        ddp_host_trim();
        e = hipHostMalloc(&p, sz, hipHostMallocPortable);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ddp_set_error("ddp_host_alloc: hipHostMalloc(%zu) failed: %s", sz, hipGetErrorString(e));
        return -2;
    }
    std::lock_guard<std::mutex> g(P.mu);
    P.live[p] = sz;
    *hptr = p;
    return 0;
}

int ddp_host_free(void *hptr)
{
    if (!hptr) return 0;
    PinnedPool &P = pinned_pool();
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.live.find(hptr);
        DDP_CHECK(it != P.live.end(), "ddp_host_free: %p was not allocated by ddp_host_alloc", hptr);
        sz = it->second;
        P.live.erase(it);
        if (P.cached + sz <= P.cap()) {
            P.free_blocks.insert({sz, hptr});
            P.cached += sz;
            return 0;
        }
    }
    DDP_HIP(hipHostFree(hptr));
    return 0;
}

int ddp_host_trim(void)
{
    PinnedPool &P = pinned_pool();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> g(P.mu);
        for (auto &kv : P.free_blocks) drop.push_back(kv.second);
        P.free_blocks.clear();
        P.cached = 0;
    }
    for (void *p : drop) (void)hipHostFree(p);
    return 0;
}

int ddp_memcpy_h2d(ddp_handle h, void *dst, const void *src, size_t bytes)
{
    DDP_DEVICE(h);
    DDP_CHECK(h, "ddp_memcpy_h2d: null handle");
    DDP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    DDP_HIP(hipStreamSynchronize(h->stream));
    return 0;
}
int ddp_memcpy_d2h(ddp_handle h, void *dst, const void *src, size_t bytes)
{
    DDP_DEVICE(h);
    DDP_CHECK(h, "ddp_memcpy_d2h: null handle");
    DDP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    DDP_HIP(hipStreamSynchronize(h->stream));
    return 0;
}
int ddp_memset(ddp_handle h, void *dst, int value, size_t bytes)
{
    DDP_DEVICE(h);
    DDP_CHECK(h, "ddp_memset: null handle");
    DDP_HIP(hipMemsetAsync(dst, value, bytes, h->stream));
    return 0;
}
int ddp_event_create(ddp_handle h, void **ev)
{
    DDP_CHECK(h && ev, "ddp_event_create: null argument");
    hipEvent_t e;
    // timing events without the system-scope cache flush of a default record (that flush made a record cost ~40 µs of stream time)
    DDP_HIP(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    *ev = (void *)e;
    return 0;
}
int ddp_event_destroy(ddp_handle h, void *ev)
{
    (void)h;
    if (ev) DDP_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
int ddp_event_record(ddp_handle h, void *ev)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && ev, "ddp_event_record: null argument");
    DDP_HIP(hipEventRecord((hipEvent_t)ev, h->stream));
    return 0;
}
int ddp_event_elapsed_ms(ddp_handle h, void *start, void *stop, float *ms)
{
    DDP_CHECK(h && start && stop && ms, "ddp_event_elapsed_ms: null argument");
    DDP_HIP(hipEventSynchronize((hipEvent_t)stop));
    DDP_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

}   // extern "C"

int ddp_scratch(ddp_handle h, size_t bytes, void **out)
{
    DDP_DEVICE(h);
    if (bytes > h->scratch_bytes) {
        DDP_HIP(hipStreamSynchronize(h->stream));
        if (h->scratch) DDP_HIP(hipFree(h->scratch));
        h->scratch = nullptr;
        h->scratch_bytes = 0;
        DDP_HIP(hipMalloc(&h->scratch, bytes));
        h->scratch_bytes = bytes;
    }
    *out = h->scratch;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// bump allocator over the handle's scratch for the host-pointer flavours
namespace {
#include "arena.h"
}   // namespace

extern "C" {

int ddp_back_pass_f64_dev(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                          const double *cxx, const double *cxu, const double *cuu, const double *fx,
                          const double *fu, const double *lambda, const double *lims, const double *u,
                          const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                          double *Vxx, double *dV, int32_t *diverge)
{
    DDP_CHECK(h && d, "back_pass: null handle/descriptor");
    DDP_CHECK(cx && cu && cxx && cxu && cuu && fx && fu && lambda, "back_pass: null input pointer");
    DDP_CHECK(K && k && Quu && Vx && Vxx && dV && diverge, "back_pass: null output pointer");
    return ddp_launch_back_pass(h, d, cx, cu, cxx, cxu, cuu, fx, fu, lambda, lims, u, active, K, k, Quu, Vx, Vxx, dV, diverge);
}

int ddp_back_pass_f64(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                      const double *cxx, const double *cxu, const double *cuu, const double *fx,
                      const double *fu, const double *lambda, const double *lims, const double *u,
                      double *K, double *k, double *Quu, double *Vx, double *Vxx, double *dV,
                      int32_t *diverge)
{
    DDP_CHECK(h && d, "back_pass: null handle/descriptor");
    const size_t n = d->n, m = d->m, N = d->N, B = d->B;
    const size_t fxc = (d->fx_tv ? N : 1) * (d->fx_batched ? B : 1), cc = (d->cost_tv ? N : 1) * (d->cost_batched ? B : 1);
    const size_t sizes_in[] = {n * N * B, m * N * B, n * n * cc, n * m * cc, m * m * cc, n * n * fxc, n * m * fxc, B, 2 * m, m * N * B};
    const size_t sizes_out[] = {m * n * N * B, m * N * B, m * m * N * B, n * N * B, n * n * N * B, 2 * B};
    Arena A; A.h = h;
    for (size_t s : sizes_in) A.want(s * 8);
    for (size_t s : sizes_out) A.want(s * 8);
    A.want(B * 4);
    int rc = A.commit();
    if (rc) return rc;
    const double *dcx = A.in(cx, sizes_in[0]), *dcu = A.in(cu, sizes_in[1]), *dcxx = A.in(cxx, sizes_in[2]),
                 *dcxu = A.in(cxu, sizes_in[3]), *dcuu = A.in(cuu, sizes_in[4]), *dfx = A.in(fx, sizes_in[5]),
                 *dfu = A.in(fu, sizes_in[6]), *dlam = A.in(lambda, sizes_in[7]),
                 *dlims = d->has_lims ? A.in(lims, sizes_in[8]) : nullptr, *du = d->has_lims ? A.in(u, sizes_in[9]) : nullptr;
    double *dK = A.outp(K, sizes_out[0]), *dk = A.outp(k, sizes_out[1]), *dQuu = A.outp(Quu, sizes_out[2]),
           *dVx = A.outp(Vx, sizes_out[3]), *dVxx = A.outp(Vxx, sizes_out[4]), *ddV = A.outp(dV, sizes_out[5]);
    int32_t *ddiv = A.outp(diverge, B);
    if ((rc = A.upload())) return rc;
    rc = ddp_back_pass_f64_dev(h, d, dcx, dcu, dcxx, dcxu, dcuu, dfx, dfu, dlam, dlims, du, nullptr, dK, dk, dQuu, dVx, dVxx, ddV, ddiv);
    if (rc) return rc;
    return A.download();
}

int ddp_forward_pass_f64(ddp_handle h, const ddp_problem *p, const double *K, const double *k,
                         const double *x0, const double *u, const double *x, const double *alpha,
                         int nalpha, const double *lims, double *xnew, double *unew, double *cnew,
                         double *csum)
{
    DDP_CHECK(h && p, "forward_pass: null handle/problem");
    DDP_CHECK(nalpha >= 1 && nalpha <= 16, "forward_pass: nalpha=%d out of [1,16]", nalpha);
    const size_t n = p->n, m = p->m, N = p->N, B = p->B, CL = ddp_cost_len(p), na = nalpha;
    const size_t dc = (p->dyn_tv ? N : 1) * (p->dyn_batched ? B : 1);
    Arena A; A.h = h;
    const size_t tot = m * n * N * B + m * N * B + n * B + m * N * B + n * N * B + 2 * m + n * n * dc + n * m * dc + n * n + m * m +
                       (n * N + m * N + CL + 1) * B * na;
    A.want(tot * 8 + 32 * 256);
    int rc = A.commit();
    if (rc) return rc;
    ddp_problem pd = *p;
    if (p->kind == DDP_PROBLEM_LQ) { pd.A = A.in(p->A, n * n * dc); pd.Bm = A.in(p->Bm, n * m * dc); }
    { const int rd_ = ddp_check_cost_diag_host(p); if (rd_) return rd_; }
    DiagVerified diag_verified_(h);                          // Q, R were tested on the host; the staged copies need no second test
    pd.Q = A.in(p->Q, n * n); pd.R = A.in(p->R, m * m);
    const double *dK = A.in(K, m * n * N * B), *dk = A.in(k, m * N * B), *dx0 = A.in(x0, n * B), *du = A.in(u, m * N * B),
                 *dx = A.in(x, n * N * B), *dl = A.in(lims, 2 * m);
    double *dxn = A.outp(xnew, n * N * B * na), *dun = A.outp(unew, m * N * B * na), *dcn = A.outp(cnew, CL * B * na),
           *dcs = A.outp(csum, B * na);
    // outputs the caller does not want still need device storage
    if (!dxn) dxn = (double *)A.take(n * N * B * na * 8);
    if (!dun) dun = (double *)A.take(m * N * B * na * 8);
    if (!dcn) dcn = (double *)A.take(CL * B * na * 8);
    if (!dcs) dcs = (double *)A.take(B * na * 8);
    if ((rc = A.upload())) return rc;
    rc = ddp_forward_pass_f64_dev(h, &pd, dK, dk, dx0, du, dx, alpha, nalpha, dl, nullptr, dxn, dun, dcn, dcs);
    if (rc) return rc;
    return A.download();
}

}   // extern "C"

// ---------------------------------------------------------------------------------------------
// standalone batched boxQP (src/boxQP.jl:29-188): one lane per problem
namespace {
template <int MM>
__global__ __launch_bounds__(DDP_WAVE) void boxqp_kernel(int m, int count, const double *Hg, const double *gg,
                                                         const double *log_, const double *upg, const double *x0g,
                                                         QPOptsDev o, double *xg, int32_t *resg, double *Hfg,
                                                         uint8_t *freeg)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    double H[MM * MM], R[MM * MM], g[MM], lo[MM], up[MM], x0[MM], x[MM];
#pragma unroll
    for (int c = 0; c < MM; ++c) {
        g[c] = (c < m) ? gg[(size_t)m * t + c] : 0.0;
        lo[c] = (c < m) ? log_[(size_t)m * t + c] : 0.0;
        up[c] = (c < m) ? upg[(size_t)m * t + c] : 0.0;
        x0[c] = (c < m) ? x0g[(size_t)m * t + c] : 0.0;
#pragma unroll
        for (int r = 0; r < MM; ++r) H[r + MM * c] = (r < m && c < m) ? Hg[(size_t)m * m * t + r + m * c] : 0.0;
    }
    unsigned clamped;
    int iters;
    int res;
    if constexpr (MM == 2) { double ri2[2]; res = boxqp_dev2(H, g, lo, up, x0, o, x, R, ri2, clamped, iters); }     // the straight-line form of the backward kernels
    else res = boxqp_dev<MM>(m, H, g, lo, up, x0, o, x, R, clamped, iters);
    resg[t] = res;
    // compact the masked factor to the leading nfree x nfree block (the reference's Hfree)
    int pos[MM], nf = 0;
#pragma unroll
    for (int c = 0; c < MM; ++c) {
        const bool fr = (c < m) && !((clamped >> c) & 1u);
        pos[c] = fr ? nf : -1;
        nf += fr ? 1 : 0;
    }
    for (int e = 0; e < m * m; ++e) Hfg[(size_t)m * m * t + e] = 0.0;
#pragma unroll
    for (int c = 0; c < MM; ++c) {
        if (c < m) {
            xg[(size_t)m * t + c] = x[c];
            freeg[(size_t)m * t + c] = (pos[c] >= 0) ? 1 : 0;
#pragma unroll
            for (int r = 0; r < MM; ++r)
                if (r <= c && pos[r] >= 0 && pos[c] >= 0) Hfg[(size_t)m * m * t + pos[r] + m * pos[c]] = R[r + MM * c];
        }
    }
}
}   // namespace

extern "C" {

int ddp_boxqp_f64_dev(ddp_handle h, int m, int count, const double *H, const double *g, const double *lower,
                      const double *upper, const double *x0, const ddp_qp_opts *opts, double *x,
                      int32_t *result, double *Hfree, uint8_t *free_out)
{
    DDP_DEVICE(h);
    DDP_CHECK(h, "boxqp: null handle");
    DDP_CHECK(m >= 1 && m <= DDP_QP_MAX_M, "boxqp: m=%d out of [1,%d]", m, DDP_QP_MAX_M);
    DDP_CHECK(count >= 1, "boxqp: count=%d", count);
    QPOptsDev o = {100, 1e-8, 1e-8, 0.6, 1e-22, 0.1};
    if (opts) o = {opts->maxIter, opts->minGrad, opts->minRelImprove, opts->stepDec, opts->minStep, opts->Armijo};
    if (m > DDP_MAX_M) return ddp_launch_boxqp_big(h, m, count, H, g, lower, upper, x0, o, x, result, Hfree, free_out);   // one work-group per problem
    const dim3 grid((count + DDP_WAVE - 1) / DDP_WAVE), block(DDP_WAVE);
#define DDP_QP_CASE(M_)                                                                                           \
    case M_:                                                                                                      \
        hipLaunchKernelGGL((boxqp_kernel<M_>), grid, block, 0, h->stream, m, count, H, g, lower, upper, x0, o, x, \
                           result, Hfree, free_out);                                                              \
        break;
    switch (m) {
        DDP_QP_CASE(1) DDP_QP_CASE(2) DDP_QP_CASE(3) DDP_QP_CASE(4)
        DDP_QP_CASE(5) DDP_QP_CASE(6) DDP_QP_CASE(7) DDP_QP_CASE(8)
    }
#undef DDP_QP_CASE
    DDP_HIP(hipGetLastError());
    return 0;
}

int ddp_boxqp_f64(ddp_handle h, int m, int count, const double *H, const double *g, const double *lower,
                  const double *upper, const double *x0, const ddp_qp_opts *opts, double *x, int32_t *result,
                  double *Hfree, uint8_t *free_out)
{
    DDP_CHECK(h, "boxqp: null handle");
    DDP_CHECK(m >= 1 && count >= 1, "boxqp: bad sizes");
    const size_t M = m, C = count;
    Arena A; A.h = h;
    A.want((2 * M * M * C + 5 * M * C) * 8 + C * 4 + M * C + 16 * 256);
    int rc = A.commit();
    if (rc) return rc;
    const double *dH = A.in(H, M * M * C), *dg = A.in(g, M * C), *dlo = A.in(lower, M * C), *dup = A.in(upper, M * C),
                 *dx0 = A.in(x0, M * C);
    double *dx = A.outp(x, M * C), *dHf = A.outp(Hfree, M * M * C);
    int32_t *dr = A.outp(result, C);
    uint8_t *df = A.outp(free_out, M * C);
    if ((rc = A.upload())) return rc;
    rc = ddp_boxqp_f64_dev(h, m, count, dH, dg, dlo, dup, dx0, opts, dx, dr, dHf, df);
    if (rc) return rc;
    return A.download();
}

}   // extern "C"
