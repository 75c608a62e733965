// ddp_internal.h — shared between the translation units of libddp_amd.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/ddp_amd.h"

#define DDP_WAVE 64
#define DDP_MAX_N_GENERIC 32      // run-time-sized kernels: n <= 32, m <= DDP_MAX_M

// The DDP_* switches of the dispatchers (kernel choice for A/B timing and for the tests that force every code path) are read from the
// environment ONCE per handle (ddp_create) and again on ddp_reload_env(): no launch calls getenv, and a setenv() in another thread
// cannot race with a launch.  ddp_env() returns the cached value or nullptr.
enum ddp_env_id { ENV_BACKPASS, ENV_SH_MIN_B, ENV_MX2, ENV_DPPW, ENV_DPPW_EXP, ENV_MX_LDS, ENV_Q4_EXP, ENV_Q4_SINGLE, ENV_Q4_LDS, ENV_GPS_Q4, ENV_GPS_Q4L, ENV_DF_DENSE, ENV_FORWARD, ENV_FORWARD64, ENV_FORWARD_FAST, ENV_FORWARD_FUSE, ENV_FORWARD_LANE, ENV_FORWARD_PEND, ENV_FORWARD_PIPE, ENV_ILQG_COMPACT, ENV_ILQG_LSGROUPS, ENV_TEST_COMPACT_ALLOC_FAIL, ENV_GPS_LANE, ENV_FCOV_Q4, ENV_FCOV_Q4L, ENV_KL_LDS, ENV_TEST_SH_ABORT, ENV_SH_NT_MAX_B, ENV_MXG_COAL, ENV_FORWARD_MID, ENV_PEND_CHUNK, ENV_COUNT };

struct ddp_handle_s {
    int          device;
    hipStream_t  stream;
    bool         owns_stream;
    // scratch owned by the handle (host-pointer entry points, iLQG driver)
    void        *scratch;
    size_t       scratch_bytes;
    int32_t     *h_pinned;        // small pinned buffer for polling
    void        *pad;             // operands / results of a backward pass padded to even sizes (back_pass.hip), grown on demand
    size_t       pad_bytes;
    void        *sh;              // back_pass_sh.hip: control block, work items, record streams of the shared-LTI backward pass
    size_t       sh_bytes;
    int          sh_timeouts;     // timed-out tiles counted by control blocks that have been freed (ddp_sh_timeouts)
    bool         sh_attr;         // its dynamic-LDS attribute has been set on this device
    int          ncu;             // compute units of the device (0: not asked yet)
    struct { const void *Q, *R; int n, m, ok; } diag_cache[8];      // verdicts of ddp_check_cost_diag (forward_pass.hip)
    int          diag_next;
    int          diag_skip;       // > 0: a host-pointer flavour has verified Q, R on the host and staged them itself (addresses recycle)
    hipStream_t  sched_aux;       // ilqg.hip, slot scheduler: side stream of the initial rollouts + its two events (created on first use)
    hipEvent_t   sched_ev[2];
    char         envv[ENV_COUNT][24];
    bool         envset[ENV_COUNT];
    const char  *last_kernel[2];  // what the last backward / forward dispatch launched (ddp_last_kernel)
    void        *sink;            // 4 KB of device memory that masked-out lanes may write (stores without an exec-mask branch)
    double      *timing;          // ddp_ilqg_set_timing: host buffer [3, timing_cap] or NULL
    int          timing_cap;
    hipEvent_t   tev[4];          // created on first use
    bool         tev_ok;
};

void ddp_set_error(const char *fmt, ...);
static inline const char *ddp_env(ddp_handle h, int id) { return h->envset[id] ? h->envv[id] : nullptr; }

#define DDP_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ddp_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                          __LINE__);                                                          \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)

#define DDP_CHECK(cond, ...)                                                                  \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            ddp_set_error(__VA_ARGS__);                                                       \
            return -1;                                                                        \
        }                                                                                     \
    } while (0)

// every public entry point makes the handle's device current first: a process may hold handles on several devices, and scratch
// allocations / kernel launches go to the CURRENT device of the calling thread.  A handle is single-threaded (ddp_amd.h).
#define DDP_DEVICE(h)                                                                         \
    do {                                                                                      \
        if (!(h)) { ddp_set_error("null handle"); return -1; }                                \
        DDP_HIP(hipSetDevice((h)->device));                                                   \
    } while (0)

// grows the handle's scratch to at least `bytes` (contents not preserved)
int ddp_scratch(ddp_handle h, size_t bytes, void **out);

// ddp_problem::cost_diag = 1 declares Q and R diagonal (the fused rollout cost reads only the diagonals): verified on the first call
// with a (Q, R) pointer pair — one small device-to-host copy — and cached per handle; 0 ok, < 0 refused
int ddp_check_cost_diag(ddp_handle h, const ddp_problem *p);
// the same test on HOST copies of Q, R (the host-pointer flavours, before they stage the problem); while a DiagVerified lives the
// device-side test is skipped for this handle (the staging addresses are recycled from call to call: a pointer-keyed verdict would go stale)
int ddp_check_cost_diag_host(const ddp_problem *p);
struct DiagVerified { ddp_handle h; explicit DiagVerified(ddp_handle h_) : h(h_) { ++h->diag_skip; } ~DiagVerified() { --h->diag_skip; } };

// kernel launchers (each in its own .hip)
int ddp_launch_back_pass(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                         const double *cxx, const double *cxu, const double *cuu, const double *fx,
                         const double *fu, const double *lambda, const double *lims, const double *u,
                         const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                         double *Vxx, double *dV, int32_t *diverge);
int ddp_launch_back_pass_gps(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const ddp_kl_cost_terms *kl, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Quui, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge);
// one lane per trajectory (n = 4, m <= 2); returns 1 when the shape is not handled
int ddp_launch_back_pass_gps_lane(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const ddp_kl_cost_terms *kl, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Quui, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge);
// n=10, m=2, no limits: one wave per trajectory, all matrices of a step in one 16x16 fp64 MFMA tile; 1 = not applicable
int ddp_launch_back_pass_mx(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                            const double *cxx, const double *cxu, const double *cuu, const double *fx,
                            const double *fu, const double *lambda, const int32_t *active, double *K,
                            double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge);
// any n <= 10, m <= 2 without limits inside the same tile (run-time sizes, results straight to global memory); 1 = not applicable
int ddp_launch_back_pass_mxr(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const int32_t *active, double *K,
                             double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge);

// back_pass_mxg.hip: the tile kernel for any n <= 12, m <= 4 (m <= 3 above n = 8), with or without limits; 1 = not applicable
int ddp_launch_back_pass_mxg(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u, const int32_t *active, double *K,
                             double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge);

// the same tile arithmetic with a chain wave + a write-back wave per trajectory (back_pass_mx2.hip); 1 = not applicable
int ddp_launch_back_pass_mx2(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const int32_t *active, double *K,
                             double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge);

// shared time-invariant operands (n=10, m=2, no limits): the matrix recursion once per distinct λ, an affine chain per trajectory
// (back_pass_sh.hip); 1 = not applicable; 0 = launched, the trajectories it left out are flagged in *fb_active
// back_pass_row.hip: the 16-lane-row kernel compiled for padded sizes — any n <= 14, m <= 4 with n + m <= 15; 1 = not applicable
int ddp_launch_back_pass_row(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge);
// back_pass_mid.hip: one wave per trajectory, the products on the fp64 matrix cores with LDS operands — any n <= 32, m <= 8; 1 = not applicable
int ddp_launch_back_pass_mid(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge);
// forward_pass_row.hip: the 16-lane-row rollout compiled for padded sizes (LQ problems, n <= 14, m <= 4); 1 = not applicable
int ddp_launch_forward_row(ddp_handle h, const ddp_problem *p, const double *K, const double *k, const double *x0,
                           const double *u, const double *x, const double *alpha, int nalpha, const double *lims,
                           const int32_t *active, double *xnew, double *unew, double *cnew, double *csum);
extern "C" int ddp_sh_max_tiles(int B, int ncu);      // capacity of its work-item list: the most consumer tiles any grouping of B trajectories can make
int ddp_launch_back_pass_sh(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                            const double *cxx, const double *cxu, const double *cuu, const double *fx,
                            const double *fu, const double *lambda, const int32_t *active, double *K,
                            double *k, double *Quu, double *Vx, double *Vxx, double *dV, int32_t *diverge,
                            const int32_t **fb_active);

// 16-lane DPP-row backward pass (4 trajectories per wave); returns 1 when the shape has no such kernel
int ddp_launch_back_pass_dpp(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge);
// n = 4, m = 1 on v_mfma_f64_4x4x4_4b, one trajectory per MFMA block (back_pass_q4.hip); returns 1 for any other shape
int ddp_launch_back_pass_q4(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                            const double *cxx, const double *cxu, const double *cuu, const double *fx,
                            const double *fu, const double *lambda, const double *lims, const double *u,
                            const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                            double *Vxx, double *dV, int32_t *diverge);
// large states (even n <= 64, even m <= 8): 256-thread work-group per trajectory; returns 1 when not applicable
int ddp_launch_back_pass_big(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge);
// n=64, m=8 on the fp64 matrix cores (v_mfma_f64_16x16x4_f64); returns 1 for any other shape
// 32 < n <= 64, m <= 8 at run time: every product on the fp64 matrix cores (back_pass_mf2.hip); 1 = not applicable, 2 = (64, 8) with real
// limits and `defer_64x8_lims`: nothing launched, the caller takes the round-5 kernel
int ddp_launch_back_pass_mf2(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                             const double *cxx, const double *cxu, const double *cuu, const double *fx,
                             const double *fu, const double *lambda, const double *lims, const double *u,
                             const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                             double *Vxx, double *dV, int32_t *diverge, bool defer_64x8_lims);
int ddp_launch_back_pass_mfma(ddp_handle h, const ddp_bp_desc *d, const double *cx, const double *cu,
                              const double *cxx, const double *cxu, const double *cuu, const double *fx,
                              const double *fu, const double *lambda, const double *lims, const double *u,
                              const int32_t *active, double *K, double *k, double *Quu, double *Vx,
                              double *Vxx, double *dV, int32_t *diverge);
// 16-lane DPP-row forward pass + separate cost kernel; returns 1 when the shape has no such kernel
int ddp_launch_forward_dpp(ddp_handle h, const ddp_problem *p, const double *K, const double *k, const double *x0,
                            const double *u, const double *x, const double *alpha, int nalpha, const double *lims,
                            const int32_t *active, double *xnew, double *unew, double *cnew, double *csum);

// LQ n=10/m=2 rollout as a producer/consumer pipeline of one work-group per 4 rollouts (forward_pass_pipe.hip); 1 = not applicable
struct QPOptsDev;
int ddp_launch_boxqp_big(ddp_handle h, int m, int count, const double *H, const double *g, const double *lower, const double *upper,
                         const double *x0, const QPOptsDev &o, double *x, int32_t *result, double *Hfree, uint8_t *free_out);   // boxqp_big.hip
int ddp_launch_forward_pipe(ddp_handle h, const ddp_problem *p, const double *K, const double *k, const double *x0,
                            const double *u, const double *x, const double *alpha, int nalpha, const double *lims,
                            const int32_t *active, double *xnew, double *unew, double *cnew, double *csum);

// 1/sqrt(x): hardware estimate (v_rsq_f64) + two Newton steps -> ~1 ulp.  The caller checks x > 0.
__device__ __forceinline__ double ddp_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}

// one wave per rollout for large states (LQ family, n <= 64); returns 1 when not applicable
int ddp_launch_forward_big(ddp_handle h, const ddp_problem *p, const double *K, const double *k, const double *x0,
                           const double *u, const double *x, const double *alpha, int nalpha, const double *lims,
                           const int32_t *active, double *xnew, double *unew, double *cnew, double *csum);

// Upper Cholesky of H (column-major M x M, upper triangle read) with RECIPROCAL pivots: fills the strictly upper
// entries of R and ri[j] = 1/R[j][j].  Division-free (v_rsq_f64 + Newton), which matters where every lane repeats
// the factorisation.  Returns 0, or j+1 for the first non-positive pivot (LAPACK potrf semantics).
template <int M>
__device__ __forceinline__ int ddp_chol_rinv(const double (&H)[M * M], double (&R)[M * M], double (&ri)[M])
{
    int fail = 0;
#pragma unroll
    for (int c = 0; c < M; ++c) {
        double ajj = H[c + M * c];
#pragma unroll
        for (int k2 = 0; k2 < c; ++k2) ajj -= R[k2 + M * c] * R[k2 + M * c];
        if (!(ajj > 0.0) && fail == 0) fail = c + 1;
        ri[c] = ddp_rsqrt(ajj);
#pragma unroll
        for (int c2 = c + 1; c2 < M; ++c2) {
            double s = H[c + M * c2];
#pragma unroll
            for (int k2 = 0; k2 < c; ++k2) s -= R[k2 + M * c] * R[k2 + M * c2];
            R[c + M * c2] = s * ri[c];
        }
    }
    return fail;
}
// b <- -(R'R)\b with the factor of ddp_chol_rinv
template <int M>
__device__ __forceinline__ void ddp_rsolve_neg(const double (&R)[M * M], const double (&ri)[M], double (&b)[M])
{
#pragma unroll
    for (int c = 0; c < M; ++c) {
        double s = b[c];
#pragma unroll
        for (int k2 = 0; k2 < c; ++k2) s -= R[k2 + M * c] * b[k2];
        b[c] = s * ri[c];
    }
#pragma unroll
    for (int c = M - 1; c >= 0; --c) {
        double s = b[c];
#pragma unroll
        for (int k2 = c + 1; k2 < M; ++k2) s -= R[c + M * k2] * b[k2];
        b[c] = s * ri[c];
    }
#pragma unroll
    for (int c = 0; c < M; ++c) b[c] = -b[c];
}

// Hand-off between the lanes of ONE wavefront through LDS.  The kernels that use it run one wave per
// work-group, so no s_barrier is needed: LDS operations of a wave execute in issue order, the only
// requirements are (a) the compiler must not move LDS accesses across the hand-off and (b) nothing may
// wait on outstanding GLOBAL stores/loads here (__syncthreads() would emit s_waitcnt vmcnt(0) and stall
// every hand-off on the HBM round trip of the step's output stores).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
