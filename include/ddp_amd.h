/*
 * ddp_amd.h — C ABI of libddp_amd.so: the MI355X (gfx950) implementation of the iLQG hot path of
 * baggepinnen/DifferentialDynamicProgramming.jl v0.5.0 (back_pass / boxQP / forward_pass and the
 * iLQG iteration around them), batched over B independent trajectories.
 *
 * The reference has no FFI for this path (it is plain Julia); each entry point below replaces the
 * Julia function cited next to it and is what a Julia `@ccall` wrapper (INTEGRATION.md,
 * differentialdynamicprogramming.jl_amd/julia/DDPAmd.jl) or any other host binds.
 *
 * Conventions
 *  - every array is fp64, Julia column-major, batch index slowest:  K[m,n,N,B]  is  B  copies of the
 *    reference's  K[m,n,N]  back to back, so B == 1 accepts the reference's arrays unchanged;
 *  - `_dev` entry points take DEVICE pointers (caller-owned, never freed here) and are asynchronous
 *    on the handle's HIP stream; the plain entry points take HOST pointers, copy H2D, run, copy D2H
 *    and synchronise;
 *  - return value: 0 = ok, < 0 = argument or HIP error (text via ddp_last_error()).  Numerical failure
 *    is NOT an error: it is reported per trajectory in `diverge` (backward_pass.jl:37-38,53-56) and
 *    `result` (boxQP.jl:172-179) exactly like the reference;
 *  - one handle = one HIP stream + scratch; a handle is not thread-safe, distinct handles are.
 *  - there is NO CPU fallback: without a gfx950 device ddp_create() fails.
 */
#ifndef DDP_AMD_H
#define DDP_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ddp_handle_s *ddp_handle;

/* ---- library / handle --------------------------------------------------------------------- */
const char *ddp_last_error(void);
const char *ddp_version(void);
int  ddp_device_count(void);
int  ddp_create(int device, ddp_handle *out);
/* adopt a caller-owned hipStream_t (e.g. the host framework's current stream) instead of creating one;
 * the stream is not destroyed by ddp_destroy() */
int  ddp_create_with_stream(int device, void *hip_stream, ddp_handle *out);
int  ddp_destroy(ddp_handle h);
int  ddp_sync(ddp_handle h);
/* The DDP_* environment switches (kernel choice for A/B timing and for the tests that force every code path; none is needed in
 * production) are read ONCE, in ddp_create(); no launch calls getenv().  ddp_reload_env() reads them again for this handle.          */
int  ddp_reload_env(ddp_handle h);
/* name of the kernel the last back_pass (which = 0) / forward_pass (which = 1) dispatch of this handle launched ("" before the first
 * one): a debug query — the tests assert through it that the timed path is the one they checked                                    */
const char *ddp_last_kernel(ddp_handle h, int which);
/* The shared-operand backward pass (one fx, fu, cxx, cxu, cuu for the batch) hands work between work-groups of one launch; every such
 * wait is time-bounded (4 s).  A tile whose wait ran out gives its trajectories to the per-trajectory kernels launched behind it — the
 * results stay correct — and is counted: this returns the number of such tiles since ddp_create (synchronises the stream; 0 in a
 * healthy run, < 0 on error).  A debug query like ddp_last_kernel.                                                                  */
int  ddp_sh_timeouts(ddp_handle h);
/* What the tiles counted by ddp_sh_timeouts were waiting for: up to 8 records of 8 ints — {work-group, group, chunk (of 8 time steps)
 * waited for, progress word last seen (chunks published | 1 << 30: group finished), milliseconds waited, XCD, groups of that launch,
 * launch number} — of the first tiles that gave up since the control block was allocated, followed by the 16 progress words as they
 * stand now.  out: cap >= 80 ints.  Returns the number of records (0 in a healthy run), < 0 on error; synchronises the stream.  (Own
 * protocol of the shared-operand kernel: no counterpart in the reference.)                                                              */
int  ddp_sh_timeout_info(ddp_handle h, int *out, int cap);
void *ddp_stream(ddp_handle h);                 /* the hipStream_t of the handle */
/* device memory helpers for hosts without their own allocator (the Julia wrapper, tests) */
int  ddp_malloc(ddp_handle h, size_t bytes, void **dptr);
int  ddp_free(ddp_handle h, void *dptr);
int  ddp_memcpy_h2d(ddp_handle h, void *dst, const void *src, size_t bytes);   /* synchronous */
int  ddp_memcpy_d2h(ddp_handle h, void *dst, const void *src, size_t bytes);   /* synchronous */
int  ddp_memset(ddp_handle h, void *dst, int value, size_t bytes);             /* async on stream */
/* Page-locked host memory for the RESULT arrays of the host-pointer entry points.  The link moves 56 GB/s into pinned or already-touched
 * pages; what a host-pointer call used to pay for is the first touch of freshly allocated pageable result arrays (1.2 GB per C2 pass:
 * 8-13 k passes/s against a link bound of ~46 k).  Blocks freed with ddp_host_free are kept in a process-wide cache (same size -> same
 * block on the next call: no pinning, no page faults), bounded by DDP_PINNED_CACHE_MB (default 2048, clamped to [0, 65536]); ddp_host_trim
 * empties the cache.  Blocks that are in use (the arrays the hosts returned) are page-locked for as long as the caller keeps them.
 * Any host pointer still works everywhere — this is an allocator the hosts (ctypes mirror, DDPAmd.jl) use for what they return.       */
int  ddp_host_alloc(size_t bytes, void **hptr);
int  ddp_host_free(void *hptr);
int  ddp_host_trim(void);
/* HIP events on the handle's stream (bench.py's per-kernel timing) */
int  ddp_event_create(ddp_handle h, void **ev);
int  ddp_event_destroy(ddp_handle h, void *ev);
int  ddp_event_record(ddp_handle h, void *ev);
int  ddp_event_elapsed_ms(ddp_handle h, void *start, void *stop, float *ms);   /* syncs on `stop` */

/* ---- back_pass — replaces back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u) ---------------
 * reference: src/backward_pass.jl:217-252 (LTI), :162-177 (LTV, time-invariant cost),
 *            :179-215 (LTV, time-varying cost), shared tail @end_backward_pass :28-79,
 *            called from src/iLQG.jl:237.                                                        */
typedef struct {
    int n, m, N, B;        /* state dim, control dim, time steps (= size(u,2)), batch              */
    int fx_tv;             /* 0: fx[n,n], fu[n,m]           1: fx[n,n,N], fu[n,m,N]                */
    int fx_batched;        /* 0: one fx/fu shared by the batch   1: one per trajectory (batch slowest) */
    int cost_tv;           /* 0: cxx[n,n], cxu[n,m], cuu[m,m]    1: [..,N]                          */
    int cost_batched;      /* as fx_batched for cxx/cxu/cuu                                         */
    int regType;           /* 1 or 2 (backward_pass.jl:245-247)                                     */
    int has_lims;          /* 0: `lims == []`   1: lims[m,2] given (the lims[1,1] > lims[1,2] test of
                              backward_pass.jl:31 is applied on the device as well)                 */
} ddp_bp_desc;

/* inputs : cx[n,N,B] cu[m,N,B]  cxx/cxu/cuu/fx/fu per the flags  lambda[B]  lims[m,2] (shared) u[m,N,B]
 *          `active` (may be NULL): int32[B]; trajectories with active[b]==0 are skipped entirely
 * outputs: K[m,n,N,B] k[m,N,B] Quu[m,m,N,B] Vx[n,N,B] Vxx[n,n,N,B] dV[2,B] diverge int32[B]
 *          (diverge: 0 ok, else the 1-based failing time index; outputs earlier in time than the
 *          failing step are zero like the reference's zero-initialised arrays)
 * shapes : m <= DDP_MAX_M (8); n <= 64 (kernel families by size: n = 10/m = 2, n = 4/m = 1, any n <= 12/m <= 4, n <= 14/m <= 4, n <= 32/m <= 8,
 *          32 < n <= 64 with m <= 8 at run time on the matrix-core kernel back_pass_mf2: padded to 16-row tiles and 8 controls inside the LDS
 *          only, no scratch on the handle).  Larger n or m: return code < 0.
 *          back_pass_gps: n <= 32.                                                                     */
int ddp_back_pass_f64_dev(ddp_handle h, const ddp_bp_desc *d,
                          const double *cx, const double *cu, const double *cxx, const double *cxu,
                          const double *cuu, const double *fx, const double *fu,
                          const double *lambda, const double *lims, const double *u,
                          const int32_t *active,
                          double *K, double *k, double *Quu, double *Vx, double *Vxx, double *dV,
                          int32_t *diverge);
int ddp_back_pass_f64(ddp_handle h, const ddp_bp_desc *d,
                      const double *cx, const double *cu, const double *cxx, const double *cxu,
                      const double *cuu, const double *fx, const double *fu,
                      const double *lambda, const double *lims, const double *u,
                      double *K, double *k, double *Quu, double *Vx, double *Vxx, double *dV,
                      int32_t *diverge);

/* ---- boxQP — replaces boxQP(H,g,lower,upper,x0) ------------------------------------------------
 * reference: src/boxQP.jl:29-188, called from src/backward_pass.jl:49.
 * `count` independent problems of dimension m <= DDP_QP_MAX_M (m <= DDP_MAX_M, the sizes the backward pass uses: one
 * lane per problem; larger m — upstream's demoQP runs m = 500, boxQP.jl:190-199 — one work-group per problem):
 * H[m,m,count] g/lower/upper/x0[m,count] -> x[m,count] result int32[count]
 * Hfree[m,m,count] (leading nfree x nfree block = upper Cholesky factor of H[free,free], zero elsewhere)
 * free uint8[m,count].                                                                             */
#define DDP_MAX_M 8
#define DDP_QP_MAX_M 1024
typedef struct {
    int    maxIter;        /* 100   */
    double minGrad;        /* 1e-8  */
    double minRelImprove;  /* 1e-8  */
    double stepDec;        /* 0.6   */
    double minStep;        /* 1e-22 */
    double Armijo;         /* 0.1   */
} ddp_qp_opts;
int ddp_boxqp_f64_dev(ddp_handle h, int m, int count, const double *H, const double *g,
                      const double *lower, const double *upper, const double *x0,
                      const ddp_qp_opts *opts /* NULL = defaults */,
                      double *x, int32_t *result, double *Hfree, uint8_t *free_out);
int ddp_boxqp_f64(ddp_handle h, int m, int count, const double *H, const double *g,
                  const double *lower, const double *upper, const double *x0,
                  const ddp_qp_opts *opts,
                  double *x, int32_t *result, double *Hfree, uint8_t *free_out);

/* ---- forward_pass — replaces forward_pass(traj_new,x0,u,x,α,f,costfun,lims,diff) ----------------
 * reference: src/forward_pass.jl:9-33, called from src/iLQG.jl:185,268.
 * The user closures f / costfun cannot run on the GPU; registered problem families stand in:
 *   DDP_PROBLEM_LQ        x+ = A x + B u,  cost_i = .5 x_i'Q x_i + .5 u_i'R u_i
 *                         (src/demo_linear.jl:42-49; cost returned per time step, the sum is the
 *                          reference's scalar)
 *   DDP_PROBLEM_PENDCART  explicit-Euler pendulum on a cart (src/system_pendcart.jl:83-89) with
 *                         cost vector of length N+1 (src/system_pendcart.jl:97-106)               */
enum { DDP_PROBLEM_LQ = 0, DDP_PROBLEM_PENDCART = 1 };
typedef struct {
    int kind;
    int n, m, N, B;
    /* LQ */
    const double *A;       /* [n,n] | [n,n,N] | [n,n,B] | [n,n,N,B] per dyn_tv / dyn_batched        */
    const double *Bm;      /* [n,m] ...                                                              */
    int dyn_tv, dyn_batched;
    const double *Q;       /* [n,n] (shared)                                                         */
    const double *R;       /* [m,m] (shared)                                                         */
    /* pendcart (n = 4, m = 1; Q [4,4], R [1,1] above) */
    double g, l, h, d;
    double goal[4];
    /* 1: the caller declares Q and R DIAGONAL (true for the reference's demos: Q = h·I, R = 0.1h·I, Q = diag(10,1,2,1)); the rollout
     * kernels of the n = 10 / m = 2 and pendcart shapes then evaluate the cost themselves from the values they hold instead of a
     * second kernel re-reading xnew, unew (only the diagonals are read).  0: general Q, R.  The declaration IS verified: the host-pointer
     * entry points test the host copies; the _dev entry points look at Q, R once per (Q, R) address pair and handle (one small
     * device-to-host copy; a PASS is cached — a caller that rewrites Q or R in place must keep the flag honest; ddp_free of
     * the allocation holding Q or R, ddp_reload_env and a new handle forget it; a refusal is never cached).  A full Q or R with cost_diag = 1 is refused (< 0).  Any value but 0 / 1 is refused,
     * so a struct that was not zero-initialised — or a caller built against the 0.1.0 layout, which ended at goal[] — fails loudly.   */
    int cost_diag;
    /* diff_fun (src/forward_pass.jl:19, iLQG.jl:160: `K*diff_fun(x̂, x)`; default `-`).  A closure cannot cross the C ABI; what stands in
     * is subtraction with the coordinates named in this bit mask (bit j = state j, n <= 32) wrapped to [-π, π]:
     *     d = x̂_j - x_j;  d - 2π·rint(d / 2π)        (Julia: rem2pi(x̂[j] - x[j], RoundNearest))
     * — the usual reason the hook exists (angles).  0 = the reference's default.  Bits at or above n are refused (a struct that was not
     * zero-initialised fails loudly); with a non-zero mask the rollout runs in the run-time-sized kernel (n <= 32).                 */
    uint32_t diff_wrap;
} ddp_problem;

/* cost vector length per trajectory: LQ -> N, pendcart -> N+1 */
int ddp_cost_len(const ddp_problem *p);

/* All `nalpha` step sizes are rolled out concurrently (one work-group slice per (trajectory, α)).
 * inputs : K[m,n,N,B], k[m,N,B] (both NULL = empty policy, forward_pass.jl:17), x0[n,B], u[m,N,B],
 *          x[n,N,B] (may be NULL with an empty policy), alpha[nalpha] (HOST pointer, <= 16 values),
 *          lims[m,2] or NULL, active int32[B] or NULL
 * outputs: xnew[n,N,B,nalpha], unew[m,N,B,nalpha], cnew[CL,B,nalpha], csum[B,nalpha] (= sum(cnew))  */
int ddp_forward_pass_f64_dev(ddp_handle h, const ddp_problem *p,
                             const double *K, const double *k, const double *x0, const double *u,
                             const double *x, const double *alpha, int nalpha, const double *lims,
                             const int32_t *active,
                             double *xnew, double *unew, double *cnew, double *csum);
int ddp_forward_pass_f64(ddp_handle h, const ddp_problem *p,
                         const double *K, const double *k, const double *x0, const double *u,
                         const double *x, const double *alpha, int nalpha, const double *lims,
                         double *xnew, double *unew, double *cnew, double *csum);

/* ---- df of the registered families (the `df` closure; STEP 1 of src/iLQG.jl:225-229) -------------
 * LQ: cx = Q x, cu = R u (src/demo_linear.jl:35-41).  pendcart: also fx[4,4,N,B], fu[4,1,N,B] by
 * exp of the 5x5 block matrix (src/system_pendcart.jl:137-154).  fx/fu may be NULL for LQ.         */
int ddp_df_f64_dev(ddp_handle h, const ddp_problem *p, const double *x, const double *u,
                   const int32_t *active, double *cx, double *cu, double *fx, double *fu);
int ddp_df_f64(ddp_handle h, const ddp_problem *p, const double *x, const double *u,
               double *cx, double *cu, double *fx, double *fu);

/* ---- iLQG — replaces iLQG(f,costfun,df,x0,u0; lims, ...) for registered families ----------------
 * reference: src/iLQG.jl:143-341.  Every trajectory of the batch runs its own iLQG (own λ, dλ,
 * line search, termination) with state resident on the device; the host only launches kernels
 * and polls one counter per iteration.                                                            */
typedef struct {
    double lambda, dlambda, lambda_factor, lambda_max, lambda_min;   /* 1, 1, 1.6, 1e10, 1e-6 */
    double tol_fun, tol_grad;                                        /* 1e-7, 1e-4            */
    int    max_iter;                                                 /* 500                   */
    int    regType;                                                  /* 1                     */
    double reduce_ratio_min;                                         /* 0                     */
    int    n_alpha;                                                  /* 11                    */
    double alpha[16];                                                /* 10^linspace(0,-3,11)  */
} ddp_ilqg_opts;
void ddp_ilqg_default_opts(ddp_ilqg_opts *o);

enum {
    DDP_EXIT_RUNNING       = 0,
    DDP_EXIT_GRAD          = 1,   /* SUCCESS: gradient norm < tol_grad   (iLQG.jl:258-261) */
    DDP_EXIT_COST          = 2,   /* SUCCESS: cost change < tol_fun      (iLQG.jl:306-309) */
    DDP_EXIT_LAMBDA        = 3,   /* EXIT: lambda > lambda_max           (iLQG.jl:319-322) */
    DDP_EXIT_MAXITER       = 4,   /* while condition exhausted           (iLQG.jl:222)     */
    DDP_EXIT_CAP           = 5,   /* NOT a state of the reference: the driver's own bound on batch-level iterations
                                     (4 max_iter + 1000) ran out while this trajectory was still running         */
    DDP_EXIT_INIT_DIVERGED = -1   /* initial control sequence diverged   (iLQG.jl:205-210) */
};

/* per-trajectory summary, one row per trajectory: stats[8,B] =
 *   [status, iter, accepted_iter, n_backpass, n_forward, lambda, g_norm, sum(cost)]               */
#define DDP_ILQG_NSTATS 8
/* host-pointer flavour. inputs x0[n,B], u0[m,N,B], lims[m,2] or NULL.
 * outputs x[n,N,B] u[m,N,B] K[m,n,N,B] k[m,N,B] (quirk: after an accepted step L.k is the control
 * sequence, iLQG.jl:303) Quu[m,m,N,B] Vx[n,N,B] Vxx[n,n,N,B] cost[CL,B] stats[8,B];
 * trace_cost (may be NULL) [trace_cap,B]: sum(cost) after every iteration (trace(:cost,...)).
 * `global_iters` (may be NULL) receives the number of batch-level iterations executed.
 * x0 / u0 must not alias x / u (the outputs are cleared before the initial rollout).                */
int ddp_ilqg_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o,
                 const double *x0, const double *u0, const double *lims,
                 double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                 double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters);
int ddp_ilqg_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o,
                     const double *x0, const double *u0, const double *lims,
                     double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                     double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters);

/* Warm start from a PRE-ROLLED trajectory (src/iLQG.jl:193-197: `size(x0,2) == N` ⇒ `x = x0`, no initial rollout, no
 * divergence test; `cost` as given or `costfun(x,u)`) — the entry an MPC loop calls with its shifted previous solution.
 * x0[n,N,B]; cost0[CL,B] or NULL.  Line-search rollouts start from x0[:,1] like the reference (iLQG.jl:268).          */
int ddp_ilqg_warm_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o,
                      const double *x0, const double *u0, const double *cost0, const double *lims,
                      double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                      double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters);
int ddp_ilqg_warm_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o,
                          const double *x0, const double *u0, const double *cost0, const double *lims,
                          double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                          double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters);
/* ---- slot scheduler: more problems than resident trajectories, and closed-loop MPC on the device -------------------------------
 * The batch of ddp_ilqg_* advances in lock step: a trajectory that has ended keeps its place until the slowest one ends (pendcart:
 * median 50 iterations, slowest 233).  Here `slots` trajectories are resident and the P = p->B problems go through them: a slot
 * whose solve has ended is flushed to its problem's rows of the outputs and armed with the next problem ON THE DEVICE, in the global
 * iteration in which it ended; the host only polls the number of busy slots.  Every solve performs the launches of its stand-alone
 * solve at batch size `slots` (initial rollout of src/iLQG.jl:181-192 included), so its results are those of ddp_ilqg_f64 with
 * p->B = slots.  inputs x0[n,P] u0[m,N,P]; outputs as ddp_ilqg_f64 with P columns; slots <= 0: min(P, 4096).  Per-trajectory
 * dynamics (dyn_batched) are refused.                                                                                             */
int ddp_ilqg_queue_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int slots,
                       const double *x0, const double *u0, const double *lims,
                       double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                       double *cost, double *stats, int *global_iters);
int ddp_ilqg_queue_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int slots,
                           const double *x0, const double *u0, const double *lims,
                           double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                           double *cost, double *stats, int *global_iters);
/* Closed loop (the receding-horizon use of the warm-start hook, src/iLQG.jl:193-197): every trajectory is solved `steps` times; after
 * solve t its first control is applied (the model is the plant: the next initial state is x_1 of the solution), the control
 * sequence is shifted by one step (tail: last column repeated, or zeros) and solved again — shift and re-solve happen on the
 * device by the same mechanism that re-arms a slot of the queue.  outputs: xcl[n,steps+1,B] (the closed-loop states, xcl[:,0,b] =
 * x0[:,b]), ucl[m,steps,B] (the applied controls), stats_cl[8,steps,B] (the summary row of every solve), x[n,N,B] / u[m,N,B] (the
 * last plan).  A solve whose initial rollout diverges ends the loop of its trajectory (later columns stay zero).                  */
int ddp_ilqg_mpc_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int steps, int zero_tail,
                     const double *x0, const double *u0, const double *lims,
                     double *xcl, double *ucl, double *stats_cl, double *x, double *u, int *global_iters);
int ddp_ilqg_mpc_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int steps, int zero_tail,
                         const double *x0, const double *u0, const double *lims,
                         double *xcl, double *ucl, double *stats_cl, double *x, double *u, int *global_iters);
/* Per-phase GPU time of the following ddp_ilqg_* calls on this handle — the time_derivs / time_backward / time_forward
 * trace keys of the reference (src/iLQG.jl:227,241,281; print_timing :343-366): host_buf[3, cap] (row-major: row r at
 * host_buf + cap*r), seconds per GLOBAL iteration (the batch advances in lock step; the call's *global_iters says how many
 * columns were written), measured with HIP events on the handle's stream.  NULL switches it off. */
int ddp_ilqg_set_timing(ddp_handle h, double *host_buf, int cap);
/* The general entry: optional pre-rolled x0 (x0_prerolled != 0: x0[n,N,B] and cost0[CL,B] or NULL) and ALL per-iteration
 * trace keys of the reference (src/iLQG.jl:257,325-330) per trajectory: trace7[7, trace_cap, B] (may be NULL), rows
 * λ, dλ, α (NaN when no step was accepted), improvement (Δcost), cost (sum), reduce_ratio, grad_norm; entry `iter-1` of
 * trajectory b is written when its iteration `iter` completes (exits by tolerance leave before the trace like upstream). */
int ddp_ilqg_ex_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o,
                    const double *x0, int x0_prerolled, const double *u0, const double *cost0, const double *lims,
                    double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                    double *cost, double *stats, int trace_cap, double *trace7, int *global_iters);
int ddp_ilqg_ex_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o,
                        const double *x0, int x0_prerolled, const double *u0, const double *cost0, const double *lims,
                        double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                        double *cost, double *stats, int trace_cap, double *trace7, int *global_iters);
/* batch-level statistics of one pass in one launch: out4 = [sum(csum[B]), sum(dV[1,:]), sum(dV[2,:]), #(diverge != 0)];
 * any input may be NULL.  This is the vector a multi-GPU job all-reduces (one small collective per pass).               */
int ddp_batch_stats_f64_dev(ddp_handle h, int B, const double *csum, const double *dV, const int32_t *diverge, double *out4);
/* ---- multi-GPU: the one collective of the path -------------------------------------------------------
 * New (the reference is single-process).  One process per GPU, trajectories sharded over the ranks, no data-path exchange; a job
 * shares one small statistics vector per pass / solve.  RCCL is loaded at first use (no link-time dependency).
 *   rank 0: ddp_comm_unique_id(id); ship the 128 bytes to the other ranks by any means (file, MPI, torch.distributed broadcast)
 *   every rank: ddp_comm_create(h, nranks, rank, id, &comm)                       (= ncclCommInitRank, collective)
 *   ddp_allreduce_stats_f64_dev(h, comm, buf, nsum, nmax): device vector buf[nsum + nmax] (<= DDP_COMM_MAX_STATS) becomes, on every
 *   rank, the SUM over ranks of its first nsum entries and the MAX over ranks of the remaining nmax — ONE RCCL call (all-gather of
 *   the tiny vectors + a local reduction), asynchronous on the handle's stream.                                         */
#define DDP_COMM_ID_BYTES 128
#define DDP_COMM_MAX_STATS 64
typedef struct ddp_comm_s *ddp_comm;
/* RCCL as the library sees it: *version = NCCL_VERSION_CODE of the librccl in use (the hand-declared ABI needs >= 2.18: return < 0 below
 * that or when no librccl can be loaded), *preloaded = 1 when the process already held a librccl (e.g. PyTorch's torch/lib/librccl.so) and
 * that instance is used — a second copy is never loaded beside it.                                                            */
int ddp_comm_rccl_info(int *version, int *preloaded);
int ddp_comm_unique_id(char id[DDP_COMM_ID_BYTES]);
int ddp_comm_create(ddp_handle h, int nranks, int rank, const char id[DDP_COMM_ID_BYTES], ddp_comm *out);
int ddp_comm_destroy(ddp_comm c);
int ddp_allreduce_stats_f64_dev(ddp_handle h, ddp_comm c, double *buf, int nsum, int nmax);
/* Receding-horizon warm start between two MPC solves (new, the reference has no MPC loop — its hook is
 * the pre-rolled `x0[n,N]` + `cost` of src/iLQG.jl:193-197, see ddp_ilqg_warm_f64): a time-major array a[d, N, B] moves
 * `shift` steps towards the present, dst[:, i, b] = src[:, i+shift, b]; the vacated tail repeats the last column
 * (zero_tail = 0: controls, nominal states) or is zero (zero_tail = 1: gains).  Out of place, device pointers.        */
int ddp_mpc_shift_f64_dev(ddp_handle h, int d, int N, int B, int shift, int zero_tail, const double *src, double *dst);
/* the `costfun` closure of the registered families on given trajectories: cost[CL,B], csum[B] (may be NULL)          */
int ddp_costfun_f64_dev(ddp_handle h, const ddp_problem *p, const double *x, const double *u, const int32_t *active,
                        double *cost, double *csum);

/* ---- KL-constrained path (BASELINE config 5) ------------------------------------------------------
 * reference: back_pass_gps src/backward_pass.jl:259-350 (called from src/iLQGkl.jl:100,191), ∇kl and kl_div_wiki
 * src/klutils.jl:8-23,70-103, forward_covariance src/forward_pass.jl:37-56, calc_η src/klutils.jl:112-133 (ddp_kl_dual_*).
 * `df(model,·)` / `covariance(model,·)` belong to the un-vendored dependency LinearTimeVaryingModelsBase: the model is
 * passed as the arrays it would return (fx, R1).                                                                 */
typedef struct {
    const double *cx, *cu;       /* cxkl[n,N,B], cukl[m,N,B]                                                     */
    const double *cxx, *cxu;     /* cxxkl[n,n,N,B], cxukl[m,n,N,B]  (m x n, the layout ∇kl returns, klutils.jl:20) */
    const double *cuu;           /* cuukl[m,m,N,B]                                                               */
    const double *eta;           /* eta_tv == 0: η[B] (ηbracket[2] per trajectory); 1: η[N,B] (ηbracket[2,i])     */
    int eta_tv;
} ddp_kl_cost_terms;

/* ∇kl(traj_prev): K[m,n,N,B], k[m,N,B], Sigmai[m,m,N,B] -> the five arrays of ddp_kl_cost_terms (outputs)      */
int ddp_kl_terms_f64_dev(ddp_handle h, int n, int m, int N, int B, const double *K, const double *k, const double *Sigmai,
                         double *cx, double *cu, double *cxx, double *cxu, double *cuu);
int ddp_kl_terms_f64(ddp_handle h, int n, int m, int N, int B, const double *K, const double *k, const double *Sigmai,
                     double *cx, double *cu, double *cxx, double *cxu, double *cuu);

/* back_pass_gps: d->fx_tv and d->cost_tv must be 1 (3-D arrays, as the reference's method signature), regType unused.
 * Outputs as ddp_back_pass plus Quui[m,m,N,B] = inv(Quu_i) (the Σ field of the returned GaussianPolicy); Quu is the
 * KL-augmented, symmetrised matrix (the Σi field).  n <= 32, m <= 8.                                              */
int ddp_back_pass_gps_f64_dev(ddp_handle h, const ddp_bp_desc *d,
                              const double *cx, const double *cu, const double *cxx, const double *cxu, const double *cuu,
                              const double *fx, const double *fu, const ddp_kl_cost_terms *kl,
                              const double *lims, const double *u, const int32_t *active,
                              double *K, double *k, double *Quu, double *Quui, double *Vx, double *Vxx, double *dV,
                              int32_t *diverge);
int ddp_back_pass_gps_f64(ddp_handle h, const ddp_bp_desc *d,
                          const double *cx, const double *cu, const double *cxx, const double *cxu, const double *cuu,
                          const double *fx, const double *fu, const ddp_kl_cost_terms *kl,
                          const double *lims, const double *u,
                          double *K, double *k, double *Quu, double *Quui, double *Vx, double *Vxx, double *dV,
                          int32_t *diverge);

/* forward_covariance: fx[n,n,N] (fx_batched: [n,n,N,B]) and R1[n,n] (shared) of the model, K[m,n,N,B], Sigma[m,m,N,B]
 * -> sigmanew[(n+m),(n+m),N,B]; entries the reference leaves undef (u-blocks of the last step) are zero.           */
int ddp_forward_covariance_f64_dev(ddp_handle h, int n, int m, int N, int B, const double *fx, int fx_batched,
                                   const double *R1, const double *K, const double *Sigma, double *sigmanew);
int ddp_forward_covariance_f64(ddp_handle h, int n, int m, int N, int B, const double *fx, int fx_batched,
                               const double *R1, const double *K, const double *Sigma, double *sigmanew);

/* kl_div_wiki: per-step divergence kldiv[N,B] (clipped at 0) and its mean over time klmean[B]; a trajectory whose
 * logdet would throw (non-positive determinant of Σ) gets klmean = +Inf like the reference's `return Inf`.          */
int ddp_kl_div_f64_dev(ddp_handle h, int n, int m, int N, int B, const double *xnew, const double *xold,
                       const double *sigmanew, const double *Kn, const double *kn, const double *Sn,
                       const double *Kp, const double *kp, const double *Sp, const double *Sip,
                       double *kldiv, double *klmean);
int ddp_kl_div_f64(ddp_handle h, int n, int m, int N, int B, const double *xnew, const double *xold,
                   const double *sigmanew, const double *Kn, const double *kn, const double *Sn,
                   const double *Kp, const double *kp, const double *Sp, const double *Sip,
                   double *kldiv, double *klmean);

/* The dual variable η of the KL constraint, per trajectory, on the device: calc_η (src/klutils.jl:112-133, scalar kl_step) and the
 * bracket / retry / exit logic of the iLQGkl loop (src/iLQGkl.jl:91-122,141,169-177).  All arrays are device pointers owned by
 * the caller; `eta` is what ddp_kl_cost_terms.eta points at.  A host loop is
 *     begin(it) -> n_live;  do { back_pass_gps;  retry(diverge) -> n_pending } while (n_pending);
 *     forward_pass; forward_covariance; kl_div;  update(klmean) -> n_live
 * and only those counts cross PCIe.  A trajectory that has left (status 1/2) keeps its `eta`, so recomputing it with the
 * others reproduces its results.                                                                                      */
typedef struct {
    double  *etab;        /* [3,B]  ηbracket of every trajectory (in/out)                                              */
    double  *eta;         /* [B]    η of the next / last back pass                                                     */
    double  *del;         /* [B]    del (iLQGkl.jl:54,103-105), initialised to del0                                    */
    double  *divergence;  /* [B]    mean KL divergence at the last update                                              */
    int32_t *satisfied;   /* [B]                                                                                       */
    int32_t *status;      /* [B]    0 running, 1 constraint satisfied (:169), 2 η > 0.999 ηmax (:174)                  */
    int32_t *live;        /* [B]    1 while the trajectory iterates (initialise to 1)                                  */
    int32_t *pend;        /* [B]    scratch: still needs a back pass in this iteration                                 */
    int32_t *iters;       /* [B]    last iteration the trajectory took part in                                         */
    int32_t *nback;       /* [B]    back passes spent                                                                  */
} ddp_kl_dual;
int ddp_kl_dual_begin_f64_dev(ddp_handle h, int B, int it, const ddp_kl_dual *s, int *n_live);
int ddp_kl_dual_retry_f64_dev(ddp_handle h, int B, const ddp_kl_dual *s, const int32_t *diverge, int *n_pending);
int ddp_kl_dual_update_f64_dev(ddp_handle h, int B, double kl_step, const ddp_kl_dual *s, const double *klmean, int *n_live);

/* ---- iLQGkl — replaces iLQGkl(dynamics,costfun,derivs,x0,traj_prev,model; kl_step,lims,max_iter,cost,ηbracket,del0) ------------
 * reference: src/iLQGkl.jl:25-178,234-252 (single KL constraint; the per-time-step branch :180-232 cannot run upstream), called from
 * src/demo_linear.jl:124.  The whole loop — derivs, ∇kl, back_pass_gps until the KL-regularised Quu is positive definite (η += del;
 * del *= 2), forward_pass(α = 1), forward_covariance, kl_div_wiki, calc_η, the exit tests — runs on device-resident arrays with one η
 * bracket per trajectory; the registered problem `p` stands in for the three closures, the model is the arrays `df(model,·)` /
 * `covariance(model,·)` would return.
 * inputs : x0[n,N,B] pre-rolled trajectory (:66-73), cost0[B] = sum(cost) of it (NULL: costfun(x0,u); the reference insists on `cost`),
 *          traj_prev = Kp[m,n,N,B], kp[m,N,B] (the previous controls u, :45), Sp / Sip[m,m,N,B]; model_fx[n,n,N] or [n,n,N,B], R1[n,n];
 *          lims[m,2] or NULL; etab[3,B] in/out (NULL: o->etabracket for every trajectory)
 * outputs: x[n,N,B] u[m,N,B] (traj_new.k = copy(u), :239) K[m,n,N,B] Sigma = inv(Quu) and Sigmai = Quu [m,m,N,B] Vx Vxx cost[CL,B] dV[2,B]
 *          stats[DDP_ILQGKL_NSTATS,B] = [status (1 SUCCESS :169, 2 η > ηmax :174, 3 max_iter :234), iter, n_backpass, satisfied,
 *          ηmin, η, ηmax, divergence, sum(cost), Δcost (:135), expected_reduction (:136), grad_norm (:125)]; *iters = batch-level iterations */
typedef struct {
    double kl_step;        /* 1                */
    int    max_iter;       /* 50               */
    double etabracket[3];  /* 1e-8, 1, 1e16    */
    double del0;           /* 1e-4             */
} ddp_ilqgkl_opts;
void ddp_ilqgkl_default_opts(ddp_ilqgkl_opts *o);
#define DDP_ILQGKL_NSTATS 12
int ddp_ilqgkl_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqgkl_opts *o, const double *x0, const double *cost0,
                       const double *Kp, const double *kp, const double *Sp, const double *Sip,
                       const double *model_fx, int model_fx_batched, const double *R1, const double *lims, double *etab,
                       double *x, double *u, double *K, double *Sigma, double *Sigmai, double *Vx, double *Vxx, double *cost,
                       double *dV, double *stats, int *iters);
int ddp_ilqgkl_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqgkl_opts *o, const double *x0, const double *cost0,
                   const double *Kp, const double *kp, const double *Sp, const double *Sip,
                   const double *model_fx, int model_fx_batched, const double *R1, const double *lims, double *etab,
                   double *x, double *u, double *K, double *Sigma, double *Sigmai, double *Vx, double *Vxx, double *cost,
                   double *dV, double *stats, int *iters);

#ifdef __cplusplus
}
#endif
#endif
