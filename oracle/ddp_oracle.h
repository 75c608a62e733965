/*
 * ddp_oracle.h — CPU restatement (plain C99, fp64) of the iLQG hot path of
 * baggepinnen/DifferentialDynamicProgramming.jl v0.5.0.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path (the HIP library
 * under differentialdynamicprogramming.jl_amd/csrc, its ctypes host, bench.py's GPU
 * leg) may call into this file.  Allowed users: tests/, __graft_entry__.smoke(),
 * and bench.py's `cpu_baseline` leg.
 *
 * PARITY PINNING: the reference holds NO numeric golden vectors for this path (its only
 * assertions are the three statistical thresholds of test/test_readme.jl:68-70, and
 * Julia is not installed here, so the reference itself cannot be executed).  This
 * restatement is therefore pinned by (i) an independent NumPy restatement
 * (oracle/np_restatement.py) agreeing to ~1e-12, (ii) analytic known-answer tests
 * derived from the cited lines (Riccati recursion, z==1 for LQ problems, boxQP KKT),
 * (iii) the reference's own statistical thresholds re-run with NumPy seeds.
 * With respect to bit-level outputs of the Julia code: "parity unpinned".
 *
 * All arrays are Julia column-major: M[a,b,t] at a + rows*b + rows*cols*t.
 * Time indices returned to the caller (`diverge`) are 1-based like the reference.
 */
#ifndef DDP_ORACLE_H
#define DDP_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- boxQP (src/boxQP.jl:29-188) ------------------------------------------------ */
typedef struct {
    int    maxIter;        /* 100   */
    double minGrad;        /* 1e-8  */
    double minRelImprove;  /* 1e-8  */
    double stepDec;        /* 0.6   */
    double minStep;        /* 1e-22 */
    double Armijo;         /* 0.1   */
} ddp_oracle_qp_opts;

void ddp_oracle_qp_default_opts(ddp_oracle_qp_opts *o);

/* returns `result` (boxQP.jl:172-179; 0 also stands for a swallowed PosDefException,
 * backward_pass.jl:48-52).  Hfree is written as an m x m column-major buffer whose
 * leading nfree x nfree block (ld = m) is the upper Cholesky factor of H[free,free];
 * free_out[i] in {0,1}; *nfree_out = count; *iters_out = final `iter`. */
int ddp_oracle_boxqp(int m, const double *H, const double *g, const double *lower,
                     const double *upper, const double *x0, const ddp_oracle_qp_opts *opts,
                     double *x, double *Hfree, int *free_out, int *nfree_out, int *iters_out);

/* ---- back_pass (src/backward_pass.jl:162-252 + macro :28-79) ---------------------- */
/* fx_tv:   0 -> fx[n,n], fu[n,m] (LTI, :217-252); 1 -> fx[n,n,N], fu[n,m,N]
 * cost_tv: 0 -> cxx[n,n], cxu[n,m], cuu[m,m];      1 -> [..,N] (:179-215; needs fx_tv=1
 *               in the reference's dispatch, the oracle accepts any combination)
 * lims: NULL (== `[]`) or [m,2]; u[m,N] only read when lims are active.
 * Outputs K[m,n,N], k[m,N], Vx[n,N], Vxx[n,n,N] are zero-filled then written;
 * Quu[m,m,N] is zero-filled where the reference leaves `undef` memory.
 * Returns diverge (0 ok, else 1-based failing time index). */
int ddp_oracle_back_pass(int n, int m, int N,
                         const double *cx, const double *cu,
                         const double *cxx, const double *cxu, const double *cuu,
                         const double *fx, const double *fu,
                         int fx_tv, int cost_tv,
                         double lambda, int regType,
                         const double *lims, const double *u,
                         double *K, double *k, double *Quu,
                         double *Vx, double *Vxx, double *dV);

/* ---- problem families (the closures f / costfun / df of the demos) ---------------- */
enum { DDP_ORACLE_LQ = 0, DDP_ORACLE_PENDCART = 1 };

typedef struct {
    int kind;
    int n, m, N;
    /* LQ  (src/demo_linear.jl:30-50): x+ = A x + B u, cost = .5 sum x.*(Qx) + .5 sum u.*(Ru) */
    const double *A;   /* [n,n] or [n,n,N] */
    const double *Bm;  /* [n,m] or [n,m,N] */
    int dyn_tv;
    const double *Q;   /* [n,n] */
    const double *R;   /* [m,m] */
    /* pendcart (src/system_pendcart.jl:51-54,83-106,137-154); n=4, m=1; Q,R as above */
    double g, l, h, d;
    double goal[4];
    /* the `diff` argument of forward_pass (src/forward_pass.jl:9,19; iLQG.jl:160 `diff_fun = -`): 0 = `-`; bit j set = coordinate j of
     * the difference wrapped to [-pi, pi] (what a caller with an angle state passes: rem2pi(a[j] - b[j], RoundNearest)) */
    unsigned diff_wrap;
} ddp_oracle_problem;

/* number of entries of the cost vector returned by costfun: LQ -> N (per-step split of the
 * reference's scalar, sum identical), pendcart -> N+1 (system_pendcart.jl:97-106) */
int ddp_oracle_cost_len(const ddp_oracle_problem *p);

/* one dynamics step x+ = f(x,u,i) (i 0-based). NaN controls are zeroed in place like the demos. */
void ddp_oracle_f(const ddp_oracle_problem *p, const double *x, double *u, int i, double *xnext);
void ddp_oracle_costfun(const ddp_oracle_problem *p, const double *X, const double *U, double *c);

/* derivatives along a trajectory (the `df` closure).  Writes cx[n,N], cu[m,N]; for pendcart also
 * fx[4,4,N], fu[4,1,N] (ZoH via exp of a 5x5 block matrix).  For LQ fx/fu are not written (they
 * alias A/B).  NaN controls are zeroed in place like the demos. */
void ddp_oracle_df(const ddp_oracle_problem *p, const double *X, double *U,
                   double *cx, double *cu, double *fx, double *fu);

/* dense matrix exponential (Higham 2005 scaling & squaring, what Julia's exp(::Matrix) uses) */
void ddp_oracle_expm(int n, const double *A, double *E);

/* ---- forward_pass (src/forward_pass.jl:9-33) ---------------------------------------- */
/* K,k NULL -> empty policy.  x may be NULL when the policy is empty.  lims NULL or [m,2].
 * Writes xnew[n,N], unew[m,N], cnew[cost_len]. */
void ddp_oracle_forward_pass(const ddp_oracle_problem *p,
                             const double *K, const double *k,
                             const double *x0, const double *u, const double *x,
                             double alpha, const double *lims,
                             double *xnew, double *unew, double *cnew);

/* ---- iLQG outer loop (src/iLQG.jl:143-341) ------------------------------------------- */
typedef struct {
    double lambda, dlambda, lambda_factor, lambda_max, lambda_min;
    double tol_fun, tol_grad;
    int    max_iter;
    int    regType;
    double reduce_ratio_min;
    int    n_alpha;
    const double *alpha;
} ddp_oracle_ilqg_opts;

void ddp_oracle_ilqg_default_opts(ddp_oracle_ilqg_opts *o);   /* iLQG.jl:143-163 */

enum {
    DDP_EXIT_RUNNING       = 0,
    DDP_EXIT_GRAD          = 1,   /* SUCCESS: gradient norm < tol_grad   (iLQG.jl:258-261) */
    DDP_EXIT_COST          = 2,   /* SUCCESS: cost change < tol_fun      (iLQG.jl:306-309) */
    DDP_EXIT_LAMBDA        = 3,   /* EXIT: lambda > lambda_max           (iLQG.jl:319-322) */
    DDP_EXIT_MAXITER       = 4,   /* while condition exhausted           (iLQG.jl:222)     */
    DDP_EXIT_INIT_DIVERGED = -1   /* initial control sequence diverged   (iLQG.jl:205-210) */
};

typedef struct {
    int    status;        /* one of DDP_EXIT_* */
    int    iter;          /* value of `iter` when the loop was left */
    int    accepted_iter;
    int    n_backpass;    /* number of back_pass calls (incl. retries) */
    int    n_forward;     /* number of line-search forward_pass calls */
    double lambda, dlambda, g_norm;
    double dV[2];
    int    trace_len;     /* entries written to the trace arrays below (if non-NULL) */
} ddp_oracle_ilqg_result;

/* x0[n]; u0[m,N]; outputs x[n,N], u[m,N], K[m,n,N], k[m,N] (quirk Q3: the control sequence
 * after an accepted step), Quu[m,m,N], Vx[n,N], Vxx[n,n,N], cost[cost_len].
 * Optional trace arrays of capacity trace_cap (may be NULL): total cost, lambda, alpha
 * (NaN for "NO STEP"), g_norm per iteration. */
int ddp_oracle_ilqg(const ddp_oracle_problem *p, const ddp_oracle_ilqg_opts *o,
                    const double *x0, const double *u0, const double *lims,
                    double *x, double *u, double *K, double *k, double *Quu,
                    double *Vx, double *Vxx, double *cost,
                    ddp_oracle_ilqg_result *res,
                    int trace_cap, double *tr_cost, double *tr_lambda, double *tr_alpha,
                    double *tr_gnorm);

/* all seven per-iteration trace keys (iLQG.jl:257,325-330): tr7[7,trace_cap] rows λ, dλ, α (NaN: no step), improvement, cost,
 * reduce_ratio, grad_norm */
int ddp_oracle_ilqg_trace7(const ddp_oracle_problem *p, const ddp_oracle_ilqg_opts *o,
                           const double *x0, const double *u0, const double *lims,
                           double *x, double *u, double *K, double *k, double *Quu,
                           double *Vx, double *Vxx, double *cost, ddp_oracle_ilqg_result *res, int trace_cap, double *tr7);

/* pre-rolled initial trajectory x0[n,N] (iLQG.jl:193-197): no initial rollout; cost0[cost_len] or NULL (= costfun(x0,u0)) */
int ddp_oracle_ilqg_prerolled(const ddp_oracle_problem *p, const ddp_oracle_ilqg_opts *o,
                              const double *x0, const double *u0, const double *cost0, const double *lims,
                              double *x, double *u, double *K, double *k, double *Quu,
                              double *Vx, double *Vxx, double *cost, ddp_oracle_ilqg_result *res);

/* Batch helpers used by bench.py's cpu_baseline leg: loop ddp_oracle_back_pass +
 * ddp_oracle_forward_pass over B trajectories (batch slowest), single thread. Returns
 * number of diverged trajectories. */
int ddp_oracle_pass_batch_lq(const ddp_oracle_problem *p, int B,
                             const double *cx, const double *cu, const double *cxx,
                             const double *cxu, const double *cuu, double lambda, int regType,
                             const double *x0, const double *u, const double *x, double alpha,
                             double *K, double *k, double *Quu, double *Vx, double *Vxx,
                             double *dV, double *xnew, double *unew, double *cnew);

int ddp_oracle_pass_batch_lq_rep(const ddp_oracle_problem *p, int B, int nrep,
                                 const double *cx, const double *cu, const double *cxx,
                                 const double *cxu, const double *cuu, double lambda, int regType,
                                 const double *x0, const double *u, const double *x, double alpha,
                                 double *K, double *k, double *Quu, double *Vx, double *Vxx,
                                 double *dV, double *xnew, double *unew, double *cnew);

/* ---- KL-constrained path (ddp_oracle_kl.c): src/backward_pass.jl:259-350, src/klutils.jl, src/forward_pass.jl:37-56,
 *      src/iLQGkl.jl (single-constraint branch).  PARITY UNPINNED — see the header of ddp_oracle_kl.c. ------------- */
void ddp_oracle_kl_terms(int n, int m, int T, const double *K, const double *k, const double *Sigmai,
                         double *cx, double *cu, double *cxx, double *cxu /* [m,n,T] */, double *cuu);
int ddp_oracle_back_pass_gps(int n, int m, int N,
                             const double *cx, const double *cu, const double *cxx, const double *cxu, const double *cuu,
                             const double *fx, const double *fu, const double *lims, const double *u,
                             const double *cxkl, const double *cukl, const double *cxxkl, const double *cxukl,
                             const double *cuukl, const double *eta, int eta_tv,
                             double *K, double *k, double *Quu, double *Quui, double *Vx, double *Vxx, double *dV);
void ddp_oracle_forward_covariance(int n, int m, int N, const double *fx, const double *R1,
                                   const double *K, const double *Sigma, double *sigmanew /* [(n+m),(n+m),N] */);
void ddp_oracle_model_covariance(int n, int m, int N, const double *fx, const double *fu, const double *x,
                                 const double *u, double *R1);
int ddp_oracle_kl_div_wiki(int n, int m, int T, const double *xnew, const double *xold, const double *sigmanew,
                           const double *Kn, const double *kn, const double *Sn,
                           const double *Kp, const double *kp, const double *Sp, const double *Sip, double *kldiv);
int ddp_oracle_calc_eta(double *etabracket3, double divergence_mean, double kl_step);

typedef struct {
    int    status;        /* 1 SUCCESS |KL - kl_step| < 0.1 kl_step (iLQGkl.jl:169), 2 EXIT eta > eta_max (:174), 3 max_iter (:234) */
    int    iter, n_backpass, satisfied;
    double eta[3], divergence, g_norm, dV[2], cost0;
} ddp_oracle_ilqgkl_result;

int ddp_oracle_ilqgkl(const ddp_oracle_problem *p, const double *x0 /* [n,N] pre-rolled */, double cost0,
                      const double *Kp, const double *kp, const double *Sp, const double *Sip,
                      const double *model_fx /* [n,n,N] */, const double *R1, const double *lims,
                      double kl_step, int max_iter, const double *etabracket3, double del0,
                      double *x, double *u, double *K, double *k, double *Quu, double *Quui,
                      double *Vx, double *Vxx, double *cost, ddp_oracle_ilqgkl_result *res);

#ifdef __cplusplus
}
#endif
#endif
