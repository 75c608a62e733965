// ilqg.hip — device-resident iLQG iteration for registered problem families.
//
// Replaces the outer loop of  iLQG(f,costfun,df,x0,u0; ...)  (src/iLQG.jl:143-341).  Every trajectory
// of the batch is an independent solve with its own scalar state machine (λ, dλ, iter, accepted_iter,
// status) held in device memory; one "global iteration" runs, for the trajectories that need it,
//   STEP 1  df            (df.hip)                    iLQG.jl:225-229
//   STEP 2  back_pass     (back_pass.hip)             iLQG.jl:235-251   (a diverged trajectory only
//                                                     updates λ here and retries next global iteration)
//           g_norm / gradient exit                    iLQG.jl:254-261
//   STEP 3  all-α line-search rollouts (forward_pass.hip), first α in list order that passes  :264-283
//   STEP 4  accept / reject, λ schedule, termination  iLQG.jl:293-323
// The host only launches kernels and polls one counter (number of running trajectories) per global
// iteration.  The λ-schedule quirks of the reference are kept: the increase uses the OLD dλ
// (tuple assignment, :246,:313), the decrease the NEW one (:299-300), λ never drops below λmin.
#include <stdlib.h>
#include <vector>
#include "ddp_internal.h"

namespace {

struct Traj {           // per-trajectory scalar state (structure of arrays in one allocation)
    double *lam, *dlam, *gnorm, *csum;
    int32_t *status, *iter, *acc, *nbp, *nfp, *flg, *run, *dodf, *dofwd, *div0;
};

struct Opt {
    double lfac, lmax, lmin, tol_fun, tol_grad, rrmin;
    int max_iter, nalpha;
    double alpha[16];
};

__global__ void init_state_kernel(int B, double lam0, double dlam0, Traj s)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    s.lam[b] = lam0; s.dlam[b] = dlam0; s.gnorm[b] = 0.0; s.csum[b] = 0.0;
    s.status[b] = DDP_EXIT_RUNNING; s.iter[b] = 1; s.acc[b] = 1; s.nbp[b] = 0; s.nfp[b] = 0;
    s.flg[b] = 1; s.run[b] = 1; s.dodf[b] = 1; s.dofwd[b] = 0; s.div0[b] = 1;
}

// u_scaled = α·u0 for the initial rollout (iLQG.jl:185)
__global__ void scale_kernel(size_t per, int B, double a, const double *u0, const int32_t *act, double *out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per * B) return;
    if (act[t / per]) out[t] = a * u0[t];
}

// initial rollout check: all(abs.(x) .< 1e8)  (iLQG.jl:187) — one wave per trajectory
__global__ __launch_bounds__(64) void init_check_kernel(int n, int m, int N, int CL, const double *xc, const double *uc,
                                                        const double *cc, const double *csum, Traj s, double *x, double *u,
                                                        double *cost)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (!s.div0[b]) return;
    const double *xb = xc + (size_t)n * N * b;
    int bad = 0;
    for (size_t e = lane; e < (size_t)n * N; e += 64) bad |= !(fabs(xb[e]) < 1e8);
    bad = __any(bad);
    if (bad) return;
    for (size_t e = lane; e < (size_t)n * N; e += 64) x[(size_t)n * N * b + e] = xb[e];
    for (size_t e = lane; e < (size_t)m * N; e += 64) u[(size_t)m * N * b + e] = uc[(size_t)m * N * b + e];
    for (size_t e = lane; e < (size_t)CL; e += 64) cost[(size_t)CL * b + e] = cc[(size_t)CL * b + e];
    if (lane == 0) { s.div0[b] = 0; s.csum[b] = csum[b]; }
}

// costfun of the registered families on a given trajectory: one wave per trajectory, lanes over time
//   LQ        c_i = .5 x_i'Q x_i + .5 u_i'R u_i                       (src/demo_linear.jl:49, split per step)
//   pendcart  c_i = .5 ((x_i-goal)'Q(x_i-goal) + R u_i^2),  c_{N+1} = .5 (x_N-goal)'Q(x_N-goal)   (src/system_pendcart.jl:97-106)
__global__ __launch_bounds__(64) void costfun_kernel(int kind, int n, int m, int N, int CL, const double *Q, const double *R,
                                                     double g0, double g1, double g2, double g3, const double *x, const double *u,
                                                     const int32_t *active, double *cost, double *csum)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (active && active[b] == 0) return;
    const double *xb = x + (size_t)n * N * b, *ub = u + (size_t)m * N * b;
    const double goal[4] = {g0, g1, g2, g3};
    const bool pend = kind == DDP_PROBLEM_PENDCART;
    double acc = 0.0;
    for (int t = lane; t < CL; t += 64) {
        const int tx = t < N ? t : N - 1;
        double qx = 0.0, ru = 0.0;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += Q[i + n * j] * (xb[(size_t)n * tx + j] - (pend ? goal[j & 3] : 0.0));
            qx += (xb[(size_t)n * tx + i] - (pend ? goal[i & 3] : 0.0)) * s;
        }
        if (t < N)
            for (int i = 0; i < m; ++i) {
                double s = 0.0;
                for (int j = 0; j < m; ++j) s += R[i + m * j] * ub[(size_t)m * t + j];
                ru += ub[(size_t)m * t + i] * s;
            }
        const double c = pend ? 0.5 * (qx + ru) : 0.5 * qx + 0.5 * ru;
        cost[(size_t)CL * b + t] = c;
        acc += c;
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0 && csum) csum[b] = acc;
}

// pre-rolled initial trajectory (iLQG.jl:193-197): x = x0[n,N], u = u0, cost given or costfun(x,u); x0c <- x0[:,1]
__global__ __launch_bounds__(64) void preroll_init_kernel(int n, int m, int N, int CL, const double *x0, const double *u0,
                                                          const double *cost0, Traj s, double *x, double *u, double *cost, double *x0c)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    for (size_t e = lane; e < (size_t)n * N; e += 64) x[(size_t)n * N * b + e] = x0[(size_t)n * N * b + e];
    for (size_t e = lane; e < (size_t)m * N; e += 64) u[(size_t)m * N * b + e] = u0[(size_t)m * N * b + e];
    for (int e = lane; e < n; e += 64) x0c[(size_t)n * b + e] = x0[(size_t)n * N * b + e];
    if (cost0) {
        double acc = 0.0;
        for (int e = lane; e < CL; e += 64) { const double c = cost0[(size_t)CL * b + e]; cost[(size_t)CL * b + e] = c; acc += c; }
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) s.csum[b] = acc;
    }
    if (lane == 0) s.div0[b] = 0;
}

__global__ void count_ok_kernel(int B, Traj s, int *counter)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && !s.div0[b]) atomicAdd(counter, 1);
}

__global__ void init_finish_kernel(int B, Traj s, int *counter)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (s.div0[b]) { s.status[b] = DDP_EXIT_INIT_DIVERGED; s.run[b] = 0; s.dodf[b] = 0; }   // iLQG.jl:205-210
    else atomicAdd(counter, 1);
}

// after back_pass: λ update on divergence, g_norm, gradient exit  (iLQG.jl:244-261) — one wave per trajectory
__global__ __launch_bounds__(64) void post_bp_kernel(int m, int N, Opt o, const int32_t *diverge, const double *k,
                                                     const double *u, Traj s)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (!s.run[b]) return;
    // g_norm = mean(maximum(abs.(k) ./ (abs.(u) .+ 1), dims=1))   (:256)
    const double *kb = k + (size_t)m * N * b, *ub = u + (size_t)m * N * b;
    double acc = 0.0;
    for (int t = lane; t < N; t += 64) {
        double mx = 0.0;
        for (int a = 0; a < m; ++a) {
            const double r = fabs(kb[(size_t)m * t + a]) / (fabs(ub[(size_t)m * t + a]) + 1.0);
            if (a == 0 || r > mx || r != r) mx = r;
        }
        acc += mx;
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane != 0) return;
    const double g_norm = acc / N;
    s.gnorm[b] = g_norm;
    s.nbp[b] += 1;
    double lam = s.lam[b], dlam = s.dlam[b];
    int dofwd = 0, status = DDP_EXIT_RUNNING;
    if (diverge[b] > 0) {
        const double dl = dlam;                                       // tuple assignment (:246)
        dlam = fmax(dl * o.lfac, o.lfac);
        lam = fmax(lam * dl, o.lmin);
        if (lam > o.lmax) {
            // inner loop left with back_pass_done == false: no line search, the "no step" branch raises
            // λ once more and terminates (:311-322)
            const double dl2 = dlam;
            dlam = fmax(dl2 * o.lfac, o.lfac);
            lam = fmax(lam * dl2, o.lmin);
            status = DDP_EXIT_LAMBDA;
        }
        // else: retry the backward pass with the larger λ in the next global iteration (`continue`)
    } else {
        if (g_norm < o.tol_grad && lam < 1e-5) status = DDP_EXIT_GRAD;    // :258-261
        else dofwd = 1;
    }
    s.lam[b] = lam; s.dlam[b] = dlam;
    s.dofwd[b] = dofwd;
    s.dodf[b] = 0;
    if (status != DDP_EXIT_RUNNING) { s.status[b] = status; s.run[b] = 0; }
}

// STEP 3 selection + STEP 4 (iLQG.jl:267-331) — one wave per trajectory
__global__ __launch_bounds__(64) void accept_kernel(int n, int m, int N, int B, int CL, Opt o, const double *dV,
                                                    const double *xnew, const double *unew, const double *cnew,
                                                    const double *csumnew, Traj s, double *x, double *u, double *cost,
                                                    double *k, int trace_cap, double *trace_cost, double *trace7, const int32_t *map,
                                                    int *counter)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int tb = map ? map[b] : b;                                   // row of the caller's trace arrays (slot -> trajectory)
    if (!s.run[b]) return;
    if (!s.dofwd[b]) {            // diverged back_pass, retrying: nothing to do this round
        if (lane == 0) atomicAdd(counter, 1);
        return;
    }
    const double c0 = s.csum[b], dV0 = dV[2 * b], dV1 = dV[2 * b + 1];
    int sel = -1;
    double dcost = 0.0, zlast = 0.0;
    for (int ai = 0; ai < o.nalpha; ++ai) {                            // serial order of the reference
        const double a = o.alpha[ai];
        dcost = c0 - csumnew[(size_t)b + (size_t)B * ai];
        const double expected = -a * (dV0 + a * dV1);
        double z;
        if (expected > 0) z = dcost / expected;
        else z = (dcost > 0) ? 1.0 : ((dcost < 0) ? -1.0 : dcost);    // sign(Δcost) (NaN stays NaN)
        zlast = z;
        if (z > o.rrmin) { sel = ai; break; }
    }
    double lam = s.lam[b], dlam = s.dlam[b];
    int status = DDP_EXIT_RUNNING, acc = s.acc[b];
    if (sel >= 0) {                                                    // :293-310
        dlam = fmin(dlam / o.lfac, 1.0 / o.lfac);
        lam = fmax(lam * dlam, o.lmin);
        const size_t src = (size_t)b + (size_t)B * sel;
        for (size_t e = lane; e < (size_t)n * N; e += 64) x[(size_t)n * N * b + e] = xnew[(size_t)n * N * src + e];
        for (size_t e = lane; e < (size_t)m * N; e += 64) {
            const double v = unew[(size_t)m * N * src + e];
            u[(size_t)m * N * b + e] = v;
            k[(size_t)m * N * b + e] = v;                              // traj_new.k = copy(u)  (:303)
        }
        for (size_t e = lane; e < (size_t)CL; e += 64) cost[(size_t)CL * b + e] = cnew[(size_t)CL * src + e];
        if (dcost < o.tol_fun) status = DDP_EXIT_COST;                 // :306-309
        else acc += 1;
    } else {                                                           // :311-323
        const double dl = dlam;
        dlam = fmax(dl * o.lfac, o.lfac);
        lam = fmax(lam * dl, o.lmin);
        if (lam > o.lmax) status = DDP_EXIT_LAMBDA;
    }
    if (lane != 0) return;
    s.nfp[b] += (sel >= 0) ? sel + 1 : o.nalpha;
    s.lam[b] = lam; s.dlam[b] = dlam; s.acc[b] = acc;
    if (sel >= 0) { s.csum[b] = csumnew[(size_t)b + (size_t)B * sel]; s.flg[b] = 1; }
    if (status == DDP_EXIT_RUNNING) {
        const int it = s.iter[b];
        if (trace_cost && it - 1 < trace_cap) trace_cost[(size_t)trace_cap * tb + (it - 1)] = s.csum[b];   // :329
        if (trace7 && it - 1 < trace_cap) {                            // :257,325-330: λ, dλ, α, improvement, cost, reduce_ratio, grad_norm
            double *t7 = trace7 + 7 * ((size_t)trace_cap * tb + (it - 1));
            t7[0] = lam; t7[1] = dlam; t7[2] = sel >= 0 ? o.alpha[sel] : NAN; t7[3] = dcost; t7[4] = s.csum[b]; t7[5] = zlast;
            t7[6] = s.gnorm[b];
        }
        s.iter[b] = it + 1;
        if (acc > o.max_iter) status = DDP_EXIT_MAXITER;               // while accepted_iter <= max_iter (:222)
    }
    if (status != DDP_EXIT_RUNNING) { s.status[b] = status; s.run[b] = 0; s.dodf[b] = 0; }
    else { s.dodf[b] = (sel >= 0) ? 1 : 0; atomicAdd(counter, 1); }
    s.dofwd[b] = 0;
}

// only_finished: the slots that stopped running (their working set is about to be dropped); cap: still running when the driver gave up
// After the first g step sizes of a line search have been rolled out: does the serial search of the reference (iLQG.jl:267-281)
// stop inside them?  more[b] = 1 for the trajectories that still need the later step sizes.
__global__ void ls_more_kernel(int B, int g, Opt o, const double *dV, const double *csumnew, Traj s, const int32_t *prev_more, int32_t *more)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int need = 0;
    if (s.run[b] && s.dofwd[b] && (!prev_more || prev_more[b])) {
        const double c0 = s.csum[b], dV0 = dV[2 * b], dV1 = dV[2 * b + 1];
        need = 1;
        for (int ai = 0; ai < g; ++ai) {
            const double a = o.alpha[ai], dcost = c0 - csumnew[(size_t)b + (size_t)B * ai], expected = -a * (dV0 + a * dV1);
            const double z = expected > 0 ? dcost / expected : ((dcost > 0) ? 1.0 : ((dcost < 0) ? -1.0 : dcost));
            if (z > o.rrmin) { need = 0; break; }
        }
    }
    more[b] = need;
}

// ---- compaction of the live trajectories (the batch advances in lock step: finished trajectories would keep their waves busy)
// idx[0..count) = the running slots in order (one wave, ballot + prefix count)
__global__ __launch_bounds__(64) void live_index_kernel(int B, Traj s, int32_t *idx, int *count)
{
    const int lane = threadIdx.x;
    int base = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        const bool live = b < B && s.run[b] != 0;
        const unsigned long long mask = __ballot(live);
        if (live) idx[base + __popcll(mask & ((1ull << lane) - 1ull))] = b;
        base += __popcll(mask);
    }
    if (lane == 0) *count = base;
}

struct WorkSet {            // where the state of the running trajectories lives: the caller's arrays, or a compacted copy
    double *x, *u, *cost, *K, *k, *Quu, *Vx, *Vxx, *x0;
    Traj s;
    int32_t *map;           // slot -> trajectory of the caller's batch (NULL: identity)
};

// slot i of `dst` <- slot idx[i] of `src` (state only: x, u, cost, x0 and the scalar state machine; gains and value function are
// recomputed by the next back pass, derivatives by the next df: dodf is set)
__global__ __launch_bounds__(64) void gather_kernel(int n, int m, int N, int CL, const int32_t *idx, WorkSet src, WorkSet dst)
{
    const int i = blockIdx.x, lane = threadIdx.x, j = idx[i];
    for (size_t e = lane; e < (size_t)n * N; e += 64) dst.x[(size_t)n * N * i + e] = src.x[(size_t)n * N * j + e];
    for (size_t e = lane; e < (size_t)m * N; e += 64) dst.u[(size_t)m * N * i + e] = src.u[(size_t)m * N * j + e];
    for (int e = lane; e < CL; e += 64) dst.cost[(size_t)CL * i + e] = src.cost[(size_t)CL * j + e];
    for (int e = lane; e < n; e += 64) dst.x0[(size_t)n * i + e] = src.x0[(size_t)n * j + e];
    if (lane == 0) {
        dst.s.lam[i] = src.s.lam[j]; dst.s.dlam[i] = src.s.dlam[j]; dst.s.gnorm[i] = src.s.gnorm[j]; dst.s.csum[i] = src.s.csum[j];
        dst.s.status[i] = src.s.status[j]; dst.s.iter[i] = src.s.iter[j]; dst.s.acc[i] = src.s.acc[j]; dst.s.nbp[i] = src.s.nbp[j];
        dst.s.nfp[i] = src.s.nfp[j]; dst.s.flg[i] = src.s.flg[j]; dst.s.run[i] = src.s.run[j]; dst.s.dodf[i] = 1; dst.s.dofwd[i] = 0;
        dst.s.div0[i] = src.s.div0[j];
        dst.map[i] = src.map ? src.map[j] : j;
    }
}

// results of the slots of a compacted working set that have stopped (all == 0) / of every slot (all != 0) go to the caller's arrays
__global__ __launch_bounds__(64) void scatter_kernel(int n, int m, int N, int CL, int all, WorkSet src, WorkSet dst)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    if (!all && src.s.run[j]) return;
    const size_t t = (size_t)src.map[j];
    auto cp = [&](double *d, const double *s_, size_t per) { for (size_t e = lane; e < per; e += 64) d[per * t + e] = s_[per * j + e]; };
    cp(dst.x, src.x, (size_t)n * N); cp(dst.u, src.u, (size_t)m * N); cp(dst.cost, src.cost, (size_t)CL);
    cp(dst.K, src.K, (size_t)m * n * N); cp(dst.k, src.k, (size_t)m * N); cp(dst.Quu, src.Quu, (size_t)m * m * N);
    cp(dst.Vx, src.Vx, (size_t)n * N); cp(dst.Vxx, src.Vxx, (size_t)n * n * N);
}

__global__ void stats_kernel(int B, Traj s, const int32_t *map, int only_finished, double *stats)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (only_finished && s.run[b]) return;
    if (!only_finished && s.run[b]) s.status[b] = DDP_EXIT_CAP;
    double *r = stats + (size_t)DDP_ILQG_NSTATS * (map ? map[b] : b);
    r[0] = s.status[b]; r[1] = s.iter[b]; r[2] = s.acc[b]; r[3] = s.nbp[b]; r[4] = s.nfp[b];
    r[5] = s.lam[b]; r[6] = s.gnorm[b]; r[7] = s.csum[b];
}


// ---- slot scheduler (ddp_ilqg_queue_f64_dev, ddp_ilqg_mpc_f64_dev): S resident slots work through P >= S problems (queue), or every
// slot re-solves its own problem `steps` times in closed loop (MPC: apply u_0, the model is the plant, shift, solve again).  A slot
// whose solve has ended is flushed and re-armed ON THE DEVICE at the end of the global iteration in which it ended; the host only
// polls the number of busy slots.  The state machine of a solve is the one of ilqg_impl (same kernels, same launches per global
// iteration), so a solve does the arithmetic of its stand-alone solve at the same batch size.
struct Sched {
    int P, mpc_steps, zero_tail, nalpha;
    double alpha[16];
    double lam0, dlam0;
    const double *x0g, *u0g;            // problems: x0[n,P], u0[m,N,P]
    double *u0s, *us, *x0s;             // per slot: the control sequence the solve starts from, its scaled copy α·u0s, the initial state
    int32_t *initm, *ai, *map, *left, *noflush, *ready;
    int *qhead;
    // results per problem (queue) / last plan per trajectory (MPC)
    double *x, *u, *cost, *K, *k, *Quu, *Vx, *Vxx, *stats;
    double *xcl, *ucl, *stats_cl;       // MPC: closed-loop states [n,steps+1,P], controls [m,steps,P], summaries [8,steps,P]
};

// end of a global iteration (and once before the first): flush the slots whose solve has ended, re-arm them — one wave per slot
constexpr int TAKE_T = 256;                                      // threads per slot of sched_take_kernel
__global__ __launch_bounds__(TAKE_T) void sched_take_kernel(int n, int m, int N, int CL, Sched q, WorkSet ws, int *counter)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    Traj &s = ws.s;
    // the slot's state as every thread of the work-group sees it BEFORE anybody changes it
    const int ready = q.ready[b], running = s.run[b], arming = q.initm[b], prob = q.map[b], left0 = q.left[b], nofl = q.noflush[b];
    __syncthreads();
    if (ready) {                                                   // its initial rollout (on the side stream) passed: the solve starts with
        if (lane == 0) { q.ready[b] = 0; q.initm[b] = 0; s.run[b] = 1; s.dodf[b] = 1; atomicAdd(counter, 1); }     // the next global iteration
        return;
    }
    if (running || arming) { if (lane == 0) atomicAdd(counter, 1); return; }
    bool again = false;                                            // MPC: the same trajectory goes on
    if (prob >= 0) {
        const bool valid = !nofl;                                  // an initial divergence leaves nothing to copy (iLQG.jl:205-210)
        const size_t t = (size_t)prob;
        // (a finished solve leaves ~150 KB for n = 4, N = 600: 16-byte pieces where the sizes allow it)
        auto cp = [&](double *d, const double *src, size_t per) {
            double *dd = d + per * t; const double *ss = src + per * (size_t)b;
            if (per % 2 == 0 && (((uintptr_t)dd | (uintptr_t)ss) & 15) == 0)
                for (size_t e = lane; e < per / 2; e += TAKE_T) ((double2 *)dd)[e] = ((const double2 *)ss)[e];
            else
                for (size_t e = lane; e < per; e += TAKE_T) dd[e] = ss[e];
        };
        if (q.mpc_steps == 0) {
            if (valid) {
                cp(q.x, ws.x, (size_t)n * N); cp(q.u, ws.u, (size_t)m * N); cp(q.cost, ws.cost, (size_t)CL);
                cp(q.K, ws.K, (size_t)m * n * N); cp(q.k, ws.k, (size_t)m * N); cp(q.Quu, ws.Quu, (size_t)m * m * N);
                cp(q.Vx, ws.Vx, (size_t)n * N); cp(q.Vxx, ws.Vxx, (size_t)n * n * N);
            }
            if (lane == 0) {
                double *r = q.stats + (size_t)DDP_ILQG_NSTATS * t;
                r[0] = s.status[b]; r[1] = s.iter[b]; r[2] = s.acc[b]; r[3] = s.nbp[b]; r[4] = s.nfp[b]; r[5] = s.lam[b]; r[6] = s.gnorm[b];
                r[7] = s.csum[b];
            }
        } else {
            const int step = q.mpc_steps - left0;                   // 0-based index of the solve that has just ended
            const double *xb = ws.x + (size_t)n * N * b, *ub = ws.u + (size_t)m * N * b;
            if (lane == 0) {
                double *r = q.stats_cl + (size_t)DDP_ILQG_NSTATS * ((size_t)q.mpc_steps * t + step);
                r[0] = s.status[b]; r[1] = s.iter[b]; r[2] = s.acc[b]; r[3] = s.nbp[b]; r[4] = s.nfp[b]; r[5] = s.lam[b]; r[6] = s.gnorm[b];
                r[7] = s.csum[b];
            }
            if (valid) {
                for (int e = lane; e < n; e += TAKE_T) q.xcl[(size_t)n * ((size_t)(q.mpc_steps + 1) * t + step) + e] = xb[e];
                for (int e = lane; e < m; e += TAKE_T) q.ucl[(size_t)m * ((size_t)q.mpc_steps * t + step) + e] = ub[e];
                for (int e = lane; e < n; e += TAKE_T) q.xcl[(size_t)n * ((size_t)(q.mpc_steps + 1) * t + step + 1) + e] = xb[(N > 1 ? n : 0) + e];
            }
            const int left = left0 - 1;
            again = valid && left > 0;                                // (a solve that diverged at its start ends the loop of its trajectory)
            if (again) {
                // the model is the plant: the next solve starts at x_1, from the shifted control sequence
                double *u0b = q.u0s + (size_t)m * N * b;
                for (size_t e = lane; e < (size_t)m * N; e += TAKE_T) {
                    const size_t i = e / m, c = e % m;
                    u0b[e] = (i + 1 < (size_t)N) ? ub[(i + 1) * m + c] : (q.zero_tail ? 0.0 : ub[(size_t)(N - 1) * m + c]);
                }
                for (int e = lane; e < n; e += TAKE_T) q.x0s[(size_t)n * b + e] = xb[(N > 1 ? n : 0) + e];
            } else if (valid) {                                       // the last plan
                cp(q.x, ws.x, (size_t)n * N); cp(q.u, ws.u, (size_t)m * N);
            }
            if (lane == 0) q.left[b] = left;
        }
    }
    if (!again) {
        // queue: the next problem; MPC: a slot takes a problem only at the initial fill (map == -1), afterwards it rests (map == -2)
        __shared__ int head_s;
        if (lane == 0) head_s = (prob != -2 && (q.mpc_steps == 0 || prob == -1)) ? atomicAdd(q.qhead, 1) : q.P;
        __syncthreads();
        const int head = head_s;
        if (head >= q.P) { if (lane == 0) q.map[b] = -2; return; }
        const double *u0p = q.u0g + (size_t)m * N * head;
        double *u0b = q.u0s + (size_t)m * N * b;
        for (size_t e = lane; e < (size_t)m * N; e += TAKE_T) u0b[e] = u0p[e];
        for (int e = lane; e < n; e += TAKE_T) q.x0s[(size_t)n * b + e] = q.x0g[(size_t)n * head + e];
        if (lane == 0) { q.map[b] = head; q.left[b] = q.mpc_steps; }
    }
    // arm the slot: the scalar state of a fresh solve (init_state_kernel), the first candidate of the initial rollout (iLQG.jl:181-192)
    {
        const double *u0b = q.u0s + (size_t)m * N * b;
        double *usb = q.us + (size_t)m * N * b;
        const double a0 = q.alpha[0];
        for (size_t e = lane; e < (size_t)m * N; e += TAKE_T) usb[e] = a0 * u0b[e];
    }
    if (lane == 0) {
        s.lam[b] = q.lam0; s.dlam[b] = q.dlam0; s.gnorm[b] = 0.0; s.csum[b] = 0.0;
        s.status[b] = DDP_EXIT_RUNNING; s.iter[b] = 1; s.acc[b] = 1; s.nbp[b] = 0; s.nfp[b] = 0;
        s.flg[b] = 1; s.run[b] = 0; s.dodf[b] = 0; s.dofwd[b] = 0; s.div0[b] = 1;
        q.initm[b] = 1; q.ai[b] = 0; q.noflush[b] = 0;
        atomicAdd(counter, 1);
    }
}

// after the initial rollout of the armed slots and init_check_kernel: bounded -> the solve starts; else the next α, or the end (:205-210)
__global__ __launch_bounds__(64) void sched_init_advance_kernel(int m, int N, Sched q, Traj s)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (!q.initm[b] || q.ready[b]) return;
    if (!s.div0[b]) {
        // bounded: hand the slot to the main stream's kernels through sched_take_kernel (this kernel runs beside df / back_pass / the
        // line search of the running slots, whose masks must not change under them)
        if (lane == 0) q.ready[b] = 1;
        return;
    }
    const int ai = q.ai[b] + 1;
    if (ai < q.nalpha) {
        const double *u0b = q.u0s + (size_t)m * N * b;
        double *usb = q.us + (size_t)m * N * b;
        const double a = q.alpha[ai];
        for (size_t e = lane; e < (size_t)m * N; e += 64) usb[e] = a * u0b[e];
        if (lane == 0) q.ai[b] = ai;
    } else if (lane == 0) {
        q.initm[b] = 0; q.noflush[b] = 1;                              // (run stays 0: the slot is flushed and re-armed by the next sched_take_kernel)
        s.status[b] = DDP_EXIT_INIT_DIVERGED; s.dodf[b] = 0; s.div0[b] = 0;
    }
}

}   // namespace

extern "C" {

void ddp_ilqg_default_opts(ddp_ilqg_opts *o)
{   // iLQG.jl:143-163
    o->lambda = 1.0; o->dlambda = 1.0; o->lambda_factor = 1.6; o->lambda_max = 1e10; o->lambda_min = 1e-6;
    o->tol_fun = 1e-7; o->tol_grad = 1e-4; o->max_iter = 500; o->regType = 1; o->reduce_ratio_min = 0.0;
    o->n_alpha = 11;
    for (int i = 0; i < 16; ++i) o->alpha[i] = i < 11 ? pow(10.0, -3.0 * i / 10.0) : 0.0;
}

static int ilqg_impl(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *oo, const double *x0, const double *u0,
                     const double *lims, double *x, double *u, double *K, double *k, double *Quu, double *Vx,
                     double *Vxx, double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters,
                     bool prerolled, const double *cost0, double *trace7 = nullptr)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && p && x0 && u0 && x && u && K && k && Quu && Vx && Vxx && cost && stats, "ilqg: null argument");
    ddp_ilqg_opts od;
    if (!oo) { ddp_ilqg_default_opts(&od); oo = &od; }
    DDP_CHECK(oo->n_alpha >= 1 && oo->n_alpha <= 16, "ilqg: n_alpha=%d out of [1,16]", oo->n_alpha);
    DDP_CHECK(x0 != x && u0 != u, "ilqg: x0/u0 must not alias the outputs x/u (the outputs are cleared before the initial rollout)");
    const size_t n = p->n, m = p->m, N = p->N, B = p->B, na = oo->n_alpha, CL = ddp_cost_len(p);
    const bool pend = p->kind == DDP_PROBLEM_PENDCART;

    // ---- workspace (device): derivatives, candidates, scalar state
    size_t bytes = 0;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t s_cx = al(n * N * B * 8), s_cu = al(m * N * B * 8), s_fx = pend ? al(n * n * N * B * 8) : 0,
                 s_fu = pend ? al(n * m * N * B * 8) : 0, s_xn = al(n * N * B * na * 8), s_un = al(m * N * B * na * 8),
                 s_cn = al(CL * B * na * 8), s_cs = al(B * na * 8), s_dV = al(2 * B * 8), s_div = al(B * 4),
                 s_cxu = al(n * m * 8), s_us = al(m * N * B * 8), s_d = al(B * 8), s_i = al(B * 4), s_x0 = al(n * B * 8);
    bytes = s_cx + s_cu + s_fx + s_fu + s_xn + s_un + s_cn + s_cs + s_dV + s_div + s_cxu + s_us + 4 * s_d + 10 * s_i + 256 + s_x0 + 2 * s_i;
    void *base;
    int rc = ddp_scratch(h, bytes, &base);
    if (rc) return rc;
    char *pp = (char *)base;
    auto take = [&](size_t b) { void *r = pp; pp += b; return r; };
    double *cx = (double *)take(s_cx), *cu = (double *)take(s_cu), *fxw = pend ? (double *)take(s_fx) : nullptr,
           *fuw = pend ? (double *)take(s_fu) : nullptr, *xn = (double *)take(s_xn), *un = (double *)take(s_un),
           *cn = (double *)take(s_cn), *cs = (double *)take(s_cs), *dV = (double *)take(s_dV);
    int32_t *div = (int32_t *)take(s_div);
    double *cxu = (double *)take(s_cxu), *us = (double *)take(s_us);
    Traj s;
    s.lam = (double *)take(s_d); s.dlam = (double *)take(s_d); s.gnorm = (double *)take(s_d); s.csum = (double *)take(s_d);
    s.status = (int32_t *)take(s_i); s.iter = (int32_t *)take(s_i); s.acc = (int32_t *)take(s_i); s.nbp = (int32_t *)take(s_i);
    s.nfp = (int32_t *)take(s_i); s.flg = (int32_t *)take(s_i); s.run = (int32_t *)take(s_i); s.dodf = (int32_t *)take(s_i);
    s.dofwd = (int32_t *)take(s_i); s.div0 = (int32_t *)take(s_i);
    int *counter = (int *)take(256);
    double *x0c = (double *)take(s_x0);                  // first column of a pre-rolled x0
    DDP_CHECK(h->h_pinned, "ilqg: pinned poll buffer missing");

    Opt o;
    o.lfac = oo->lambda_factor; o.lmax = oo->lambda_max; o.lmin = oo->lambda_min; o.tol_fun = oo->tol_fun;
    o.tol_grad = oo->tol_grad; o.rrmin = oo->reduce_ratio_min; o.max_iter = oo->max_iter; o.nalpha = (int)na;
    for (int i = 0; i < 16; ++i) o.alpha[i] = oo->alpha[i];

    hipStream_t st = h->stream;
    const unsigned gB = (unsigned)((B + 255) / 256);
    hipLaunchKernelGGL(init_state_kernel, dim3(gB), dim3(256), 0, st, (int)B, oo->lambda, oo->dlambda, s);
    DDP_HIP(hipMemsetAsync(cxu, 0, n * m * 8, st));
    DDP_HIP(hipMemsetAsync(K, 0, m * n * N * B * 8, st));
    DDP_HIP(hipMemsetAsync(k, 0, m * N * B * 8, st));
    DDP_HIP(hipMemsetAsync(Quu, 0, m * m * N * B * 8, st));
    DDP_HIP(hipMemsetAsync(Vx, 0, n * N * B * 8, st));
    DDP_HIP(hipMemsetAsync(Vxx, 0, n * n * N * B * 8, st));
    DDP_HIP(hipMemsetAsync(x, 0, n * N * B * 8, st));
    DDP_HIP(hipMemsetAsync(u, 0, m * N * B * 8, st));
    DDP_HIP(hipMemsetAsync(cost, 0, CL * B * 8, st));
    if (trace_cost && trace_cap > 0) DDP_HIP(hipMemsetAsync(trace_cost, 0, (size_t)trace_cap * B * 8, st));
    if (trace7 && trace_cap > 0) DDP_HIP(hipMemsetAsync(trace7, 0, (size_t)7 * trace_cap * B * 8, st));

    // ---- initial trajectory (iLQG.jl:181-192): first α for which the open-loop rollout of α·u0 stays bounded
    const double one = 1.0;
    if (prerolled) {                                     // iLQG.jl:193-197: x = x0, cost given or costfun(x, u); no divergence test
        hipLaunchKernelGGL(preroll_init_kernel, dim3((unsigned)B), dim3(64), 0, st, (int)n, (int)m, (int)N, (int)CL, x0, u0, cost0, s,
                           x, u, cost, x0c);
        if (!cost0)
            hipLaunchKernelGGL(costfun_kernel, dim3((unsigned)B), dim3(64), 0, st, p->kind, (int)n, (int)m, (int)N, (int)CL, p->Q, p->R,
                               p->goal[0], p->goal[1], p->goal[2], p->goal[3], x, u, (const int32_t *)nullptr, cost, s.csum);
        x0 = x0c;
    }
    for (size_t ai = 0; !prerolled && ai < na; ++ai) {
        const size_t tot = m * N * B;
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m * N, (int)B, oo->alpha[ai], u0,
                           s.div0, us);
        rc = ddp_forward_pass_f64_dev(h, p, nullptr, nullptr, x0, us, nullptr, &one, 1, lims, s.div0, xn, un, cn, cs);
        if (rc) return rc;
        hipLaunchKernelGGL(init_check_kernel, dim3((unsigned)B), dim3(64), 0, st, (int)n, (int)m, (int)N, (int)CL, xn, un, cn, cs, s,
                           x, u, cost);
        // most batches are done after the first α; poll only if there are more candidates to try
        if (ai + 1 < na) {
            DDP_HIP(hipMemsetAsync(counter, 0, 4, st));
            hipLaunchKernelGGL(count_ok_kernel, dim3(gB), dim3(256), 0, st, (int)B, s, counter);
            DDP_HIP(hipMemcpyAsync(h->h_pinned, counter, 4, hipMemcpyDeviceToHost, st));
            DDP_HIP(hipStreamSynchronize(st));
            if (h->h_pinned[0] == (int)B) break;
        }
    }
    DDP_HIP(hipMemsetAsync(counter, 0, 4, st));
    hipLaunchKernelGGL(init_finish_kernel, dim3(gB), dim3(256), 0, st, (int)B, s, counter);
    DDP_HIP(hipMemcpyAsync(h->h_pinned, counter, 4, hipMemcpyDeviceToHost, st));
    DDP_HIP(hipStreamSynchronize(st));
    int running = h->h_pinned[0];

    ddp_bp_desc d;
    d.n = (int)n; d.m = (int)m; d.N = (int)N; d.B = (int)B;
    d.fx_tv = pend ? 1 : p->dyn_tv; d.fx_batched = pend ? 1 : p->dyn_batched;
    d.cost_tv = 0; d.cost_batched = 0; d.regType = oo->regType; d.has_lims = lims != nullptr;
    const double *fx = pend ? fxw : p->A, *fu = pend ? fuw : p->Bm;

    // ---- the working set: the caller's arrays until the first compaction
    WorkSet user = {x, u, cost, K, k, Quu, Vx, Vxx, const_cast<double *>(x0), s, nullptr};
    WorkSet ws = user;
    ddp_problem pw = *p;                                   // the problem as the kernels see it (B = slots of the working set)
    size_t Bw = B;
    std::vector<void *> owned;                             // compacted working sets (freed on every way out)
    struct Free { std::vector<void *> &v; ~Free() { for (void *q : v) hipFree(q); } } free_owned{owned};
    // Two schedulers for THROUGHPUT-bound batches; both leave every per-trajectory result unchanged (tests/test_gpu_edge_cases.py).
    // Up to about two waves per SIMD a pass takes the same time however many trajectories are live (each wave walks its N steps at
    // the latency of one step), so neither helps there — measured on C3 at B = 4096: compaction 0.117 -> 0.117 s of back passes, the
    // grouped line search 0.083 -> 0.19 s because its groups run one after the other.  They are switched on by the size of the
    // working set, or forced / forbidden through the environment (tests, A/B timing).
    //  * compaction: the batch advances in lock step, so a finished trajectory keeps the wave it shares with live ones busy
    //    (C3: mean 59 iterations per solve, 264 for the slowest).  Once half of the slots have stopped the live ones move to a
    //    smaller working set.  Not for per-trajectory dynamics (their operands would have to move as well).
    //    DDP_ILQG_COMPACT=0: never;  =k (k > 1): whenever a working set of >= k slots is half empty.
    //  * line search in groups: the reference stops at the first accepted step size (iLQG.jl:267-281); groups [0,1) [1,3) [3,n_alpha)
    //    roll the later step sizes out only for the trajectories whose search is still open.  DDP_ILQG_LSGROUPS=0 / 1: never / always.
    const char *cenv = ddp_env(h, ENV_ILQG_COMPACT);
    bool may_compact = !(cenv && cenv[0] == '0') && !(p->kind == DDP_PROBLEM_LQ && p->dyn_batched);
    // trajectories per wave of the backward kernel the dispatcher picks (back_pass.hip) -> slots that make two waves per SIMD
    const double tpw = (n == 4 && m == 1) ? 4.0 : (n == 10 && m == 2) ? (B < 5120 ? 1.0 : 4.0) : (n > DDP_MAX_N_GENERIC ? 0.25 : 1.0);
    const size_t min_slots = (cenv && atoi(cenv) > 1) ? (size_t)atoi(cenv) : (size_t)(2048.0 * tpw);
    const char *genv = ddp_env(h, ENV_ILQG_LSGROUPS);
    const double rpw = pend ? 64.0 : (n <= 16 ? 4.0 : 1.0);                       // rollouts per wave of the forward kernels
    // Measured: it pays where the rollouts of all step sizes together are throughput-bound (two waves per SIMD and more) AND the first
    // step size is usually accepted, as in the linear-quadratic family (1 024 n=10 solves with 11 step sizes: 15.7 -> 10.6 ms); the
    // pendulum's line search regularly needs the later ones (B = 32 768: 0.93 vs 0.88 s of rollouts) and its lane-per-rollout
    // launches are latency-bound below that (B = 4 096: 0.083 -> 0.19 s).  Results are the same either way.
    const bool groups = na > 1 && (genv ? genv[0] == '1' : (p->kind == DDP_PROBLEM_LQ && (double)na * (double)B / rpw >= 2048.0));
    int32_t *more = (int32_t *)take(0);                    // two masks of B int32 were reserved behind x0c (see `bytes`)
    int32_t *more2 = more + B;

    int git = 0;
    constexpr int POLL = 4;
    const long hard_cap = 4L * oo->max_iter + 1000;      // every global iteration advances iter or λ of each running trajectory
    // ddp_ilqg_set_timing: the time_derivs / time_backward / time_forward keys of the reference's trace (iLQG.jl:227,241,281)
    // per global iteration, from HIP events on the stream (the loop synchronises once per iteration anyway)
    const bool timed = h->timing != nullptr;
    if (timed && !h->tev_ok) {
        for (int e = 0; e < 4; ++e) DDP_HIP(hipEventCreate(&h->tev[e]));
        h->tev_ok = true;
    }
    auto launch_stats = [&](int only_finished) {
        hipLaunchKernelGGL(stats_kernel, dim3((unsigned)((Bw + 255) / 256)), dim3(256), 0, st, (int)Bw, ws.s, ws.map, only_finished, stats);
    };
    bool polled = true;
    while (running > 0 && git < hard_cap) {
        if (polled && may_compact && (size_t)running * 2 <= Bw && Bw >= min_slots) {      // `running` is exact right after a poll only
            // ---- drop the finished slots.  Compaction is an optimisation: the block is allocated BEFORE anything is written to the
            // caller's arrays, and a failed allocation (a memory-tight batch) only switches compaction off for the rest of the solve.
            const size_t R = (size_t)running;
            const size_t w_d = al(n * N * R * 8) + al(m * N * R * 8) + al(CL * R * 8) + al(m * n * N * R * 8) + al(m * N * R * 8) +
                               al(m * m * N * R * 8) + al(n * N * R * 8) + al(n * n * N * R * 8) + al(n * R * 8) + 4 * al(R * 8) + 12 * al(R * 4) +
                               al(Bw * 4) + 256;
            void *blk = nullptr;
            if (ddp_env(h, ENV_TEST_COMPACT_ALLOC_FAIL) || hipMalloc(&blk, w_d) != hipSuccess) {      // (the variable: tests of this path)
                (void)hipGetLastError();                         // clear the sticky error: the solve goes on with the current working set
                may_compact = false;
                continue;
            }
            owned.push_back(blk);
            // their summary and (from a compacted set) their results go to the caller's arrays first
            launch_stats(1);
            if (ws.map) hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)Bw), dim3(64), 0, st, (int)n, (int)m, (int)N, (int)CL, 0, ws, user);
            char *q = (char *)blk;
            auto tk = [&](size_t b_) { void *r_ = q; q += al(b_); return r_; };
            WorkSet nw;
            nw.x = (double *)tk(n * N * R * 8); nw.u = (double *)tk(m * N * R * 8); nw.cost = (double *)tk(CL * R * 8);
            nw.K = (double *)tk(m * n * N * R * 8); nw.k = (double *)tk(m * N * R * 8); nw.Quu = (double *)tk(m * m * N * R * 8);
            nw.Vx = (double *)tk(n * N * R * 8); nw.Vxx = (double *)tk(n * n * N * R * 8); nw.x0 = (double *)tk(n * R * 8);
            nw.s.lam = (double *)tk(R * 8); nw.s.dlam = (double *)tk(R * 8); nw.s.gnorm = (double *)tk(R * 8); nw.s.csum = (double *)tk(R * 8);
            nw.s.status = (int32_t *)tk(R * 4); nw.s.iter = (int32_t *)tk(R * 4); nw.s.acc = (int32_t *)tk(R * 4); nw.s.nbp = (int32_t *)tk(R * 4);
            nw.s.nfp = (int32_t *)tk(R * 4); nw.s.flg = (int32_t *)tk(R * 4); nw.s.run = (int32_t *)tk(R * 4); nw.s.dodf = (int32_t *)tk(R * 4);
            nw.s.dofwd = (int32_t *)tk(R * 4); nw.s.div0 = (int32_t *)tk(R * 4);
            nw.map = (int32_t *)tk(R * 4);
            int32_t *idx = (int32_t *)tk(Bw * 4);
            hipLaunchKernelGGL(live_index_kernel, dim3(1), dim3(64), 0, st, (int)Bw, ws.s, idx, counter);
            hipLaunchKernelGGL(gather_kernel, dim3((unsigned)R), dim3(64), 0, st, (int)n, (int)m, (int)N, (int)CL, idx, ws, nw);
            ws = nw;
            Bw = R;
            pw.B = (int)R; d.B = (int)R;
        }
        polled = false;
        if (timed) DDP_HIP(hipEventRecord(h->tev[0], st));
        rc = ddp_df_f64_dev(h, &pw, ws.x, ws.u, ws.s.dodf, cx, cu, fxw, fuw);                                  // STEP 1
        if (rc) return rc;
        if (timed) DDP_HIP(hipEventRecord(h->tev[1], st));
        rc = ddp_launch_back_pass(h, &d, cx, cu, p->Q, cxu, p->R, fx, fu, ws.s.lam, lims, ws.u, ws.s.run, ws.K, ws.k, ws.Quu, ws.Vx, ws.Vxx,
                                  dV, div);                                                                  // STEP 2
        if (rc) return rc;
        hipLaunchKernelGGL(post_bp_kernel, dim3((unsigned)Bw), dim3(64), 0, st, (int)m, (int)N, o, div, ws.k, ws.u, ws.s);
        if (timed) DDP_HIP(hipEventRecord(h->tev[2], st));
        // STEP 3: all step sizes at once, or in groups with the later ones masked by `more`
        const size_t gb[4] = {0, groups ? 1 : na, groups ? (na < 3 ? na : 3) : na, na};
        const int32_t *mask = ws.s.dofwd;
        for (int gi = 0; gi < 3; ++gi) {
            const size_t a0 = gb[gi], a1 = gb[gi + 1];
            if (a1 <= a0) continue;
            if (a0 > 0) {
                int32_t *mk = (gi == 1) ? more : more2;
                hipLaunchKernelGGL(ls_more_kernel, dim3((unsigned)((Bw + 255) / 256)), dim3(256), 0, st, (int)Bw, (int)a0, o, dV, cs, ws.s,
                                   gi == 1 ? (const int32_t *)nullptr : (const int32_t *)more, mk);
                mask = mk;
            }
            rc = ddp_forward_pass_f64_dev(h, &pw, ws.K, ws.k, ws.x0, ws.u, ws.x, o.alpha + a0, (int)(a1 - a0), lims, mask, xn + n * N * Bw * a0,
                                          un + m * N * Bw * a0, cn + CL * Bw * a0, cs + Bw * a0);
            if (rc) return rc;
        }
        if (timed) DDP_HIP(hipEventRecord(h->tev[3], st));
        // The host looks at the number of running trajectories every POLL iterations only (every iteration while the per-iteration
        // timing keys are recorded): a synchronisation per iteration leaves the GPU idle for a round trip, and iterations launched
        // after the last trajectory has stopped find nothing to do (every wave leaves at once).  Each iteration of a group writes
        // its own count, so `global_iters` stays exact.
        const int slot = git % POLL;
        DDP_HIP(hipMemsetAsync(counter + slot, 0, 4, st));
        hipLaunchKernelGGL(accept_kernel, dim3((unsigned)Bw), dim3(64), 0, st, (int)n, (int)m, (int)N, (int)Bw, (int)CL, o, dV, xn,
                           un, cn, cs, ws.s, ws.x, ws.u, ws.cost, ws.k, trace_cap, trace_cost, trace7, ws.map, counter + slot);    // STEP 4
        ++git;
        if (timed || slot == POLL - 1 || git >= hard_cap) {
            const int cnt = slot + 1;
            DDP_HIP(hipMemcpyAsync(h->h_pinned, counter, 4 * cnt, hipMemcpyDeviceToHost, st));
            DDP_HIP(hipStreamSynchronize(st));
            running = h->h_pinned[cnt - 1];
            polled = true;
            for (int e2 = 0; e2 < cnt; ++e2)
                if (h->h_pinned[e2] == 0) { git -= cnt - 1 - e2; running = 0; break; }     // the iterations after it did nothing
            if (timed && git - 1 < h->timing_cap) {
                for (int e = 0; e < 3; ++e) {
                    float ms = 0.0f;
                    DDP_HIP(hipEventElapsedTime(&ms, h->tev[e], h->tev[e + 1]));
                    h->timing[(size_t)h->timing_cap * e + (git - 1)] = 1e-3 * (double)ms;
                }
            }
        }
    }
    launch_stats(0);                                       // trajectories still running here: DDP_EXIT_CAP (the driver's own bound)
    if (ws.map) hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)Bw), dim3(64), 0, st, (int)n, (int)m, (int)N, (int)CL, 1, ws, user);
    DDP_HIP(hipGetLastError());
    DDP_HIP(hipStreamSynchronize(st));
    if (global_iters) *global_iters = git;
    return 0;
}

int ddp_ilqg_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *oo, const double *x0, const double *u0,
                     const double *lims, double *x, double *u, double *K, double *k, double *Quu, double *Vx,
                     double *Vxx, double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters)
{
    return ilqg_impl(h, p, oo, x0, u0, lims, x, u, K, k, Quu, Vx, Vxx, cost, stats, trace_cap, trace_cost, global_iters, false, nullptr);
}

int ddp_ilqg_warm_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *oo, const double *x0, const double *u0,
                          const double *cost0, const double *lims, double *x, double *u, double *K, double *k, double *Quu,
                          double *Vx, double *Vxx, double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters)
{
    return ilqg_impl(h, p, oo, x0, u0, lims, x, u, K, k, Quu, Vx, Vxx, cost, stats, trace_cap, trace_cost, global_iters, true, cost0);
}

// ---- the slot scheduler (kernels above).  S slots, P problems (queue: P >= S; MPC: S == P, `steps` solves per trajectory).
static int ilqg_sched_impl(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *oo, int slots, int mpc_steps, int zero_tail,
                           const double *x0, const double *u0, const double *lims, double *x, double *u, double *K, double *k,
                           double *Quu, double *Vx, double *Vxx, double *cost, double *stats, double *xcl, double *ucl,
                           double *stats_cl, int *global_iters)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && p && x0 && u0 && x && u, "ilqg_sched: null argument");
    ddp_ilqg_opts od;
    if (!oo) { ddp_ilqg_default_opts(&od); oo = &od; }
    DDP_CHECK(oo->n_alpha >= 1 && oo->n_alpha <= 16, "ilqg_sched: n_alpha=%d out of [1,16]", oo->n_alpha);
    const size_t n = p->n, m = p->m, N = p->N, P = p->B, na = oo->n_alpha, CL = ddp_cost_len(p);
    const bool pend = p->kind == DDP_PROBLEM_PENDCART, mpc = mpc_steps > 0;
    DDP_CHECK(N >= 2, "ilqg_sched: N=%d (at least two time steps)", (int)N);
    if (mpc) slots = (int)P;                                            // every trajectory keeps its slot
    if (slots <= 0 || (size_t)slots > P) slots = (int)(P < 4096 ? P : 4096);
    const size_t S = (size_t)slots;
    DDP_CHECK(!(p->kind == DDP_PROBLEM_LQ && p->dyn_batched) || mpc, "ilqg_queue: per-trajectory dynamics (dyn_batched) are not supported by the queue");
    DDP_CHECK(mpc ? (xcl && ucl && stats_cl) : (K && k && Quu && Vx && Vxx && cost && stats), "ilqg_sched: null output");

    // ---- one block: working set of S slots, derivative / candidate workspace, scheduler state
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t f_x = al(n * N * S * 8), f_u = al(m * N * S * 8), f_c = al(CL * S * 8), f_K = al(m * n * N * S * 8), f_Q = al(m * m * N * S * 8),
                 f_V = al(n * n * N * S * 8), f_fx = pend ? al(n * n * N * S * 8) : 0, f_fu = pend ? al(n * m * N * S * 8) : 0,
                 f_d = al(S * 8), f_i = al(S * 4);
    const size_t bytes = 2 * f_x /* x, Vx */ + 2 * f_u /* u, k */ + f_c + f_K + f_Q + f_V + al(n * S * 8) /* x0s */ + 2 * f_u /* u0s, us */ +
                         f_x + f_u /* cx, cu */ + f_fx + f_fu + na * (f_x + f_u + f_c + f_d) /* candidates */ + (f_x + f_u + f_c + f_d) /* initial rollout */ +
                         al(2 * S * 8) + f_i /* dV, div */ + al(n * m * 8) + 4 * f_d + 10 * f_i /* Traj */ + 6 * f_i /* scheduler */ + 2 * f_i /* more */ + 512;
    void *blk = nullptr;
    DDP_HIP(hipMalloc(&blk, bytes));
    struct Free { void *q; ~Free() { hipFree(q); } } free_blk{blk};
    char *pp = (char *)blk;
    auto take = [&](size_t b) { void *r = pp; pp += b; return r; };
    WorkSet ws;
    ws.x = (double *)take(f_x); ws.Vx = (double *)take(f_x); ws.u = (double *)take(f_u); ws.k = (double *)take(f_u); ws.cost = (double *)take(f_c);
    ws.K = (double *)take(f_K); ws.Quu = (double *)take(f_Q); ws.Vxx = (double *)take(f_V); ws.x0 = (double *)take(al(n * S * 8));
    double *u0s = (double *)take(f_u), *us = (double *)take(f_u);
    double *cx = (double *)take(f_x), *cu = (double *)take(f_u), *fxw = pend ? (double *)take(f_fx) : nullptr, *fuw = pend ? (double *)take(f_fu) : nullptr;
    double *xn = (double *)take(na * f_x), *un = (double *)take(na * f_u), *cn = (double *)take(na * f_c), *cs = (double *)take(na * f_d);
    double *xi = (double *)take(f_x), *ui = (double *)take(f_u), *ci = (double *)take(f_c), *csi = (double *)take(f_d);
    double *dV = (double *)take(al(2 * S * 8));
    int32_t *div = (int32_t *)take(f_i);
    double *cxu = (double *)take(al(n * m * 8));
    Traj &s = ws.s;
    s.lam = (double *)take(f_d); s.dlam = (double *)take(f_d); s.gnorm = (double *)take(f_d); s.csum = (double *)take(f_d);
    s.status = (int32_t *)take(f_i); s.iter = (int32_t *)take(f_i); s.acc = (int32_t *)take(f_i); s.nbp = (int32_t *)take(f_i);
    s.nfp = (int32_t *)take(f_i); s.flg = (int32_t *)take(f_i); s.run = (int32_t *)take(f_i); s.dodf = (int32_t *)take(f_i);
    s.dofwd = (int32_t *)take(f_i); s.div0 = (int32_t *)take(f_i);
    Sched q;
    q.P = (int)P; q.mpc_steps = mpc_steps; q.zero_tail = zero_tail; q.nalpha = (int)na;
    for (int i = 0; i < 16; ++i) q.alpha[i] = oo->alpha[i];
    q.lam0 = oo->lambda; q.dlam0 = oo->dlambda; q.x0g = x0; q.u0g = u0; q.u0s = u0s; q.us = us; q.x0s = ws.x0;
    q.initm = (int32_t *)take(f_i); q.ai = (int32_t *)take(f_i); q.map = (int32_t *)take(f_i); q.left = (int32_t *)take(f_i);
    q.noflush = (int32_t *)take(f_i); q.ready = (int32_t *)take(f_i);
    int32_t *more = (int32_t *)take(f_i), *more2 = (int32_t *)take(f_i);
    int *counter = (int *)take(256);
    q.qhead = (int *)take(256);
    q.x = x; q.u = u; q.cost = cost; q.K = K; q.k = k; q.Quu = Quu; q.Vx = Vx; q.Vxx = Vxx; q.stats = stats; q.xcl = xcl; q.ucl = ucl; q.stats_cl = stats_cl;
    ws.map = q.map;
    DDP_CHECK(h->h_pinned, "ilqg_sched: pinned poll buffer missing");

    Opt o;
    o.lfac = oo->lambda_factor; o.lmax = oo->lambda_max; o.lmin = oo->lambda_min; o.tol_fun = oo->tol_fun;
    o.tol_grad = oo->tol_grad; o.rrmin = oo->reduce_ratio_min; o.max_iter = oo->max_iter; o.nalpha = (int)na;
    for (int i = 0; i < 16; ++i) o.alpha[i] = oo->alpha[i];

    hipStream_t st = h->stream;
    if (!h->sched_aux) {
        DDP_HIP(hipStreamCreateWithFlags(&h->sched_aux, hipStreamNonBlocking));
        for (int e = 0; e < 2; ++e) DDP_HIP(hipEventCreateWithFlags(&h->sched_ev[e], hipEventDisableTiming));
    }
    // slots start empty: nothing runs, nothing to flush (map = -1); the working set is finite from the start (the line-search launches
    // touch every slot's operands only under their masks, but a masked wave may still prefetch)
    DDP_HIP(hipMemsetAsync(blk, 0, bytes, st));
    DDP_HIP(hipMemsetAsync(q.map, 0xff, S * 4, st));
    // outputs of problems that never produce any (initial divergence) are zero, like the arrays ilqg_impl clears
    if (!mpc) {
        DDP_HIP(hipMemsetAsync(K, 0, m * n * N * P * 8, st)); DDP_HIP(hipMemsetAsync(k, 0, m * N * P * 8, st));
        DDP_HIP(hipMemsetAsync(Quu, 0, m * m * N * P * 8, st)); DDP_HIP(hipMemsetAsync(Vx, 0, n * N * P * 8, st));
        DDP_HIP(hipMemsetAsync(Vxx, 0, n * n * N * P * 8, st)); DDP_HIP(hipMemsetAsync(cost, 0, CL * P * 8, st));
    } else {
        DDP_HIP(hipMemsetAsync(xcl, 0, n * (size_t)(mpc_steps + 1) * P * 8, st)); DDP_HIP(hipMemsetAsync(ucl, 0, m * (size_t)mpc_steps * P * 8, st));
        DDP_HIP(hipMemsetAsync(stats_cl, 0, (size_t)DDP_ILQG_NSTATS * mpc_steps * P * 8, st));
    }
    DDP_HIP(hipMemsetAsync(x, 0, n * N * P * 8, st));
    DDP_HIP(hipMemsetAsync(u, 0, m * N * P * 8, st));

    ddp_problem pw = *p;
    pw.B = (int)S;
    ddp_bp_desc d;
    d.n = (int)n; d.m = (int)m; d.N = (int)N; d.B = (int)S;
    d.fx_tv = pend ? 1 : p->dyn_tv; d.fx_batched = pend ? 1 : p->dyn_batched;
    d.cost_tv = 0; d.cost_batched = 0; d.regType = oo->regType; d.has_lims = lims != nullptr;
    const double *fx = pend ? fxw : p->A, *fu = pend ? fuw : p->Bm;
    // the line search in the groups ilqg_impl uses at this batch size (same launches -> same kernels -> same bits per solve)
    const char *genv = ddp_env(h, ENV_ILQG_LSGROUPS);
    const double rpw = pend ? 64.0 : (n <= 16 ? 4.0 : 1.0);
    const bool groups = na > 1 && (genv ? genv[0] == '1' : (p->kind == DDP_PROBLEM_LQ && (double)na * (double)S / rpw >= 2048.0));
    const double one = 1.0;

    hipLaunchKernelGGL(sched_take_kernel, dim3((unsigned)S), dim3(TAKE_T), 0, st, (int)n, (int)m, (int)N, (int)CL, q, ws, counter);      // initial fill
    int git = 0, running = (int)S;
    constexpr int POLL = 4;
    const long hard_cap = ((long)(mpc ? mpc_steps : (P + S - 1) / S) + 1) * (4L * oo->max_iter + 1000 + na);
    int rc = 0;
    while (running > 0 && git < hard_cap) {
        // armed slots: the initial rollout of α·u0 with an empty policy, its bound test, the step to the next α (iLQG.jl:181-192, :205-210)
        // — on the side stream, beside STEPS 1-3 of the running slots (a 600-step rollout of a handful of slots costs the full latency
        // of the kernel: 0.19 of the 1.2 ms a global iteration of 4 096 pendulums takes); the slots join at the next sched_take_kernel
        DDP_HIP(hipEventRecord(h->sched_ev[0], st));
        DDP_HIP(hipStreamWaitEvent(h->sched_aux, h->sched_ev[0], 0));
        h->stream = h->sched_aux;
        rc = ddp_forward_pass_f64_dev(h, &pw, nullptr, nullptr, ws.x0, us, nullptr, &one, 1, lims, q.initm, xi, ui, ci, csi);
        h->stream = st;
        if (rc) return rc;
        hipLaunchKernelGGL(init_check_kernel, dim3((unsigned)S), dim3(64), 0, h->sched_aux, (int)n, (int)m, (int)N, (int)CL, xi, ui, ci, csi, s, ws.x, ws.u, ws.cost);
        hipLaunchKernelGGL(sched_init_advance_kernel, dim3((unsigned)S), dim3(64), 0, h->sched_aux, (int)m, (int)N, q, s);
        DDP_HIP(hipEventRecord(h->sched_ev[1], h->sched_aux));
        rc = ddp_df_f64_dev(h, &pw, ws.x, ws.u, s.dodf, cx, cu, fxw, fuw);                                     // STEP 1
        if (rc) return rc;
        rc = ddp_launch_back_pass(h, &d, cx, cu, p->Q, cxu, p->R, fx, fu, s.lam, lims, ws.u, s.run, ws.K, ws.k, ws.Quu, ws.Vx, ws.Vxx, dV, div);   // STEP 2
        if (rc) return rc;
        hipLaunchKernelGGL(post_bp_kernel, dim3((unsigned)S), dim3(64), 0, st, (int)m, (int)N, o, div, ws.k, ws.u, s);
        const size_t gb[4] = {0, groups ? 1 : na, groups ? (na < 3 ? na : 3) : na, na};                         // STEP 3
        const int32_t *mask = s.dofwd;
        for (int gi = 0; gi < 3; ++gi) {
            const size_t a0 = gb[gi], a1 = gb[gi + 1];
            if (a1 <= a0) continue;
            if (a0 > 0) {
                int32_t *mk = (gi == 1) ? more : more2;
                hipLaunchKernelGGL(ls_more_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, (int)S, (int)a0, o, dV, cs, s,
                                   gi == 1 ? (const int32_t *)nullptr : (const int32_t *)more, mk);
                mask = mk;
            }
            rc = ddp_forward_pass_f64_dev(h, &pw, ws.K, ws.k, ws.x0, ws.u, ws.x, o.alpha + a0, (int)(a1 - a0), lims, mask, xn + n * N * S * a0,
                                          un + m * N * S * a0, cn + CL * S * a0, cs + S * a0);
            if (rc) return rc;
        }
        const int slot = git % POLL;
        DDP_HIP(hipMemsetAsync(counter + slot, 0, 4, st));
        hipLaunchKernelGGL(accept_kernel, dim3((unsigned)S), dim3(64), 0, st, (int)n, (int)m, (int)N, (int)S, (int)CL, o, dV, xn, un, cn, cs, s,
                           ws.x, ws.u, ws.cost, ws.k, 0, (double *)nullptr, (double *)nullptr, (const int32_t *)nullptr, counter + 32);      // STEP 4
        DDP_HIP(hipStreamWaitEvent(st, h->sched_ev[1], 0));
        hipLaunchKernelGGL(sched_take_kernel, dim3((unsigned)S), dim3(TAKE_T), 0, st, (int)n, (int)m, (int)N, (int)CL, q, ws, counter + slot);
        ++git;
        if (slot == POLL - 1 || git >= hard_cap) {
            DDP_HIP(hipMemcpyAsync(h->h_pinned, counter, 4 * (slot + 1), hipMemcpyDeviceToHost, st));
            DDP_HIP(hipStreamSynchronize(st));
            running = h->h_pinned[slot];
            for (int e2 = 0; e2 <= slot; ++e2)
                if (h->h_pinned[e2] == 0) { git -= slot - e2; running = 0; break; }       // the iterations after it did nothing
        }
    }
    DDP_HIP(hipGetLastError());
    DDP_HIP(hipStreamSynchronize(st));
    DDP_CHECK(running == 0, "ilqg_sched: %d slots still busy after %d global iterations (the driver's own bound)", running, git);
    if (global_iters) *global_iters = git;
    return 0;
}

int ddp_ilqg_queue_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int slots, const double *x0, const double *u0,
                           const double *lims, double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                           double *cost, double *stats, int *global_iters)
{
    return ilqg_sched_impl(h, p, o, slots, 0, 0, x0, u0, lims, x, u, K, k, Quu, Vx, Vxx, cost, stats, nullptr, nullptr, nullptr, global_iters);
}

int ddp_ilqg_mpc_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int steps, int zero_tail, const double *x0,
                         const double *u0, const double *lims, double *xcl, double *ucl, double *stats_cl, double *x, double *u,
                         int *global_iters)
{
    DDP_CHECK(steps >= 1, "ilqg_mpc: steps=%d", steps);
    return ilqg_sched_impl(h, p, o, 0, steps, zero_tail, x0, u0, lims, x, u, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           xcl, ucl, stats_cl, global_iters);
}

// batch-level line-search statistics in ONE launch: out[4] = [Σ_b csum, Σ_b dV[1,b], Σ_b dV[2,b], #diverged] — the vector a
// multi-GPU job all-reduces once per pass (bench.py, sharding.py)
__global__ __launch_bounds__(256) void batch_stats_kernel(int B, const double *csum, const double *dV, const int32_t *diverge, double *out)
{
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        a0 += csum ? csum[b] : 0.0;
        if (dV) { a1 += dV[2 * b]; a2 += dV[2 * b + 1]; }
        a3 += (diverge && diverge[b] != 0) ? 1.0 : 0.0;
    }
    __shared__ double red[4][4];
    for (int off = 32; off >= 1; off >>= 1) {
        a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64); a3 += __shfl_xor(a3, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = a0; red[w][1] = a1; red[w][2] = a2; red[w][3] = a3; }
    __syncthreads();
    if (threadIdx.x < 4) out[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

int ddp_batch_stats_f64_dev(ddp_handle h, int B, const double *csum, const double *dV, const int32_t *diverge, double *out4)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && out4 && B >= 1, "batch_stats: bad argument");
    hipLaunchKernelGGL(batch_stats_kernel, dim3(1), dim3(256), 0, h->stream, B, csum, dV, diverge, out4);
    DDP_HIP(hipGetLastError());
    return 0;
}

// everything at once: optional pre-rolled x0 (+ cost0) and the seven per-iteration trace keys of the reference
// (iLQG.jl:257,325-330): trace7[7, trace_cap, B] rows λ, dλ, α (NaN: no step accepted), improvement, cost, reduce_ratio, grad_norm
int ddp_ilqg_ex_f64_dev(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *oo, const double *x0, int x0_prerolled,
                        const double *u0, const double *cost0, const double *lims, double *x, double *u, double *K, double *k,
                        double *Quu, double *Vx, double *Vxx, double *cost, double *stats, int trace_cap, double *trace7,
                        int *global_iters)
{
    return ilqg_impl(h, p, oo, x0, u0, lims, x, u, K, k, Quu, Vx, Vxx, cost, stats, trace_cap, nullptr, global_iters, x0_prerolled != 0,
                     x0_prerolled ? cost0 : nullptr, trace7);
}

namespace {
// dst[:, i, b] = src[:, min(i + shift, N-1), b]  (tail: the last column repeated, or zeros)
__global__ __launch_bounds__(256) void mpc_shift_kernel(int d, int N, long total, int shift, int zero_tail, const double *src, double *dst)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long per = (long)d * N;
    const long b = e / per, r = e % per;
    const int i = (int)(r / d), c = (int)(r % d);
    const int is = i + shift;
    dst[e] = (is < N) ? src[b * per + (long)is * d + c] : (zero_tail ? 0.0 : src[b * per + (long)(N - 1) * d + c]);
}
}   // namespace

int ddp_mpc_shift_f64_dev(ddp_handle h, int d, int N, int B, int shift, int zero_tail, const double *src, double *dst)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && src && dst && src != dst, "mpc_shift: null or aliased argument (out of place only)");
    DDP_CHECK(d > 0 && N > 0 && B > 0 && shift >= 0, "mpc_shift: d=%d N=%d B=%d shift=%d", d, N, B, shift);
    const long total = (long)d * N * B;
    hipLaunchKernelGGL(mpc_shift_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, d, N, total, shift, zero_tail, src, dst);
    DDP_HIP(hipGetLastError());
    return 0;
}

int ddp_costfun_f64_dev(ddp_handle h, const ddp_problem *p, const double *x, const double *u, const int32_t *active,
                        double *cost, double *csum)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && p && x && u && cost, "costfun: null argument");
    hipLaunchKernelGGL(costfun_kernel, dim3((unsigned)p->B), dim3(64), 0, h->stream, p->kind, p->n, p->m, p->N, ddp_cost_len(p), p->Q, p->R,
                       p->goal[0], p->goal[1], p->goal[2], p->goal[3], x, u, active, cost, csum);
    DDP_HIP(hipGetLastError());
    return 0;
}

static int ilqg_host(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, const double *x0, const double *u0,
                     const double *cost0, bool prerolled,
                     const double *lims, double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                     double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters, double *trace7 = nullptr)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && p && x0 && u0, "ilqg: null argument");
    const size_t n = p->n, m = p->m, N = p->N, B = p->B, CL = ddp_cost_len(p);
    const size_t dc = (p->dyn_tv ? N : 1) * (p->dyn_batched ? B : 1);
    // the driver uses the handle's scratch itself, so the host flavour owns separate allocations
    // cost_diag = 1 is a declaration about Q, R: verified here, on the host copies, before anything is staged
    { const int rd_ = ddp_check_cost_diag_host(p); if (rd_) return rd_; }
    DiagVerified diag_verified_(h);                          // Q, R were tested on the host; the staged copies need no second test
    struct Buf { void *d; void *hdst; size_t bytes; };
    std::vector<Buf> bufs;
    bool failed = false;
    auto dev = [&](const void *src, void *dst, size_t bytes) -> void * {
        void *d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) { failed = true; return nullptr; }
        if (src && hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess) failed = true;
        bufs.push_back({d, dst, bytes});
        return d;
    };
    ddp_problem pd = *p;
    if (p->kind == DDP_PROBLEM_LQ) { pd.A = (double *)dev(p->A, nullptr, n * n * dc * 8); pd.Bm = (double *)dev(p->Bm, nullptr, n * m * dc * 8); }
    pd.Q = (double *)dev(p->Q, nullptr, n * n * 8);
    pd.R = (double *)dev(p->R, nullptr, m * m * 8);
    double *dx0 = (double *)dev(x0, nullptr, n * (prerolled ? N : 1) * B * 8), *du0 = (double *)dev(u0, nullptr, m * N * B * 8),
           *dl = lims ? (double *)dev(lims, nullptr, 2 * m * 8) : nullptr,
           *dc0 = (prerolled && cost0) ? (double *)dev(cost0, nullptr, CL * B * 8) : nullptr;
    double *dx = (double *)dev(nullptr, x, n * N * B * 8), *du = (double *)dev(nullptr, u, m * N * B * 8),
           *dK = (double *)dev(nullptr, K, m * n * N * B * 8), *dk = (double *)dev(nullptr, k, m * N * B * 8),
           *dQuu = (double *)dev(nullptr, Quu, m * m * N * B * 8), *dVx = (double *)dev(nullptr, Vx, n * N * B * 8),
           *dVxx = (double *)dev(nullptr, Vxx, n * n * N * B * 8), *dcost = (double *)dev(nullptr, cost, CL * B * 8),
           *dstats = (double *)dev(nullptr, stats, DDP_ILQG_NSTATS * B * 8),
           *dtr = (trace_cost && trace_cap > 0) ? (double *)dev(nullptr, trace_cost, (size_t)trace_cap * B * 8) : nullptr,
           *dt7 = (trace7 && trace_cap > 0) ? (double *)dev(nullptr, trace7, (size_t)7 * trace_cap * B * 8) : nullptr;
    int rc = failed ? -2 : 0;
    if (failed) ddp_set_error("ilqg: device allocation / upload failed");
    if (!rc) rc = ilqg_impl(h, &pd, o, dx0, du0, dl, dx, du, dK, dk, dQuu, dVx, dVxx, dcost, dstats, trace_cap, dtr, global_iters, prerolled, dc0, dt7);
    if (!rc)
        for (auto &bf : bufs)
            if (bf.hdst && hipMemcpyAsync(bf.hdst, bf.d, bf.bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = -2;
    hipStreamSynchronize(h->stream);
    for (auto &bf : bufs) hipFree(bf.d);
    return rc;
}

int ddp_ilqg_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, const double *x0, const double *u0,
                 const double *lims, double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                 double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters)
{
    return ilqg_host(h, p, o, x0, u0, nullptr, false, lims, x, u, K, k, Quu, Vx, Vxx, cost, stats, trace_cap, trace_cost, global_iters);
}

int ddp_ilqg_warm_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, const double *x0, const double *u0,
                      const double *cost0, const double *lims, double *x, double *u, double *K, double *k, double *Quu,
                      double *Vx, double *Vxx, double *cost, double *stats, int trace_cap, double *trace_cost, int *global_iters)
{
    return ilqg_host(h, p, o, x0, u0, cost0, true, lims, x, u, K, k, Quu, Vx, Vxx, cost, stats, trace_cap, trace_cost, global_iters);
}

int ddp_ilqg_ex_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, const double *x0, int x0_prerolled,
                    const double *u0, const double *cost0, const double *lims, double *x, double *u, double *K, double *k,
                    double *Quu, double *Vx, double *Vxx, double *cost, double *stats, int trace_cap, double *trace7,
                    int *global_iters)
{
    return ilqg_host(h, p, o, x0, u0, x0_prerolled ? cost0 : nullptr, x0_prerolled != 0, lims, x, u, K, k, Quu, Vx, Vxx, cost, stats,
                     trace_cap, nullptr, global_iters, trace7);
}

// host-pointer flavours of the slot scheduler: upload, run, download (the same staging as ilqg_host)
static int sched_host(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int slots, int steps, int zero_tail, const double *x0,
                      const double *u0, const double *lims, double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx,
                      double *cost, double *stats, double *xcl, double *ucl, double *stats_cl, int *global_iters)
{
    DDP_DEVICE(h);
    DDP_CHECK(h && p && x0 && u0, "ilqg_sched: null argument");
    const size_t n = p->n, m = p->m, N = p->N, P = p->B, CL = ddp_cost_len(p);
    const size_t dc = (p->dyn_tv ? N : 1) * (p->dyn_batched ? P : 1);
    // cost_diag = 1 is a declaration about Q, R: verified here, on the host copies, before anything is staged
    { const int rd_ = ddp_check_cost_diag_host(p); if (rd_) return rd_; }
    DiagVerified diag_verified_(h);                          // Q, R were tested on the host; the staged copies need no second test
    struct Buf { void *d; void *hdst; size_t bytes; };
    std::vector<Buf> bufs;
    bool failed = false;
    auto dev = [&](const void *src, void *dst, size_t bytes) -> void * {
        if (!src && !dst) return nullptr;
        void *d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) { failed = true; return nullptr; }
        if (src && hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess) failed = true;
        bufs.push_back({d, dst, bytes});
        return d;
    };
    ddp_problem pd = *p;
    if (p->kind == DDP_PROBLEM_LQ) { pd.A = (double *)dev(p->A, nullptr, n * n * dc * 8); pd.Bm = (double *)dev(p->Bm, nullptr, n * m * dc * 8); }
    pd.Q = (double *)dev(p->Q, nullptr, n * n * 8);
    pd.R = (double *)dev(p->R, nullptr, m * m * 8);
    double *dx0 = (double *)dev(x0, nullptr, n * P * 8), *du0 = (double *)dev(u0, nullptr, m * N * P * 8),
           *dl = lims ? (double *)dev(lims, nullptr, 2 * m * 8) : nullptr;
    double *dx = (double *)dev(nullptr, x, n * N * P * 8), *du = (double *)dev(nullptr, u, m * N * P * 8),
           *dK = (double *)dev(nullptr, K, m * n * N * P * 8), *dk = (double *)dev(nullptr, k, m * N * P * 8),
           *dQuu = (double *)dev(nullptr, Quu, m * m * N * P * 8), *dVx = (double *)dev(nullptr, Vx, n * N * P * 8),
           *dVxx = (double *)dev(nullptr, Vxx, n * n * N * P * 8), *dcost = (double *)dev(nullptr, cost, CL * P * 8),
           *dstats = (double *)dev(nullptr, stats, DDP_ILQG_NSTATS * P * 8),
           *dxcl = (double *)dev(nullptr, xcl, n * (size_t)(steps + 1) * P * 8), *ducl = (double *)dev(nullptr, ucl, m * (size_t)steps * P * 8),
           *dscl = (double *)dev(nullptr, stats_cl, (size_t)DDP_ILQG_NSTATS * steps * P * 8);
    int rc = failed ? -2 : 0;
    if (failed) ddp_set_error("ilqg_sched: device allocation / upload failed");
    if (!rc) rc = ilqg_sched_impl(h, &pd, o, slots, steps, zero_tail, dx0, du0, dl, dx, du, dK, dk, dQuu, dVx, dVxx, dcost, dstats, dxcl, ducl, dscl,
                                  global_iters);
    if (!rc)
        for (auto &bf : bufs)
            if (bf.hdst && hipMemcpyAsync(bf.hdst, bf.d, bf.bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = -2;
    hipStreamSynchronize(h->stream);
    for (auto &bf : bufs) hipFree(bf.d);
    return rc;
}

int ddp_ilqg_queue_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int slots, const double *x0, const double *u0,
                       const double *lims, double *x, double *u, double *K, double *k, double *Quu, double *Vx, double *Vxx, double *cost,
                       double *stats, int *global_iters)
{
    DDP_CHECK(x && u && K && k && Quu && Vx && Vxx && cost && stats, "ilqg_queue: null output");
    return sched_host(h, p, o, slots, 0, 0, x0, u0, lims, x, u, K, k, Quu, Vx, Vxx, cost, stats, nullptr, nullptr, nullptr, global_iters);
}

int ddp_ilqg_mpc_f64(ddp_handle h, const ddp_problem *p, const ddp_ilqg_opts *o, int steps, int zero_tail, const double *x0, const double *u0,
                     const double *lims, double *xcl, double *ucl, double *stats_cl, double *x, double *u, int *global_iters)
{
    DDP_CHECK(steps >= 1 && xcl && ucl && stats_cl && x && u, "ilqg_mpc: steps=%d or a null output", steps);
    return sched_host(h, p, o, 0, steps, zero_tail, x0, u0, lims, x, u, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, xcl, ucl,
                      stats_cl, global_iters);
}

}   // extern "C"


int ddp_ilqg_set_timing(ddp_handle h, double *host_buf, int cap)
{
    DDP_CHECK(h, "ilqg_set_timing: null handle");
    DDP_CHECK(host_buf == nullptr || cap > 0, "ilqg_set_timing: cap=%d", cap);
    h->timing = host_buf;
    h->timing_cap = host_buf ? cap : 0;
    return 0;
}
