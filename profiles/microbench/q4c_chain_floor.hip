// The floor of ONE step of the pendulum backward pass (n = 4, m = 1: csrc/back_pass_q4.hip, `q4_step` — the step function the production
// kernels back_pass_q4c / q4p / q4 call; src/backward_pass.jl:165-177 + :28-79, boxQP.jl:58-151 for m = 1) on one wave of gfx950, with
// everything that is not on the dependency chain taken away.  The step function is the PRODUCTION one (this file includes the kernel
// source): what is removed is what the kernels put around it.
//   mode 0  the step without control limits: 12 v_mfma_f64_4x4x4_4b in their dependency pattern (Vxx'fu -> Quu, Qux; Vxx fx -> Qxx),
//           the scalar gain (reciprocal + Newton), the value update — operands in registers (made opaque per step so that nothing is hoisted)
//   mode 1  + control limits: the straight-line two-iteration box-QP (`boxqp1_two_iterations`; limits far away: the usual free path)
//   mode 2  + regType 2 (the three extra products fx'fu, fu'fx, fu'fu)
//   mode 3  + the step record as the chunk kernels write it: the previous step's five outputs into an LDS image behind the products
//   mode 4  + the operand fetch of the next step from an LDS image (ten LDS reads per step, issued a step ahead)
// Prints ns and shader-clock ticks per step.  Production (C3: pendcart, N = 600, B = 4 096, limits, regType 2, one wave per SIMD):
// 0.389 ms / 599 steps = 650 ns per step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I../../differentialdynamicprogramming.jl_amd/csrc q4c_chain_floor.hip -o q4c_chain_floor
#include <cstdarg>
#include "../../differentialdynamicprogramming.jl_amd/csrc/back_pass_q4.hip"

// what the kernel source's host half refers to (never called here)
void ddp_set_error(const char *, ...) {}

namespace {
template <int MODE>
__global__ __launch_bounds__(64) void chain(int steps, double *out, long long *ticks, double lam)
{
    __shared__ double img[2][16 * 64], rec[5 * 64 * 2];
    const int lane = threadIdx.x, r = lane >> 4, c = lane & 3;
    constexpr bool LIMS = MODE >= 1, REG2 = MODE >= 2, REC = MODE >= 3, FETCH = MODE >= 4;
    // a stable pendulum-like linearisation (h = 0.01): x+ = A x + B u, running cost diag(10, 1, 2, 1) / 1
    Q4In in;
    in.fx = (r == c) ? (r == 3 ? 0.99 : 1.0) : ((c == r + 1) ? 0.01 : ((r == 3 && c == 0) ? 0.28 : 0.0));
    in.fu = (r == 1) ? 0.01 : (r == 3 ? 0.028 : 0.0);
    in.cx = 0.1 * (r + 1); in.cu = 0.05; in.u = 0.3;
    const double qd[4] = {10.0, 1.0, 2.0, 1.0};
    in.cxx = (r == c) ? qd[r] : 0.0; in.cxxT = in.cxx; in.cxuc = 0.0; in.cxur = 0.0; in.cuu = 1.0;
    for (int e = lane; e < 2 * 16 * 64; e += 64) (&img[0][0])[e] = 0.0;
    {   // the LDS image of the operands: ten values per lane and step, as the chunk kernels keep them
        const double v[10] = {in.fx, in.fu, in.cx, in.cu, in.u, in.cxx, in.cxxT, in.cxuc, in.cxur, in.cuu};
        for (int k = 0; k < 10; ++k) { img[0][64 * k + lane] = v[k]; img[1][64 * k + lane] = v[k]; }
    }
    __syncthreads();
    Q4Par par;
    par.lam = lam; par.limlo = -5.0; par.limhi = 5.0; par.nolims = false; par.ieta = 1.0;
    Q4State s;
    s.V = in.cxx; s.VT = in.cxxT; s.vxc = in.cx; s.kprev = 0.0; s.dV0 = 0.0; s.dV1 = 0.0; s.diverge = 0;
    Q4Out prev;
    prev.Vn = s.V; prev.Kc = 0.0; prev.vx = in.cx; prev.kk = 0.0; prev.Quu = 1.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = steps - 1; i >= 0; --i) {
        Q4In o = in, nx = in;
        if (FETCH) {                                           // the next step's operands: ten LDS reads in flight under this step's products
            const double *p = img[i & 1] + lane;
            nx.fx = p[0]; nx.fu = p[64]; nx.cx = p[128]; nx.cu = p[192]; nx.u = p[256]; nx.cxx = p[320]; nx.cxxT = p[384]; nx.cxuc = p[448]; nx.cxur = p[512]; nx.cuu = p[576];
        }
        // the gradient moves from step to step as along a real trajectory: with stationary operands the warm start k_{i+1} IS the solution of
        // step i, its gradient is below minGrad, and the two-iteration form hands every step to the generic loop (exit 5) — not the
        // production path (the first version of this file measured exactly that: 646 ns for mode 1)
        o.cu += 0.013 * (double)(i & 3); o.cx += 0.007 * (double)((i >> 1) & 3);
        // nothing of a step is loop-invariant in production (time-varying Jacobians): keep the compiler from hoisting products of constants
        asm volatile("" : "+v"(o.fx), "+v"(o.fu), "+v"(o.cx), "+v"(o.cu), "+v"(o.u));
        asm volatile("" : "+v"(o.cxx), "+v"(o.cxxT), "+v"(o.cxuc), "+v"(o.cxur), "+v"(o.cuu));
        Q4Out res;
        if (REC) {
            auto mid = [&]() __attribute__((always_inline)) {
                double *q = rec + 5 * 64 * (i & 1) + lane;
                q[0] = prev.Vn; q[64] = prev.Kc; q[128] = prev.vx; q[192] = prev.kk; q[256] = prev.Quu;
            };
            q4_step<LIMS, REG2, 0, decltype(mid), false>(i, o, s, res, par, mid);
        } else {
            q4_step<LIMS, REG2, 0, Q4NoMid, false>(i, o, s, res, par);
        }
        prev = res;
        if (FETCH) in = nx;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[lane] = s.V + s.vxc + s.dV0 + s.dV1 + prev.Kc + prev.kk + (REC ? rec[lane] : 0.0) + (double)s.diverge;
    if (lane == 0) ticks[0] = t1 - t0;
}

template <int MODE> void run(const char *name)
{
    const int steps = 200000;
    double *d; long long *t, ht;
    (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&t, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    chain<MODE><<<1, 64>>>(2000, d, t, 1.0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); chain<MODE><<<1, 64>>>(steps, d, t, 1.0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
    double h0; (void)hipMemcpy(&h0, d, 8, hipMemcpyDeviceToHost);
    printf("%-72s %7.1f ns / step  %7.1f ticks / step   (check %.6g)\n", name, ms * 1e6 / steps, (double)ht / steps, h0);
    (void)hipFree(d); (void)hipFree(t);
}
}   // namespace

int main()
{
    run<0>("mode 0: 12 products (4x4x4_4b) + scalar gain + value update");
    run<1>("mode 1: + control limits (straight-line two-iteration box-QP, m = 1)");
    run<2>("mode 2: + regType 2 (three more products)");
    run<3>("mode 3: + step record (five LDS writes of the previous step behind the products)");
    run<4>("mode 4: + operand fetch of the next step (ten LDS reads)");
    printf("production back_pass_q4c<LIMS, REG2, CH = 8> at C3 (N = 600, B = 4 096, one wave per SIMD): 0.389 ms / 599 steps = 650 ns per step\n");
    return 0;
}
