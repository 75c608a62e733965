// pend_math.h — sin and cos for the pendulum-on-a-cart rollouts (src/system_pendcart.jl:83-89 evaluates sin(θ), cos(θ) once per step;
// a rollout is a chain of dependent steps, so the ~70 vector instructions of the library's sincos are a third of a step).
// Included inside each file's anonymous namespace.
#pragma once

// |x| < 1e9: argument reduction by π/2 in three parts with fused multiply-adds (each product n·P is exact inside the fma, so the only
// rounding errors are those of the three partial results, relative to THEIR size: no cancellation near multiples of π/2, where the
// pendulum's goal θ = π sits), then the minimax kernels of fdlibm's k_sin.c / k_cos.c on [-π/4, π/4] (the published coefficients;
// the low word of the reduced argument enters to first order).  ~42 instructions, no branch.
// d = a·b + c as the three-address instruction: left to itself the compiler picks the two-address v_fmac_f64 and, because the polynomial
// coefficients stay live across the steps, copies each of them into the accumulator first (ten extra moves per sincos)
__device__ __forceinline__ double pend_fma3(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// the polynomial coefficients in vector registers, materialised ONCE per kernel (set up before the time loop)
struct PendTrig {
    double s1, s2, s3, s4, s5, s6, c1, c2, c3, c4, c5, c6;
    __device__ __forceinline__ void init()
    {
        s1 = -1.66666666666666324348e-01; s2 = 8.33333333332248946124e-03; s3 = -1.98412698298579493134e-04;
        s4 = 2.75573137070700676789e-06; s5 = -2.50507602534068634195e-08; s6 = 1.58969099521155010221e-10;
        c1 = 4.16666666666666019037e-02; c2 = -1.38888888888741095749e-03; c3 = 2.48015872894767294178e-05;
        c4 = -2.75573143513906633035e-07; c5 = 2.08757232129817482790e-09; c6 = -1.13596475577881948265e-11;
        asm volatile("" : "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6));
        asm volatile("" : "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6));
    }
};

__device__ __forceinline__ void pend_sincos_small(const PendTrig &t, double x, double &sn, double &cs)
{
    constexpr double TWO_OVER_PI = 0x1.45f306dc9c883p-1;
    constexpr double P1 = 0x1.921fb54442d18p+0, P2 = 0x1.1a62633145c07p-54, P3 = -0x1.f1976b7ed8fbcp-110;      // π/2 = P1 + P2 + P3 + O(1e-49)
    const double nd = __builtin_rint(x * TWO_OVER_PI);
    const double r0 = __builtin_fma(-nd, P1, x);
    const double r = __builtin_fma(-nd, P2, r0);
    // the low word of the reduced argument: the rounding error of the line above (r0 - r is exact) and the third part of π/2
    const double y = __builtin_fma(-nd, P3, __builtin_fma(-nd, P2, r0 - r));
    const int q = (int)nd;
    const double z = r * r;
    // sin(r) = r + r z (S1 + z (S2 + ... ))
    double ps = pend_fma3(z, t.s6, t.s5);
    ps = pend_fma3(z, ps, t.s4);
    ps = pend_fma3(z, ps, t.s3);
    ps = pend_fma3(z, ps, t.s2);
    ps = pend_fma3(z, ps, t.s1);
    double sk = __builtin_fma(r * z, ps, r);
    // cos(r) = w + (((1 - w) - z/2) + z² (C1 + z (C2 + ...))),  w = 1 - z/2
    double pc = pend_fma3(z, t.c6, t.c5);
    pc = pend_fma3(z, pc, t.c4);
    pc = pend_fma3(z, pc, t.c3);
    pc = pend_fma3(z, pc, t.c2);
    pc = pend_fma3(z, pc, t.c1);
    const double hz = 0.5 * z, w = 1.0 - hz;
    double ck = w + (((1.0 - w) - hz) + ((z * z) * pc - r * y));         // cos(r + y) = cos r - y sin r
    sk = __builtin_fma(y, w, sk);                                          // sin(r + y) = sin r + y cos r
    // quadrant: sin(r + q π/2), cos(r + q π/2)
    const bool swap = (q & 1) != 0;
    double s = swap ? ck : sk, c = swap ? sk : ck;
    const unsigned sflip = ((unsigned)q & 2u) << 30, cflip = (((unsigned)q + 1u) & 2u) << 30;
    sn = __hiloint2double(__double2hiint(s) ^ (int)sflip, __double2loint(s));
    cs = __hiloint2double(__double2hiint(c) ^ (int)cflip, __double2loint(c));
}

// sin(x), cos(x): the short path when every lane's argument is below 1e9 in magnitude (NaN and Inf come out as NaN either way), the
// library's otherwise — one wave-uniform branch, practically never taken (θ of a rollout that has not diverged stays within a few turns)
__device__ __forceinline__ void pend_sincos(const PendTrig &t, double x, double &sn, double &cs)
{
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(__builtin_fabs(x) >= 1e9 && __builtin_fabs(x) < __builtin_inf()) != 0, 0)) {
        sincos(x, &sn, &cs);
    } else {
        pend_sincos_small(t, x, sn, cs);
    }
}
