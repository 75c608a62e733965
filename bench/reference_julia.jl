#!/usr/bin/env julia
# reference_julia.jl — the CPU column of BASELINE.md measured on the REAL reference: iLQG passes per second
# (1 pass = one back_pass over N-1 steps + one forward_pass rollout, one α) of DifferentialDynamicProgramming.jl v0.5.0 on
# BASELINE config 1 (demo_linear: n = 10, m = 2, T = 1000, LTI, no control limits, regType 1).
#
#     julia --project=/path/to/DifferentialDynamicProgramming.jl bench/reference_julia.jl            # one core
#     julia -t auto --project=... bench/reference_julia.jl                                           # + all cores
#
# NOT EXECUTED in this repository's build image (no Julia toolchain); bench.py's `cpu_baseline` is the C restatement
# (kind "port").  Put the line this prints next to it (BASELINE.md).  The workload mirrors bench.py::make_workload (same
# distributions; Julia's RNG stream differs from NumPy's, which does not matter for a timing).
using LinearAlgebra, Random, Printf
using DifferentialDynamicProgramming
const DDP = DifferentialDynamicProgramming

function make_problem(rng; n=10, m=2, T=1000, h=0.01)
    A0 = randn(rng, n, n)
    A = exp(h * (A0 - A0'))
    B = h * randn(rng, n, m)
    Q = h * Matrix{Float64}(I, n, n)
    R = 0.1h * Matrix{Float64}(I, m, m)
    x0 = ones(n) + 0.1 * randn(rng, n)
    u0 = 0.1 * randn(rng, m, T)
    f(x, u, i) = A * x + B * u
    costfun(x, u) = 0.5 * sum(x .* (Q * x)) + 0.5 * sum(u .* (R * u))
    return (A=A, B=B, Q=Q, R=R, x0=x0, u0=u0, f=f, costfun=costfun, cxu=zeros(n, m))
end

# one pass on a nominal trajectory: STEP 2 + one rollout of STEP 3 of src/iLQG.jl:235-281
function one_pass(P, x, u)
    cx = P.Q * x
    cu = P.R * u
    diverge, traj, Vx, Vxx, dV = DDP.back_pass(cx, cu, P.Q, P.cxu, P.R, P.A, P.B, 1.0, 1, [], x, u)
    xnew, unew, cnew = DDP.forward_pass(traj, P.x0, u, x, 1.0, P.f, P.costfun, [], -)
    return diverge, cnew
end

function nominal(P)
    empty = GaussianPolicy(Float64)
    x, u, c = DDP.forward_pass(empty, P.x0, P.u0, [], 1, P.f, P.costfun, [], -)
    return x, u
end

function main()
    BLAS.set_num_threads(1)                       # the matrices are 10x10: BLAS threads only add overhead
    rng = MersenneTwister(1234)
    P = make_problem(rng)
    x, u = nominal(P)
    one_pass(P, x, u)                             # compile
    reps = 0
    t0 = time()
    while time() - t0 < 10.0
        one_pass(P, x, u)
        reps += 1
    end
    t1 = time() - t0
    @printf("{\"kind\": \"reference\", \"julia\": \"%s\", \"cores\": 1, \"passes\": %d, \"seconds\": %.2f, \"value\": %.1f, \"unit\": \"iterations/s\"}\n",
            string(VERSION), reps, t1, reps / t1)
    nt = Threads.nthreads()
    if nt > 1                                     # B independent problems, one per thread (the reference itself is single-threaded)
        Ps = [make_problem(MersenneTwister(100 + i)) for i in 1:nt]
        xs = [nominal(p) for p in Ps]
        counts = zeros(Int, nt)
        t0 = time()
        Threads.@threads for i in 1:nt
            while time() - t0 < 10.0
                one_pass(Ps[i], xs[i][1], xs[i][2])
                counts[i] += 1
            end
        end
        t1 = time() - t0
        @printf("{\"kind\": \"reference\", \"julia\": \"%s\", \"cores\": %d, \"passes\": %d, \"seconds\": %.2f, \"value\": %.1f, \"unit\": \"iterations/s\"}\n",
                string(VERSION), nt, sum(counts), t1, sum(counts) / t1)
    end
end

main()
