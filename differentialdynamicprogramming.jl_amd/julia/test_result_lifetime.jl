# For whoever has Julia + an MI355X (the build image has neither Julia nor a GPU; tests/test_julia_binding.py checks this binding
# statically against include/ddp_amd.h).  What it guards: results that reach the caller with singleton axes dropped must not share a
# pinned block with a wrapper that can be collected first (Julia >= 1.11: reshape / dropdims share the Memory, not the wrapper) —
# DDPAmd.result_pair.  forward_pass twice at the same size with a GC in between: the first result must still be what it was.
#   julia --project=. differentialdynamicprogramming.jl_amd/julia/test_result_lifetime.jl
include(joinpath(@__DIR__, "DDPAmd.jl"))
using .DDPAmd, LinearAlgebra, Random, Test

Random.seed!(1)
n, m, N = 10, 2, 20_000                      # 10 * 20 000 * 8 B = 1.6 MB: above the 1 MB threshold of the pinned result cache
h = 0.01
A0 = randn(n, n); A = exp(h * (A0 - A0')); B = h * randn(n, m)
prob = DDPAmd.LQProblem(A, B, h * Matrix(I, n, n), 0.1h * Matrix(I, m, m))
x0a, x0b = ones(n), 2 .* ones(n)
u0 = 0.1 .* randn(m, N)
xa, ua, ca = DDPAmd.forward_pass(DDPAmd.GaussianPolicy(Float64), x0a, u0, [], 1.0, prob, [])      # unbatched, scalar α: both axes dropped
keep = copy(xa)
GC.gc(); GC.gc()
xb, ub, cb = DDPAmd.forward_pass(DDPAmd.GaussianPolicy(Float64), x0b, u0, [], 1.0, prob, [])      # same size: same cache block if it was freed
@test size(xa) == (n, N) && size(xb) == (n, N)
@test xa == keep                              # the first result was not recycled under the caller
@test xb[:, 1] == x0b && xa[:, 1] == x0a
println("result lifetime ok")
