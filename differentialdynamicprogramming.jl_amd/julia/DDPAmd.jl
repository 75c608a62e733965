# DDPAmd.jl — `@ccall` binding of libddp_amd.so (include/ddp_amd.h) that keeps the reference's call signatures for the hot path
# and adds what the GPU needs to pay off: a BATCH axis and DEVICE-RESIDENT operands.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia toolchain.  Every `@ccall` below is checked against the
# header by tests/test_julia_binding.py (name, arity, argument and return types, struct layouts) and mirrored one-to-one by
# the ctypes host in ../__init__.py, which is what the GPU tests drive.  INTEGRATION.md shows how a maintainer of
# DifferentialDynamicProgramming.jl wires it in.
#
#   drop-in entry points (one trajectory = the reference's arrays unchanged; a trailing batch axis solves B problems at once)
#     DDPAmd.iLQG(f, costfun, df, x0, u0; lims, α, tol_fun, ...)          ↔ src/iLQG.jl:143-341
#         f a registered problem (LQProblem / PendcartProblem are callable like `f`)  → device-resident driver
#         f any other closure  → the reference's own loop with `back_pass` (STEP 2, iLQG.jl:235-251) on the GPU (`install!`)
#     DDPAmd.back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u)        ↔ src/backward_pass.jl:162-252
#     DDPAmd.boxQP(H,g,lower,upper,x0)                                    ↔ src/boxQP.jl:29-188
#     DDPAmd.forward_pass(traj_new,x0,u,x,α,problem,lims)                 ↔ src/forward_pass.jl:9-33
#     DDPAmd.∇kl / back_pass_gps / forward_covariance / kl_div_wiki       ↔ src/klutils.jl, backward_pass.jl:259-350, forward_pass.jl:37-56
#     DDPAmd.iLQGkl(problem, x0, traj_prev, fx_model, R1; ...)            ↔ src/iLQGkl.jl:25-178 (single KL constraint, one library call)
#   device-resident (`DevArray`, or any array type whose `pointer` is a device pointer, e.g. AMDGPU.ROCArray)
#     back_pass_dev!, forward_pass_dev!, df_dev!, iLQG_dev!
module DDPAmd

using LinearAlgebra

const libddp = get(ENV, "DDP_AMD_LIB", joinpath(@__DIR__, "..", "libddp_amd.so"))

# ---- C structs (field order = include/ddp_amd.h) -------------------------------------------------
struct BPDesc
    n::Cint
    m::Cint
    N::Cint
    B::Cint
    fx_tv::Cint
    fx_batched::Cint
    cost_tv::Cint
    cost_batched::Cint
    regType::Cint
    has_lims::Cint
end

struct QPOpts
    maxIter::Cint
    minGrad::Cdouble
    minRelImprove::Cdouble
    stepDec::Cdouble
    minStep::Cdouble
    Armijo::Cdouble
end

struct CProblem
    kind::Cint
    n::Cint
    m::Cint
    N::Cint
    B::Cint
    A::Ptr{Float64}
    Bm::Ptr{Float64}
    dyn_tv::Cint
    dyn_batched::Cint
    Q::Ptr{Float64}
    R::Ptr{Float64}
    g::Cdouble
    l::Cdouble
    h::Cdouble
    d::Cdouble
    goal::NTuple{4,Cdouble}
    cost_diag::Cint
    diff_wrap::Cuint
end

struct ILQGOpts
    lambda::Cdouble
    dlambda::Cdouble
    lambda_factor::Cdouble
    lambda_max::Cdouble
    lambda_min::Cdouble
    tol_fun::Cdouble
    tol_grad::Cdouble
    max_iter::Cint
    regType::Cint
    reduce_ratio_min::Cdouble
    n_alpha::Cint
    alpha::NTuple{16,Cdouble}
end

struct KLCostTerms
    cx::Ptr{Float64}
    cu::Ptr{Float64}
    cxx::Ptr{Float64}
    cxu::Ptr{Float64}
    cuu::Ptr{Float64}
    eta::Ptr{Float64}
    eta_tv::Cint
end

struct ILQGKLOpts
    kl_step::Cdouble
    max_iter::Cint
    etabracket::NTuple{3,Cdouble}
    del0::Cdouble
end

# ddp_kl_dual (include/ddp_amd.h): device pointers
struct KLDual
    etab::Ptr{Float64}
    eta::Ptr{Float64}
    del::Ptr{Float64}
    divergence::Ptr{Float64}
    satisfied::Ptr{Int32}
    status::Ptr{Int32}
    live::Ptr{Int32}
    pend::Ptr{Int32}
    iters::Ptr{Int32}
    nback::Ptr{Int32}
end

const NULLF = Ptr{Float64}(C_NULL)
const NULLI = Ptr{Int32}(C_NULL)

# ---- handle ---------------------------------------------------------------------------------------
last_error() = unsafe_string(@ccall libddp.ddp_last_error()::Cstring)
"a call the library refused or that failed on the device (the Python mirror's `DDPError`); `rc` is the C return code"
struct DDPError <: Exception
    rc::Int
    msg::String
end
Base.showerror(io::IO, e::DDPError) = print(io, "libddp_amd: ", e.msg, " (rc=", e.rc, ")")
check(rc) = rc == 0 ? nothing : throw(DDPError(Int(rc), last_error()))   # there is no CPU fallback inside the library

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(device::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(@ccall libddp.ddp_create(device::Cint, r::Ptr{Ptr{Cvoid}})::Cint)
        h = new(r[])
        finalizer(hh -> (@ccall libddp.ddp_destroy(hh.ptr::Ptr{Cvoid})::Cint), h)
        return h
    end
end

const _default = Ref{Union{Nothing,Handle}}(nothing)
default_handle() = (_default[] === nothing && (_default[] = Handle(0)); _default[]::Handle)
sync(h::Handle=default_handle()) = check(@ccall libddp.ddp_sync(h.ptr::Ptr{Cvoid})::Cint)
device_count() = Int(@ccall libddp.ddp_device_count()::Cint)
"re-read the DDP_* switches (kernel choice for A/B timing / tests) for this handle; they are read once, in ddp_create"
reload_env(h::Handle=default_handle()) = check(@ccall libddp.ddp_reload_env(h.ptr::Ptr{Cvoid})::Cint)
"kernel of the last back_pass (0) / forward_pass (1) dispatch of the handle (debug query)"
last_kernel(which::Integer=0; handle::Handle=default_handle()) = unsafe_string(@ccall libddp.ddp_last_kernel(handle.ptr::Ptr{Cvoid}, which::Cint)::Cstring)
# tiles of the shared-operand backward pass that gave their trajectories to the per-trajectory kernels after a timed-out wait (0 in a healthy run)
sh_timeouts(; handle::Handle=default_handle()) = Int(@ccall libddp.ddp_sh_timeouts(handle.ptr::Ptr{Cvoid})::Cint)
"what those tiles waited for: rows {work-group, group, chunk, progress word seen, ms waited, XCD, groups, launch} + the 16 progress words now"
function sh_timeout_info(; handle::Handle=default_handle())
    buf = zeros(Cint, 80)
    nrec = @ccall libddp.ddp_sh_timeout_info(handle.ptr::Ptr{Cvoid}, buf::Ptr{Cint}, 80::Cint)::Cint
    nrec < 0 && check(nrec)
    return permutedims(reshape(buf[1:64], 8, 8))[1:nrec, :], buf[65:80]
end

"""
    result_array(dims...) -> Array{Float64}

Array for a RESULT of a host-pointer call, in page-locked memory from the library's cache (`ddp_host_alloc`) when it is large (>= 1 MB;
`ENV["DDP_PINNED_RESULTS"] = "0"`: plain `zeros`).  The device writes straight into it at link speed; a finalizer hands the block back to
the cache, so the next call of the same size pays neither pinning nor first-touch page faults (1.2 GB of results per C2 pass).
"""
function result_array(dims::Integer...)
    d = map(Int, dims)
    nbytes = prod(d) * sizeof(Float64)
    (nbytes < (1 << 20) || get(ENV, "DDP_PINNED_RESULTS", "1") == "0") && return zeros(d...)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(@ccall libddp.ddp_host_alloc(nbytes::Csize_t, r::Ptr{Ptr{Cvoid}})::Cint)
    a = unsafe_wrap(Array, Ptr{Float64}(r[]), d; own=false)
    finalizer(x -> (@ccall libddp.ddp_host_free(pointer(x)::Ptr{Cvoid})::Cint), a)
    return a
end

"""
    result_pair(work_dims, final_dims) -> (work, final)

A result that is handed to the C call with one shape (`work`, batch / step-size axes included) and returned to the caller with another
(`final`, singleton axes dropped).  `reshape` / `dropdims` of a pinned `result_array` must NOT be used for that: from Julia 1.11 on a
reshaped `Array` shares the `Memory`, not the wrapper object the finalizer sits on, so the wrapper could be collected — and the block
handed back to the library's cache and overwritten by the next call of the same size — while the caller still holds the reshaped
result.  Here both shapes are wrappers of the same block made at allocation time; the block is freed when the LAST of them is
collected (a shared atomic count).  Small results are ordinary GC memory, where `reshape` is safe.
"""
function result_pair(work::Tuple, final::Tuple)
    w = map(Int, work); f = map(Int, final)
    prod(w) == prod(f) || error("result_pair: shapes differ in length")
    nbytes = prod(w) * sizeof(Float64)
    if nbytes < (1 << 20) || get(ENV, "DDP_PINNED_RESULTS", "1") == "0"
        z = zeros(w...)
        return z, (w == f ? z : reshape(z, f))
    end
    w == f && (a = result_array(w...); return a, a)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(@ccall libddp.ddp_host_alloc(nbytes::Csize_t, r::Ptr{Ptr{Cvoid}})::Cint)
    ptr = r[]
    live = Threads.Atomic{Int}(2)
    release = _ -> (Threads.atomic_sub!(live, 1) == 1 && (@ccall libddp.ddp_host_free(ptr::Ptr{Cvoid})::Cint); nothing)
    a = unsafe_wrap(Array, Ptr{Float64}(ptr), w; own=false)
    b = unsafe_wrap(Array, Ptr{Float64}(ptr), f; own=false)
    finalizer(release, a); finalizer(release, b)
    return a, b
end

# ---- output container (src/iLQG.jl:39-53); with a batch the arrays carry a trailing axis ------------
mutable struct GaussianPolicy{P}
    T::Int
    n::Int
    m::Int
    K::Array{P}
    k::Array{P}
    Σ::Array{P}
    Σi::Array{P}
end
GaussianPolicy(P::Type) = GaussianPolicy{P}(0, 0, 0, zeros(P, 0, 0, 0), zeros(P, 0, 0), zeros(P, 0, 0, 0), zeros(P, 0, 0, 0))
Base.isempty(p::GaussianPolicy) = p.T == p.n == p.m == 0
Base.length(p::GaussianPolicy) = p.T
_isempty_policy(p) = p === nothing || (p.T == 0 && p.n == 0 && p.m == 0)

# ---- registered problem families: callable like the closure `f(x,u,i)`, so `iLQG(problem, costfun, df, x0, u0)` dispatches ----
abstract type RegisteredProblem end

"x⁺ = A x + B u, cost ½Σx∘Qx + ½Σu∘Ru (src/demo_linear.jl:30-50); A,B: [n,n]/[n,m], [..,N] (LTV), [..,B] / [..,N,B] with `dyn_batched`"
struct LQProblem <: RegisteredProblem
    A::Array{Float64}
    B::Array{Float64}
    Q::Matrix{Float64}
    R::Matrix{Float64}
    dyn_batched::Bool
end
LQProblem(A, B, Q, R; dyn_batched::Bool=false) = LQProblem(Array{Float64}(A), Array{Float64}(B), Matrix{Float64}(Q), Matrix{Float64}(R), dyn_batched)
(p::LQProblem)(x, u, i) = (ndims(p.A) == 2 ? p.A : view(p.A, :, :, i)) * x + (ndims(p.B) == 2 ? p.B : view(p.B, :, :, i)) * u

"pendulum on a cart (src/system_pendcart.jl:42-59,83-106): explicit Euler step, quadratic cost of length N+1"
Base.@kwdef struct PendcartProblem <: RegisteredProblem
    g::Float64 = 9.82
    l::Float64 = 0.35
    h::Float64 = 0.01
    d::Float64 = 0.99
    Q::Matrix{Float64} = Matrix(Diagonal([10.0, 1, 2, 1]))
    R::Matrix{Float64} = fill(1.0, 1, 1)
    goal::Vector{Float64} = [π, 0, 0, 0]
end
(p::PendcartProblem)(x, u, i) = [x[1] + p.h * x[2], x[2] + p.h * (-p.g / p.l * sin(x[1]) + u[1] / p.l * cos(x[1]) - p.d * x[2]),
                                 x[3] + p.h * x[4], x[4] + p.h * u[1]]

dims(p::LQProblem) = (size(p.A, 1), size(p.B, 2))
dims(::PendcartProblem) = (4, 1)
cost_len(::LQProblem, N) = N
cost_len(::PendcartProblem, N) = N + 1
dyn_tv(p::LQProblem) = (ndims(p.A) - (p.dyn_batched ? 1 : 0)) == 3

"""
    WrappedDiff(coords...)

What stands in for a user `diff_fun` (src/forward_pass.jl:5,19; iLQG.jl:156) on the device: subtraction with the listed state coordinates
(1-based) wrapped to [-π, π], `rem2pi(a[j] - b[j], RoundNearest)` — `ddp_problem::diff_wrap`.  It is callable, so the same object can be
handed to the reference's own `forward_pass`.  A Julia closure cannot cross the C ABI: any `diff_fun` other than `-` and a `WrappedDiff`
is refused.
"""
struct WrappedDiff
    coords::Vector{Int}
    WrappedDiff(coords::Integer...) = (all(c -> 1 <= c <= 32, coords) || error("WrappedDiff: coordinates must be in 1..32"); new(collect(Int, coords)))
end
(w::WrappedDiff)(a, b) = (d = a - b; for c in w.coords; d[c] = rem2pi(d[c], RoundNearest); end; d)
_diff_mask(::typeof(-), n) = Cuint(0)
_diff_mask(w::WrappedDiff, n) = (all(c -> c <= n, w.coords) || error("WrappedDiff names a coordinate beyond the state length $n");
                                 reduce(|, (Cuint(1) << (c - 1) for c in w.coords); init=Cuint(0)))
_diff_mask(f, n) = error("diff_fun must be `-` or a DDPAmd.WrappedDiff: a Julia closure cannot run on the device")

# C view of a problem; `A`, `Bm`, `Q`, `R` are host or device pointers depending on the entry point it is passed to
cproblem(p::LQProblem, N, B; A=pointer(p.A), Bm=pointer(p.B), Q=pointer(p.Q), R=pointer(p.R), diff=-) =
    CProblem(0, size(p.A, 1), size(p.B, 2), N, B, A, Bm, dyn_tv(p), p.dyn_batched, Q, R, 0.0, 0.0, 0.0, 0.0, (0.0, 0.0, 0.0, 0.0), isdiag(p.Q) && isdiag(p.R),
             _diff_mask(diff, size(p.A, 1)))
cproblem(p::PendcartProblem, N, B; A=NULLF, Bm=NULLF, Q=pointer(p.Q), R=pointer(p.R), diff=-) =
    CProblem(1, 4, 1, N, B, A, Bm, 0, 0, Q, R, p.g, p.l, p.h, p.d, (p.goal[1], p.goal[2], p.goal[3], p.goal[4]), isdiag(p.Q) && isdiag(p.R),
             _diff_mask(diff, 4))

_f64(a) = Array{Float64}(a)      # dense column-major copy (handles Diagonal cxx, Vector cuu of the demos)
_lims(lims) = (lims === nothing || isempty(lims)) ? Float64[] : _f64(lims)
_ptr_or_null(a::Array{Float64}) = isempty(a) ? NULLF : pointer(a)

# ======================================================================================= host-array API
"""
    back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u) -> diverge, GaussianPolicy, Vx, Vxx, dV

Same signature, dispatch (array rank selects LTI / LTV / time-varying cost) and return values as the reference's linear-system
`back_pass` methods (src/backward_pass.jl:162,179,217).  `cx[n,N,B]` solves a batch: `λ` may then be a vector, `fx`/`cxx` of
rank 4 are per trajectory and time-varying, every output carries the batch axis and `diverge` is a vector.  Per-trajectory
TIME-INVARIANT operands (`fx[n,n,B]`, `cxx[n,n,B]`) have the rank of the reference's time-varying ones: say so with
`batched_dynamics=true` / `batched_cost=true` (the last axis is then the batch, not time).
"""
function back_pass(cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u; handle::Handle=default_handle(), policy=GaussianPolicy{Float64},
                   batched_dynamics::Union{Nothing,Bool}=nothing, batched_cost::Union{Nothing,Bool}=nothing)
    batched = ndims(cx) == 3
    n, N = size(cx, 1), size(cx, 2)
    m = size(cu, 1)
    B = batched ? size(cx, 3) : 1
    cx, cu, cxx, cxu, fx, fu, u = map(_f64, (cx, cu, cxx, cxu, fx, fu, u))
    cuu = reshape(_f64(cuu), m, m, size(_f64(cuu))[3:end]...)
    # rank 4 is per trajectory AND time-varying; a rank-3 `fx` is read as time-varying (the reference's :162 method) unless the caller
    # says `batched_dynamics=true` (per-trajectory, time-invariant: `fx[n,n,B]`) — no guess from `size(fx,3) == B` (wrong when B == N)
    fx_batched = batched_dynamics === nothing ? ndims(fx) == 4 : batched_dynamics
    cost_batched = batched_cost === nothing ? ndims(cxx) == 4 : batched_cost
    (fx_batched || cost_batched) && !batched && error("batched_dynamics / batched_cost need a batch: cx[n,N,B]")
    fx_tv = (ndims(fx) - (fx_batched ? 1 : 0)) == 3
    cost_tv = (ndims(cxx) - (cost_batched ? 1 : 0)) == 3
    @assert size(cu) == (m, N, (batched ? (B,) : ())...) "size(cu) should be (m, N)"
    @assert size(fx)[1:2] == (n, n) && size(fu)[1:2] == (n, m) "size(fx), size(fu)"
    @assert size(cxx)[1:2] == (n, n) "size(cxx) should be (n, n)"
    @assert size(cxu)[1:2] == (n, m) "size(cxu) should be (n, m)"
    limsp = _lims(lims)
    has_lims = !isempty(limsp)
    d = BPDesc(n, m, N, B, fx_tv, fx_batched, cost_tv, cost_batched, regType, has_lims)
    bt = batched ? (B,) : ()
    K = result_array(m, n, N, bt...); k = result_array(m, N, bt...); Quu = result_array(m, m, N, bt...)
    Vx = result_array(n, N, bt...); Vxx = result_array(n, n, N, bt...); dV = zeros(2, bt...)
    diverge = zeros(Int32, B)
    lam = λ isa Number ? fill(Float64(λ), B) : _f64(λ)
    GC.@preserve cx cu cxx cxu cuu fx fu u lam limsp K k Quu Vx Vxx dV diverge begin
        check(@ccall libddp.ddp_back_pass_f64(handle.ptr::Ptr{Cvoid}, Ref(d)::Ptr{BPDesc},
            cx::Ptr{Float64}, cu::Ptr{Float64}, cxx::Ptr{Float64}, cxu::Ptr{Float64}, cuu::Ptr{Float64},
            fx::Ptr{Float64}, fu::Ptr{Float64}, lam::Ptr{Float64},
            _ptr_or_null(limsp)::Ptr{Float64}, (has_lims ? pointer(u) : NULLF)::Ptr{Float64},
            K::Ptr{Float64}, k::Ptr{Float64}, Quu::Ptr{Float64}, Vx::Ptr{Float64}, Vxx::Ptr{Float64},
            dV::Ptr{Float64}, diverge::Ptr{Int32})::Cint)
    end
    # Σ (= Quui) is never written by the reference's back_pass (`undef`, backward_pass.jl:231); zeros here
    pol = policy(N, n, m, K, k, zeros(m, m, N, bt...), Quu)
    return (batched ? Vector{Int}(diverge) : Int(diverge[1])), pol, Vx, Vxx, dV
end

"""
    boxQP(H,g,lower,upper,x0; maxIter=100, ...) -> x, result, Hfree, free, trace        (src/boxQP.jl:29-36)

`H[m,m,count]` with matrix-shaped `g, lower, upper, x0` solves `count` problems in one launch.
"""
function boxQP(H, g, lower, upper, x0::AbstractVecOrMat; maxIter=100, minGrad=1e-8, minRelImprove=1e-8,
               stepDec=0.6, minStep=1e-22, Armijo=0.1, print=0, handle::Handle=default_handle())
    batched = ndims(H) == 3
    m = size(H, 1)
    cnt = batched ? size(H, 3) : 1
    H, g, lower, upper, x0 = map(_f64, (H, g, lower, upper, x0))
    x = zeros(m, cnt); Hf = zeros(m, m, cnt); res = zeros(Int32, cnt); fr = zeros(UInt8, m, cnt)
    opts = QPOpts(maxIter, minGrad, minRelImprove, stepDec, minStep, Armijo)
    GC.@preserve H g lower upper x0 x Hf res fr begin
        check(@ccall libddp.ddp_boxqp_f64(handle.ptr::Ptr{Cvoid}, m::Cint, cnt::Cint, H::Ptr{Float64}, g::Ptr{Float64},
            lower::Ptr{Float64}, upper::Ptr{Float64}, x0::Ptr{Float64}, Ref(opts)::Ptr{QPOpts},
            x::Ptr{Float64}, res::Ptr{Int32}, Hf::Ptr{Float64}, fr::Ptr{UInt8})::Cint)
    end
    if !batched
        free = BitVector(fr[:, 1] .!= 0); nf = count(free)
        return x[:, 1], Int(res[1]), UpperTriangular(Hf[1:nf, 1:nf, 1]), free, nothing
    end
    return x, Vector{Int}(res), Hf, fr .!= 0, nothing
end

"""
    demoQP(; n=500, kwargs...)

src/boxQP.jl:190-199: `H = M*M'` with `M = randn(n,n)`, `g = randn(n)`, bounds ±1, a random start, timed (one work-group per problem
for m > 8, csrc/boxqp_big.hip).
"""
function demoQP(; n=500, kwargs...)
    g = randn(n); M = randn(n, n); H = M * M'
    @time boxQP(H, g, -ones(n), ones(n), randn(n); kwargs...)
end

"""
    forward_pass(traj_new, x0, u, x, α, problem, lims) -> xnew, unew, cnew

`problem` (LQProblem / PendcartProblem) replaces the closures `f`, `costfun` of src/forward_pass.jl:9; `diff` is `-` or a `WrappedDiff`.
`u[m,N,B]` with `x0[n,B]` rolls a batch out; a vector `α` rolls all step sizes out concurrently (trailing α axis).
"""
function forward_pass(traj_new, x0, u, x, α, problem::RegisteredProblem, lims, diff=-; handle::Handle=default_handle())
    batched = ndims(u) == 3
    m, N = size(u, 1), size(u, 2)
    n = size(x0, 1)
    B = batched ? size(u, 3) : 1
    P = cproblem(problem, N, B; diff=diff)
    CL = cost_len(problem, N)
    empty = _isempty_policy(traj_new)
    al = α isa Number ? [Float64(α)] : _f64(α)
    na = length(al)
    # the shapes the caller gets: batch axis only when batched, step-size axis only for a vector α (forward_pass.jl:9-33 returns [n,N])
    fin(lead...) = (lead..., (batched ? (B,) : ())..., (α isa Number ? () : (na,))...)
    xnew, xnew_r = result_pair((n, N, B, na), fin(n, N)); unew, unew_r = result_pair((m, N, B, na), fin(m, N))
    cnew, cnew_r = result_pair((CL, B, na), fin(CL)); csum = zeros(B, na)
    x0 = _f64(x0); u = _f64(u)
    Kh = empty ? Float64[] : _f64(traj_new.K); kh = empty ? Float64[] : _f64(traj_new.k); xh = empty ? Float64[] : _f64(x)
    limsp = _lims(lims)
    GC.@preserve problem Kh kh xh x0 u al limsp xnew unew cnew csum begin
        check(@ccall libddp.ddp_forward_pass_f64(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, _ptr_or_null(Kh)::Ptr{Float64},
            _ptr_or_null(kh)::Ptr{Float64}, x0::Ptr{Float64}, u::Ptr{Float64}, _ptr_or_null(xh)::Ptr{Float64}, al::Ptr{Float64},
            na::Cint, _ptr_or_null(limsp)::Ptr{Float64},
            xnew::Ptr{Float64}, unew::Ptr{Float64}, cnew::Ptr{Float64}, csum::Ptr{Float64})::Cint)
    end
    return xnew_r, unew_r, cnew_r
end

"""
    df(problem, x, u) -> fx,fu,fxx,fxu,fuu,cx,cu,cxx,cxu,cuu         (the `df` closure of the registered families, iLQG.jl:225-229)
"""
function df(problem::RegisteredProblem, x, u; handle::Handle=default_handle())
    batched = ndims(u) == 3
    m, N = size(u, 1), size(u, 2)
    n = size(x, 1)
    B = batched ? size(u, 3) : 1
    P = cproblem(problem, N, B)
    x = _f64(x); u = _f64(u)
    bt = batched ? (B,) : ()
    cx = zeros(n, N, bt...); cu = zeros(m, N, bt...)
    pend = problem isa PendcartProblem
    fx = pend ? zeros(n, n, N, bt...) : Float64[]
    fu = pend ? zeros(n, m, N, bt...) : Float64[]
    GC.@preserve problem x u cx cu fx fu begin
        check(@ccall libddp.ddp_df_f64(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, x::Ptr{Float64}, u::Ptr{Float64},
            cx::Ptr{Float64}, cu::Ptr{Float64}, _ptr_or_null(fx)::Ptr{Float64}, _ptr_or_null(fu)::Ptr{Float64})::Cint)
    end
    if !pend
        fx, fu = problem.A, problem.B
    end
    return fx, fu, [], [], [], cx, cu, problem.Q, zeros(n, m), problem.R
end

const DEFAULT_ALPHA = exp10.(range(0, stop=-3, length=11))     # iLQG.jl:145

function _opts(α, tol_fun, tol_grad, max_iter, λ, dλ, λfactor, λmax, λmin, regType, reduce_ratio_min)
    length(α) <= 16 || error("at most 16 line-search step sizes are supported")
    al = ntuple(i -> i <= length(α) ? Float64(α[i]) : 0.0, 16)
    return ILQGOpts(λ, dλ, λfactor, λmax, λmin, tol_fun, tol_grad, max_iter, regType, reduce_ratio_min, length(α), al)
end

"""
    iLQG(problem, x0, u0; lims=[], α=..., tol_fun=1e-7, ...) -> x, u, traj_new, Vx, Vxx, cost, trace

Device-resident solve for a registered problem family; keyword arguments and defaults of src/iLQG.jl:143-163.
`u0[m,N,B]` with `x0[n,B]` solves B independent problems (own λ schedule, line search and termination each); `x0[n,N(,B)]` is a
PRE-ROLLED trajectory with `cost` (iLQG.jl:193-197).  Returns `nothing` when the initial controls diverge (one trajectory, :209).
`trace` is a Dict with the reference's trace keys per iteration (and trajectory) plus the per-trajectory summary `:stats`.
"""
function iLQG(problem::RegisteredProblem, x0, u0; lims=[], α=DEFAULT_ALPHA, tol_fun=1e-7, tol_grad=1e-4, max_iter=500, λ=1.0, dλ=1.0,
              λfactor=1.6, λmax=1e10, λmin=1e-6, regType=1, reduce_ratio_min=0, diff_fun=-, plot=1, verbosity=2, plot_fun=x -> 0,
              cost=[], traj_prev=0, print_head=10, handle::Handle=default_handle(), policy=GaussianPolicy{Float64})
    batched = ndims(u0) == 3
    m, N = size(u0, 1), size(u0, 2)
    n = size(x0, 1)
    B = batched ? size(u0, 3) : 1
    prerolled = size(x0, 2) == N && ndims(x0) == (batched ? 3 : 2) && N != 1
    if !prerolled && ndims(x0) == (batched ? 3 : 2) && !(ndims(x0) == 2 && batched)
        size(x0, 2) == 1 || error("pre-rolled initial trajectory must be of correct length (size(x0,2) == N)")   # iLQG.jl:199
    end
    P = cproblem(problem, N, B; diff=diff_fun)
    CL = cost_len(problem, N)
    o = _opts(α, tol_fun, tol_grad, max_iter, λ, dλ, λfactor, λmax, λmin, regType, reduce_ratio_min)
    bt = batched ? (B,) : ()
    x = result_array(n, N, bt...); u = result_array(m, N, bt...); K = result_array(m, n, N, bt...); k = result_array(m, N, bt...)
    Quu = result_array(m, m, N, bt...); Vx = result_array(n, N, bt...); Vxx = result_array(n, n, N, bt...); costo = result_array(CL, bt...)
    stats = zeros(8, B)
    cap = min(4max_iter + 64, 4096); tr7 = zeros(7, cap, B); git = Ref{Cint}(0)
    x0h = prerolled ? _f64(x0) : _f64(reshape(x0, n, B)); u0h = _f64(u0)
    c0 = (prerolled && !isempty(cost)) ? _f64(cost) : Float64[]
    limsp = _lims(lims)
    tcap = 4max_iter + 1000; timing = fill(NaN, tcap, 3)     # C layout [3, tcap]: column r of the Julia array is row r
    GC.@preserve problem x0h u0h c0 limsp timing x u K k Quu Vx Vxx costo stats tr7 begin
        check(@ccall libddp.ddp_ilqg_set_timing(handle.ptr::Ptr{Cvoid}, timing::Ptr{Float64}, tcap::Cint)::Cint)
        rc = @ccall libddp.ddp_ilqg_ex_f64(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, Ref(o)::Ptr{ILQGOpts},
            x0h::Ptr{Float64}, (prerolled ? 1 : 0)::Cint, u0h::Ptr{Float64}, _ptr_or_null(c0)::Ptr{Float64}, _ptr_or_null(limsp)::Ptr{Float64},
            x::Ptr{Float64}, u::Ptr{Float64}, K::Ptr{Float64}, k::Ptr{Float64}, Quu::Ptr{Float64}, Vx::Ptr{Float64}, Vxx::Ptr{Float64},
            costo::Ptr{Float64}, stats::Ptr{Float64}, cap::Cint, tr7::Ptr{Float64}, git::Ptr{Cint})::Cint
        @ccall libddp.ddp_ilqg_set_timing(handle.ptr::Ptr{Cvoid}, NULLF::Ptr{Float64}, 0::Cint)::Cint
        check(rc)
    end
    (!batched && stats[1, 1] == -1) && return nothing      # EXIT: Initial control sequence caused divergence (iLQG.jl:205-210)
    g = Int(git[])
    keys7 = (:λ, :dλ, :α, :improvement, :cost, :reduce_ratio, :grad_norm)                                   # iLQG.jl:257,325-330
    trace = Dict{Symbol,Any}(:stats => stats, :status => Int.(stats[1, :]), :iter => Int.(stats[2, :]), :global_iters => g,
                             :time_derivs => timing[1:g, 1], :time_backward => timing[1:g, 2], :time_forward => timing[1:g, 3])
    for (r, key) in enumerate(keys7)
        trace[key] = batched ? tr7[r, :, :] : tr7[r, 1:max(Int(stats[2, 1]) - 1, 0), 1]
    end
    return x, u, policy(N, n, m, K, k, zeros(m, m, N, bt...), Quu), Vx, Vxx, costo, trace
end

# ---- more problems than resident trajectories, and the closed loop ----------------------------------------------------
"""
    iLQG_queue(problem, x0[n,P], u0[m,N,P]; slots=0, lims=[], α, tol_fun, ...) -> x, u, traj_new, Vx, Vxx, cost, trace

`P` independent solves of `iLQG` through `slots` resident trajectories (`ddp_ilqg_queue_f64`): a slot whose solve has ended is flushed
and armed with the next problem ON THE DEVICE, in the global iteration in which it ended, instead of idling until the slowest
trajectory of a lock-step batch has ended (32 768 pendulum solves: 0.70 s against 1.43 s as eight batches of 4 096).  Every solve is the
solve `iLQG` performs at batch size `slots`.  `trace[:stats]` is `stats[8,P]`.
"""
function iLQG_queue(problem::RegisteredProblem, x0::AbstractMatrix, u0::AbstractArray{<:Real,3}; slots::Integer=0, lims=[], α=DEFAULT_ALPHA,
                    tol_fun=1e-7, tol_grad=1e-4, max_iter=500, λ=1.0, dλ=1.0, λfactor=1.6, λmax=1e10, λmin=1e-6, regType=1,
                    reduce_ratio_min=0, diff_fun=-, handle::Handle=default_handle(), policy=GaussianPolicy{Float64})
    m, N, P_ = size(u0); n = size(x0, 1)
    size(x0, 2) == P_ || error("iLQG_queue: x0[n,P], u0[m,N,P]")
    P = cproblem(problem, N, P_; diff=diff_fun)
    CL = cost_len(problem, N)
    o = _opts(α, tol_fun, tol_grad, max_iter, λ, dλ, λfactor, λmax, λmin, regType, reduce_ratio_min)
    x = result_array(n, N, P_); u = result_array(m, N, P_); K = result_array(m, n, N, P_); k = result_array(m, N, P_)
    Quu = result_array(m, m, N, P_); Vx = result_array(n, N, P_); Vxx = result_array(n, n, N, P_); costo = result_array(CL, P_)
    stats = zeros(8, P_); git = Ref{Cint}(0)
    x0h = _f64(x0); u0h = _f64(u0); limsp = _lims(lims)
    GC.@preserve problem x0h u0h limsp x u K k Quu Vx Vxx costo stats begin
        check(@ccall libddp.ddp_ilqg_queue_f64(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, Ref(o)::Ptr{ILQGOpts}, slots::Cint,
            x0h::Ptr{Float64}, u0h::Ptr{Float64}, _ptr_or_null(limsp)::Ptr{Float64},
            x::Ptr{Float64}, u::Ptr{Float64}, K::Ptr{Float64}, k::Ptr{Float64}, Quu::Ptr{Float64}, Vx::Ptr{Float64}, Vxx::Ptr{Float64},
            costo::Ptr{Float64}, stats::Ptr{Float64}, git::Ptr{Cint})::Cint)
    end
    trace = Dict{Symbol,Any}(:stats => stats, :status => Int.(stats[1, :]), :iter => Int.(stats[2, :]), :global_iters => Int(git[]))
    return x, u, policy(N, n, m, K, k, zeros(m, m, N, P_), Quu), Vx, Vxx, costo, trace
end

"""
    iLQG_mpc(problem, x0[n,B], u0[m,N,B], steps; zero_tail=false, lims=[], ...) -> xcl[n,steps+1,B], ucl[m,steps,B], stats[8,steps,B], xplan, uplan

Closed loop on the device (`ddp_ilqg_mpc_f64`): every trajectory is solved `steps` times; after each solve its first control is applied
(the model is the plant: the next initial state is `x[:,2]` of the solution), the control sequence is shifted by one step and the problem
is solved again without returning to the host — the receding-horizon use of the warm-start hook of the reference (iLQG.jl:193-197).
"""
function iLQG_mpc(problem::RegisteredProblem, x0::AbstractMatrix, u0::AbstractArray{<:Real,3}, steps::Integer; zero_tail::Bool=false, lims=[],
                  α=DEFAULT_ALPHA, tol_fun=1e-7, tol_grad=1e-4, max_iter=500, λ=1.0, dλ=1.0, λfactor=1.6, λmax=1e10, λmin=1e-6, regType=1,
                  reduce_ratio_min=0, diff_fun=-, handle::Handle=default_handle())
    m, N, B = size(u0); n = size(x0, 1)
    size(x0, 2) == B || error("iLQG_mpc: x0[n,B], u0[m,N,B]")
    P = cproblem(problem, N, B; diff=diff_fun)
    o = _opts(α, tol_fun, tol_grad, max_iter, λ, dλ, λfactor, λmax, λmin, regType, reduce_ratio_min)
    xcl = zeros(n, steps + 1, B); ucl = zeros(m, steps, B); scl = zeros(8, steps, B)
    x = result_array(n, N, B); u = result_array(m, N, B); git = Ref{Cint}(0)
    x0h = _f64(x0); u0h = _f64(u0); limsp = _lims(lims)
    GC.@preserve problem x0h u0h limsp xcl ucl scl x u begin
        check(@ccall libddp.ddp_ilqg_mpc_f64(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, Ref(o)::Ptr{ILQGOpts}, steps::Cint,
            (zero_tail ? 1 : 0)::Cint, x0h::Ptr{Float64}, u0h::Ptr{Float64}, _ptr_or_null(limsp)::Ptr{Float64},
            xcl::Ptr{Float64}, ucl::Ptr{Float64}, scl::Ptr{Float64}, x::Ptr{Float64}, u::Ptr{Float64}, git::Ptr{Cint})::Cint)
    end
    return xcl, ucl, scl, x, u
end

# ---- the drop-in entry point ------------------------------------------------------------------------------------------
const _ref = Ref{Any}(nothing)         # the reference module once `install!` has added the GPU methods to its back_pass
const _installed = Method[]            # the methods `install!` added (what `uninstall!` deletes)
const MAX_N = 64                       # include/ddp_amd.h: the backward kernels take n <= 64, m <= 8
const MAX_M = 8

# What the GPU methods accept: dense Float64 arrays (and the `Diagonal` / vector cost terms of the reference's demos).  Everything else —
# Float32, BigFloat, dual numbers, views, sparse or static arrays — stays on the reference's own `AbstractArray{T}` methods, which
# `install!` leaves in place: the methods below are strictly MORE SPECIFIC than backward_pass.jl:162,179,217, not replacements.
const DenseCost2 = Union{Matrix{Float64},Diagonal{Float64,Vector{Float64}}}
const DenseCost3 = Array{Float64,3}

# the reference's own method for these arguments, by its declared signature (`invoke` does not see the narrower GPU method)
_ref_sig(::Val{:ltv_ti}) = Tuple{Any,Any,AbstractArray{Float64,2},Any,Any,AbstractArray{Float64,3},Any,Any,Any,Any,Any,Any}    # :162
_ref_sig(::Val{:ltv_tv}) = Tuple{Any,Any,AbstractArray{Float64,3},Any,Any,AbstractArray{Float64,3},Any,Any,Any,Any,Any,Any}    # :179
_ref_sig(::Val{:lti})    = Tuple{Any,Any,AbstractArray{Float64,2},Any,Any,AbstractMatrix{Float64},Any,Any,Any,Any,Any,Any}      # :217

# true when the C library takes this call (sizes in range, every operand a Float64 array)
function _gpu_takes(cx, cu, cxu, cuu, fu, x, u)
    n, m = size(cx, 1), size(cu, 1)
    (1 <= n <= MAX_N && 1 <= m <= MAX_M) || return false
    return all(a -> a isa AbstractArray{Float64}, (cx, cu, cxu, cuu, fu, x, u))
end

"""
    install!(ref::Module = Main.DifferentialDynamicProgramming)

ADDS three methods to the loaded reference package's `back_pass`, on signatures strictly narrower than its own linear-system methods
(src/backward_pass.jl:162,179,217: `cxx::AbstractArray{T,2|3}`, `fx::AbstractArray{T,3}` / `AbstractMatrix{T}` — here `Matrix{Float64}` /
`Diagonal{Float64}` / `Array{Float64,3}`), so that the reference's own `iLQG(f,costfun,df,x0,u0; ...)` — arbitrary Julia closures, its own
line search, trace and printing — runs STEP 2 (iLQG.jl:235-251) on the GPU.  The reference's methods are NOT overwritten: they remain the
fallback, reached by dispatch for every other element or array type and by `invoke` from the GPU methods when the problem is outside
the kernels' range (n > $MAX_N, m > $MAX_M, a non-`Float64` operand) or the library refuses the shape (`DDPError`).  The second-order
methods (`fxx, fxu, fuu`, :81,:132) have another arity and are untouched.  Returns policies of the reference's own `GaussianPolicy` type.
`uninstall!` removes the three methods again.
"""
function install!(ref::Module=getfield(Main, :DifferentialDynamicProgramming))
    _ref[] === ref && !isempty(_installed) && return ref
    isempty(_installed) || uninstall!()
    pol(N, n, m, K, k, Σ, Σi) = ref.GaussianPolicy(N, n, m, K, k, Σ, Σi)
    function bp(kind::Val, cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u)
        fallback() = invoke(ref.back_pass, _ref_sig(kind), cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u)
        _gpu_takes(cx, cu, cxu, cuu, fu, x, u) || return fallback()
        try
            return back_pass(cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u; policy=pol)
        catch err
            err isa DDPError || rethrow()
            return fallback()             # a shape the library refuses: the reference's own method, as before `install!`
        end
    end
    before = Set(methods(ref.back_pass))
    @eval ref begin
        back_pass(cx, cu, cxx::$DenseCost2, cxu, cuu, fx::Array{Float64,3}, fu, λ, regType, lims, x, u) = $bp(Val(:ltv_ti), cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u)
        back_pass(cx, cu, cxx::$DenseCost3, cxu, cuu, fx::Array{Float64,3}, fu, λ, regType, lims, x, u) = $bp(Val(:ltv_tv), cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u)
        back_pass(cx, cu, cxx::$DenseCost2, cxu, cuu, fx::Matrix{Float64}, fu, λ, regType, lims, x, u) = $bp(Val(:lti), cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u)
    end
    append!(_installed, (mth for mth in methods(ref.back_pass) if !(mth in before)))
    length(_installed) == 3 || @warn "install!: expected three new back_pass methods" length(_installed)
    _ref[] = ref
    return ref
end

"""
    uninstall!()

Deletes the methods `install!` added; the reference's `back_pass` dispatches as it did before.
"""
function uninstall!()
    for mth in _installed
        Base.delete_method(mth)
    end
    empty!(_installed)
    _ref[] = nothing
    return nothing
end

"""
    iLQG(f, costfun, df, x0, u0; lims=[], α=..., tol_fun=1e-7, tol_grad=1e-4, max_iter=500, λ=1., dλ=1., λfactor=1.6, λmax=1e10,
         λmin=1e-6, regType=1, reduce_ratio_min=0, diff_fun=-, plot=1, verbosity=2, plot_fun=x->0, cost=[], traj_prev=0, print_head=10)

THE drop-in for src/iLQG.jl:143-163 (same positional and keyword arguments, same return tuple).
* `f` a registered problem (`LQProblem`, `PendcartProblem` — they are callable like `f(x,u,i)`): the whole iteration runs on the
  device (`costfun`, `df` are the family's own); `u0[m,N,B]` solves a batch.
* any other `f`: the reference's own loop with its `back_pass` rebound to the GPU (`install!`); needs the reference package loaded.
"""
function iLQG(f, costfun, df_, x0, u0; kwargs...)
    if f isa RegisteredProblem
        return iLQG(f, x0, u0; kwargs...)
    end
    ref = _ref[] === nothing ? install!() : _ref[]
    return Base.invokelatest(ref.iLQG, f, costfun, df_, x0, u0; kwargs...)
end

"""
    mpc_shift(a, shift=1; zero_tail=false)

Receding-horizon warm start between two solves: `out[:, i] = a[:, i+shift]` along the time axis (the last axis of `u[m,N]`,
`x[n,N]`, `K[m,n,N]`), the vacated tail repeats the last column or is zero.  Host arrays; device-resident loops call
`ddp_mpc_shift_f64_dev` on their buffers (`mpc_shift_dev!`).
"""
function mpc_shift(a::AbstractArray, shift::Integer=1; zero_tail::Bool=false)
    N = size(a, ndims(a)); out = similar(a)
    for i in 1:N
        src = i + shift
        selectdim(out, ndims(a), i) .= src <= N ? selectdim(a, ndims(a), src) : (zero_tail ? zero(eltype(a)) : selectdim(a, ndims(a), N))
    end
    return out
end

# ======================================================================================= device-resident API
"""
    DevArray(dims...; handle) / DevArray(hostarray; handle)

fp64 device buffer owned by the library's allocator (`ddp_malloc`); `Array(d)` copies back.  Any array type whose `pointer` is a
device pointer (AMDGPU.ROCArray) can be passed to the `_dev!` functions instead.
"""
mutable struct DevArray{T}
    ptr::Ptr{T}
    dims::Dims
    handle::Handle
    function DevArray{T}(dims::Dims; handle::Handle=default_handle()) where {T}
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(@ccall libddp.ddp_malloc(handle.ptr::Ptr{Cvoid}, (sizeof(T) * max(prod(dims), 1))::Csize_t, r::Ptr{Ptr{Cvoid}})::Cint)
        d = new{T}(Ptr{T}(r[]), dims, handle)
        finalizer(dd -> (@ccall libddp.ddp_free(dd.handle.ptr::Ptr{Cvoid}, dd.ptr::Ptr{Cvoid})::Cint), d)
        return d
    end
end
DevArray(dims::Integer...; handle::Handle=default_handle()) = DevArray{Float64}(Dims(dims); handle=handle)
function DevArray(a::AbstractArray; handle::Handle=default_handle())
    T = eltype(a) <: Integer ? Int32 : Float64
    h = Array{T}(a)
    d = DevArray{T}(size(h); handle=handle)
    GC.@preserve h check(@ccall libddp.ddp_memcpy_h2d(handle.ptr::Ptr{Cvoid}, d.ptr::Ptr{Cvoid}, h::Ptr{Cvoid}, sizeof(h)::Csize_t)::Cint)
    return d
end
Base.size(d::DevArray) = d.dims
Base.pointer(d::DevArray) = d.ptr
function Base.Array(d::DevArray{T}) where {T}
    h = Array{T}(undef, d.dims...)
    GC.@preserve h check(@ccall libddp.ddp_memcpy_d2h(d.handle.ptr::Ptr{Cvoid}, h::Ptr{Cvoid}, d.ptr::Ptr{Cvoid}, sizeof(h)::Csize_t)::Cint)
    return h
end
dptr(a) = a === nothing ? NULLF : Ptr{Float64}(UInt(pointer(a)))
diptr(a) = a === nothing ? NULLI : Ptr{Int32}(UInt(pointer(a)))

"""
    back_pass_dev!(K,k,Quu,Vx,Vxx,dV,diverge, desc::BPDesc, cx,cu,cxx,cxu,cuu,fx,fu, λ, lims,u; active=nothing)

All operands device-resident (layouts of include/ddp_amd.h: batch slowest); asynchronous on the handle's stream.
"""
function back_pass_dev!(K, k, Quu, Vx, Vxx, dV, diverge, desc::BPDesc, cx, cu, cxx, cxu, cuu, fx, fu, λ, lims, u; active=nothing,
                        handle::Handle=default_handle())
    check(@ccall libddp.ddp_back_pass_f64_dev(handle.ptr::Ptr{Cvoid}, Ref(desc)::Ptr{BPDesc}, dptr(cx)::Ptr{Float64}, dptr(cu)::Ptr{Float64},
        dptr(cxx)::Ptr{Float64}, dptr(cxu)::Ptr{Float64}, dptr(cuu)::Ptr{Float64}, dptr(fx)::Ptr{Float64}, dptr(fu)::Ptr{Float64},
        dptr(λ)::Ptr{Float64}, dptr(lims)::Ptr{Float64}, dptr(u)::Ptr{Float64}, diptr(active)::Ptr{Int32},
        dptr(K)::Ptr{Float64}, dptr(k)::Ptr{Float64}, dptr(Quu)::Ptr{Float64}, dptr(Vx)::Ptr{Float64}, dptr(Vxx)::Ptr{Float64},
        dptr(dV)::Ptr{Float64}, diptr(diverge)::Ptr{Int32})::Cint)
end

"`P::CProblem` with DEVICE pointers (`cproblem(problem, N, B; A=dptr(dA), ...)`); `alpha` is a host vector"
function forward_pass_dev!(xnew, unew, cnew, csum, P::CProblem, K, k, x0, u, x, alpha::Vector{Float64}, lims; active=nothing,
                           handle::Handle=default_handle())
    GC.@preserve alpha check(@ccall libddp.ddp_forward_pass_f64_dev(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, dptr(K)::Ptr{Float64},
        dptr(k)::Ptr{Float64}, dptr(x0)::Ptr{Float64}, dptr(u)::Ptr{Float64}, dptr(x)::Ptr{Float64}, alpha::Ptr{Float64},
        length(alpha)::Cint, dptr(lims)::Ptr{Float64}, diptr(active)::Ptr{Int32},
        dptr(xnew)::Ptr{Float64}, dptr(unew)::Ptr{Float64}, dptr(cnew)::Ptr{Float64}, dptr(csum)::Ptr{Float64})::Cint)
end

function df_dev!(cx, cu, fx, fu, P::CProblem, x, u; active=nothing, handle::Handle=default_handle())
    check(@ccall libddp.ddp_df_f64_dev(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, dptr(x)::Ptr{Float64}, dptr(u)::Ptr{Float64},
        diptr(active)::Ptr{Int32}, dptr(cx)::Ptr{Float64}, dptr(cu)::Ptr{Float64}, dptr(fx)::Ptr{Float64}, dptr(fu)::Ptr{Float64})::Cint)
end

"""
    iLQG_dev!(x,u,K,k,Quu,Vx,Vxx,cost,stats, P::CProblem, x0, u0, lims; prerolled=false, cost0=nothing, opts...) -> global_iters

Whole batched solves on device buffers (what an MPC loop keeps calling with its shifted solution: `prerolled=true`); x0/u0 must
not alias x/u.  `stats[8,B]` = status, iter, accepted_iter, n_backpass, n_forward, λ, g_norm, sum(cost) per trajectory.
"""
function iLQG_dev!(x, u, K, k, Quu, Vx, Vxx, cost, stats, P::CProblem, x0, u0, lims; prerolled::Bool=false, cost0=nothing, trace7=nothing,
                   trace_cap::Integer=0, α=DEFAULT_ALPHA, tol_fun=1e-7, tol_grad=1e-4, max_iter=500, λ=1.0, dλ=1.0, λfactor=1.6, λmax=1e10,
                   λmin=1e-6, regType=1, reduce_ratio_min=0, handle::Handle=default_handle())
    o = _opts(α, tol_fun, tol_grad, max_iter, λ, dλ, λfactor, λmax, λmin, regType, reduce_ratio_min)
    git = Ref{Cint}(0)
    check(@ccall libddp.ddp_ilqg_ex_f64_dev(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, Ref(o)::Ptr{ILQGOpts}, dptr(x0)::Ptr{Float64},
        (prerolled ? 1 : 0)::Cint, dptr(u0)::Ptr{Float64}, dptr(cost0)::Ptr{Float64}, dptr(lims)::Ptr{Float64},
        dptr(x)::Ptr{Float64}, dptr(u)::Ptr{Float64}, dptr(K)::Ptr{Float64}, dptr(k)::Ptr{Float64}, dptr(Quu)::Ptr{Float64},
        dptr(Vx)::Ptr{Float64}, dptr(Vxx)::Ptr{Float64}, dptr(cost)::Ptr{Float64}, dptr(stats)::Ptr{Float64}, trace_cap::Cint,
        dptr(trace7)::Ptr{Float64}, git::Ptr{Cint})::Cint)
    return Int(git[])
end

"`dst[:, i, b] = src[:, i+shift, b]` on device buffers `a[d, N, B]` (out of place)"
function mpc_shift_dev!(dst, src, d::Integer, N::Integer, B::Integer, shift::Integer=1; zero_tail::Bool=false, handle::Handle=default_handle())
    check(@ccall libddp.ddp_mpc_shift_f64_dev(handle.ptr::Ptr{Cvoid}, d::Cint, N::Cint, B::Cint, shift::Cint, (zero_tail ? 1 : 0)::Cint,
        dptr(src)::Ptr{Float64}, dptr(dst)::Ptr{Float64})::Cint)
end

# ======================================================================================= KL-constrained path
"""
    ∇kl(traj_prev) -> cx, cu, cxx, cxu, cuu          (klutils.jl:8-23; cxu is m×n×T like the reference; batch axis allowed)
"""
function ∇kl(traj_prev; handle::Handle=default_handle())
    _isempty_policy(traj_prev) && return (0, 0, 0, 0, 0)
    m, n, T = traj_prev.m, traj_prev.n, traj_prev.T
    K, k, Σi = _f64(traj_prev.K), _f64(traj_prev.k), _f64(traj_prev.Σi)
    B = ndims(k) == 3 ? size(k, 3) : 1
    bt = ndims(k) == 3 ? (B,) : ()
    cx, cu, cxx, cxu, cuu = zeros(n, T, bt...), zeros(m, T, bt...), zeros(n, n, T, bt...), zeros(m, n, T, bt...), zeros(m, m, T, bt...)
    GC.@preserve K k Σi cx cu cxx cxu cuu begin
        check(@ccall libddp.ddp_kl_terms_f64(handle.ptr::Ptr{Cvoid}, n::Cint, m::Cint, T::Cint, B::Cint, K::Ptr{Float64},
            k::Ptr{Float64}, Σi::Ptr{Float64}, cx::Ptr{Float64}, cu::Ptr{Float64}, cxx::Ptr{Float64}, cxu::Ptr{Float64},
            cuu::Ptr{Float64})::Cint)
    end
    return cx, cu, cxx, cxu, cuu
end

"""
    back_pass_gps(cx,cu,cxx,cxu,cuu,fx,fu,lims,x,u,kl_cost_terms) -> diverge, GaussianPolicy(N,n,m,K,k,Quui,Quu), Vx, Vxx, dV

Same signature and return values as backward_pass.jl:259; `kl_cost_terms = (∇kl(traj_prev), ηbracket)` with `ηbracket` a 3-vector
or a 3×N matrix.  One trajectory (a batch goes through `iLQGkl` below / `ddp_back_pass_gps_f64_dev`).  Shapes: n ≤ 32, m ≤ 8 (the library
returns an error beyond; the plain `back_pass` reaches n ≤ 64).
"""
function back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, kl_cost_terms; handle::Handle=default_handle(), policy=GaussianPolicy{Float64})
    n, N = size(cx); m = size(cu, 1)
    cx, cu, cxx, cxu, cuu, fx, fu, u = map(_f64, (cx, cu, cxx, cxu, cuu, fx, fu, u))
    cxkl, cukl, cxxkl, cxukl, cuukl = map(_f64, kl_cost_terms[1])
    ηb = kl_cost_terms[2]
    η = isa(ηb, AbstractMatrix) ? _f64(ηb[2, :]) : [Float64(ηb[2])]
    limsp = _lims(lims)
    has_lims = !isempty(limsp)
    d = BPDesc(n, m, N, 1, 1, 0, 1, 0, 1, has_lims)
    K = zeros(m, n, N); k = zeros(m, N); Quu = zeros(m, m, N); Quui = zeros(m, m, N); Vx = zeros(n, N); Vxx = zeros(n, n, N)
    dV = zeros(2); diverge = zeros(Int32, 1)
    GC.@preserve cx cu cxx cxu cuu fx fu u cxkl cukl cxxkl cxukl cuukl η limsp K k Quu Quui Vx Vxx dV diverge begin
        t = KLCostTerms(pointer(cxkl), pointer(cukl), pointer(cxxkl), pointer(cxukl), pointer(cuukl), pointer(η), isa(ηb, AbstractMatrix))
        check(@ccall libddp.ddp_back_pass_gps_f64(handle.ptr::Ptr{Cvoid}, Ref(d)::Ptr{BPDesc}, cx::Ptr{Float64}, cu::Ptr{Float64},
            cxx::Ptr{Float64}, cxu::Ptr{Float64}, cuu::Ptr{Float64}, fx::Ptr{Float64}, fu::Ptr{Float64}, Ref(t)::Ptr{KLCostTerms},
            _ptr_or_null(limsp)::Ptr{Float64}, (has_lims ? pointer(u) : NULLF)::Ptr{Float64},
            K::Ptr{Float64}, k::Ptr{Float64}, Quu::Ptr{Float64}, Quui::Ptr{Float64}, Vx::Ptr{Float64}, Vxx::Ptr{Float64},
            dV::Ptr{Float64}, diverge::Ptr{Int32})::Cint)
    end
    return Int(diverge[1]), policy(N, n, m, K, k, Quui, Quu), Vx, Vxx, dV
end

"""
    forward_covariance(fx, R1, traj) -> sigmanew       (forward_pass.jl:37-56 with `df(model,·)[1]`, `covariance(model,·)` passed in)
"""
function forward_covariance(fx::Array{Float64,3}, R1::Matrix{Float64}, traj; handle::Handle=default_handle())
    n, m, N = traj.n, traj.m, traj.T
    S = zeros(n + m, n + m, N); K, Σ = _f64(traj.K), _f64(traj.Σ)
    GC.@preserve fx R1 K Σ S begin
        check(@ccall libddp.ddp_forward_covariance_f64(handle.ptr::Ptr{Cvoid}, n::Cint, m::Cint, N::Cint, 1::Cint, fx::Ptr{Float64},
            0::Cint, R1::Ptr{Float64}, K::Ptr{Float64}, Σ::Ptr{Float64}, S::Ptr{Float64})::Cint)
    end
    return S
end

"""
    kl_div_wiki(xnew, xold, Σ_new, traj_new, traj_prev) -> kldiv (or Inf)      (klutils.jl:70-103)
"""
function kl_div_wiki(xnew, xold, Σ_new, traj_new, traj_prev; handle::Handle=default_handle())
    n, m, T = traj_new.n, traj_new.m, traj_new.T
    kld = zeros(T); mean_ = zeros(1)
    xnew, xold, Σ_new = map(_f64, (xnew, xold, Σ_new))
    Kn, kn, Σn, Kp, kp, Σp, Σip = map(_f64, (traj_new.K, traj_new.k, traj_new.Σ, traj_prev.K, traj_prev.k, traj_prev.Σ, traj_prev.Σi))
    GC.@preserve xnew xold Σ_new Kn kn Σn Kp kp Σp Σip kld mean_ begin
        check(@ccall libddp.ddp_kl_div_f64(handle.ptr::Ptr{Cvoid}, n::Cint, m::Cint, T::Cint, 1::Cint, xnew::Ptr{Float64},
            xold::Ptr{Float64}, Σ_new::Ptr{Float64}, Kn::Ptr{Float64}, kn::Ptr{Float64}, Σn::Ptr{Float64}, Kp::Ptr{Float64},
            kp::Ptr{Float64}, Σp::Ptr{Float64}, Σip::Ptr{Float64}, kld::Ptr{Float64}, mean_::Ptr{Float64})::Cint)
    end
    return isinf(mean_[1]) && all(isfinite, kld) ? Inf : kld
end
# For ONE trajectory calc_η, geom and the iLQGkl loop stay the reference's scalar Julia code (klutils.jl:112-155, iLQGkl.jl): with
# the four functions above rebound, src/iLQGkl.jl:90,100,133 and klutils.jl:114 run on the GPU unchanged.  A BATCH of
# KL-constrained solves keeps the dual variable of every trajectory on the device:

"""
    KLDualState(ηbracket::Matrix (3×B), del0; handle) — the per-trajectory dual state of a batch (device arrays)
    kl_dual_begin!(s, it) -> n_live;  kl_dual_retry!(s, diverge::DevArray{Int32}) -> n_pending;
    kl_dual_update!(s, kl_step, klmean::DevArray) -> n_live        (calc_η klutils.jl:112-133, iLQGkl.jl:91-122,169-177)
`s.eta` is the array `KLCostTerms.eta` points at for `ddp_back_pass_gps_f64_dev`.
"""
struct KLDualState
    etab::DevArray{Float64}
    eta::DevArray{Float64}
    del::DevArray{Float64}
    divergence::DevArray{Float64}
    satisfied::DevArray{Int32}
    status::DevArray{Int32}
    live::DevArray{Int32}
    pend::DevArray{Int32}
    iters::DevArray{Int32}
    nback::DevArray{Int32}
    handle::Handle
end
function KLDualState(ηbracket::AbstractMatrix, del0::Real; handle::Handle=default_handle())
    B = size(ηbracket, 2)
    size(ηbracket, 1) == 3 || throw(ArgumentError("ηbracket must be 3×B"))
    zi() = DevArray(zeros(Int32, B); handle=handle)
    KLDualState(DevArray(_f64(ηbracket); handle=handle), DevArray(_f64(ηbracket[2, :]); handle=handle), DevArray(fill(Float64(del0), B); handle=handle),
                DevArray(zeros(B); handle=handle), zi(), zi(), DevArray(ones(Int32, B); handle=handle), zi(), zi(), zi(), handle)
end
_kldual(s::KLDualState) = KLDual(s.etab.ptr, s.eta.ptr, s.del.ptr, s.divergence.ptr, s.satisfied.ptr, s.status.ptr, s.live.ptr, s.pend.ptr,
                                 s.iters.ptr, s.nback.ptr)
function kl_dual_begin!(s::KLDualState, it::Integer)
    c = Ref{Cint}(0); B = prod(s.eta.dims)
    GC.@preserve s check(@ccall libddp.ddp_kl_dual_begin_f64_dev(s.handle.ptr::Ptr{Cvoid}, B::Cint, it::Cint, Ref(_kldual(s))::Ptr{KLDual}, c::Ptr{Cint})::Cint)
    return Int(c[])
end
function kl_dual_retry!(s::KLDualState, diverge::DevArray{Int32})
    c = Ref{Cint}(0); B = prod(s.eta.dims)
    GC.@preserve s diverge check(@ccall libddp.ddp_kl_dual_retry_f64_dev(s.handle.ptr::Ptr{Cvoid}, B::Cint, Ref(_kldual(s))::Ptr{KLDual},
                                                                          diverge.ptr::Ptr{Int32}, c::Ptr{Cint})::Cint)
    return Int(c[])
end
function kl_dual_update!(s::KLDualState, kl_step::Real, klmean::DevArray{Float64})
    c = Ref{Cint}(0); B = prod(s.eta.dims)
    GC.@preserve s klmean check(@ccall libddp.ddp_kl_dual_update_f64_dev(s.handle.ptr::Ptr{Cvoid}, B::Cint, kl_step::Cdouble, Ref(_kldual(s))::Ptr{KLDual},
                                                                          klmean.ptr::Ptr{Float64}, c::Ptr{Cint})::Cint)
    return Int(c[])
end

"""
    iLQGkl(problem, x0, traj_prev, fx_model, R1; kl_step=1, lims=[], max_iter=50, cost, ηbracket=[1e-8,1,1e16], del0=1e-4)
        -> x, u, traj_new, Vx, Vxx, cost, trace

The single-constraint loop of src/iLQGkl.jl:25-178,234-252 as ONE library call (`ddp_ilqgkl_f64`): `problem` stands in for the three
closures, `fx_model[n,n,N(,B)]` / `R1[n,n]` are what `df(model,·)` / `covariance(model,·)` of the un-vendored model package return.
`x0[n,N(,B)]` must be pre-rolled and `cost` given (:66-73); with a batch every trajectory owns its η bracket (`ηbracket` 3 or 3×B).
`trace` is a Dict: :status (1 SUCCESS :169, 2 η > ηmax :174, 3 max_iter :234), :iter, :n_backpass, :satisfied, :η (3×B), :divergence,
:cost, :improvement, :expected_reduction, :grad_norm, :dV.  `traj_prev.k` is left untouched (the reference zeroes and restores it, :51,247).
"""
function iLQGkl(problem::RegisteredProblem, x0, traj_prev, fx_model, R1; kl_step=1.0, lims=[], max_iter=50, cost=[],
                ηbracket=[1e-8, 1.0, 1e16], del0=1e-4, constrain_per_step=false, diff_fun=-, handle::Handle=default_handle(), policy=GaussianPolicy{Float64})
    constrain_per_step && error("constrain_per_step (iLQGkl.jl:180-232) is not offloaded (it cannot run upstream either: klutils.jl:195)")
    isempty(cost) && error("Initial trajectory supplied, initial cost must also be supplied")                 # :69
    batched = ndims(x0) == 3
    n, N = size(x0, 1), size(x0, 2)
    u0 = _f64(traj_prev.k)
    m = size(u0, 1)
    size(u0, 2) == N || error("pre-rolled initial trajectory must be of correct length (size(x0,2) == N)")     # :72
    B = batched ? size(x0, 3) : 1
    P = cproblem(problem, N, B; diff=diff_fun)
    CL = cost_len(problem, N)
    x0 = _f64(x0); Kp = _f64(traj_prev.K); Sp = _f64(traj_prev.Σ); Sip = _f64(traj_prev.Σi); fxm = _f64(fx_model); R1 = _f64(R1)
    c0 = batched ? (ndims(cost) == 2 ? vec(sum(cost, dims=1)) : _f64(vec(cost))) : [Float64(sum(cost))]      # only sum(cost) enters (:74,135)
    length(c0) == B || error("cost must hold one entry (or one column) per trajectory")
    etab = ndims(ηbracket) == 2 ? _f64(copy(ηbracket)) : repeat(_f64(ηbracket), 1, B)                        # copy (:52)
    size(etab) == (3, B) || error("ηbracket must be a 3-vector or 3×B")
    limsp = _lims(lims)
    o = ILQGKLOpts(kl_step, max_iter, (1e-8, 1.0, 1e16), del0)
    bt = batched ? (B,) : ()                                     # unbatched: the caller's arrays have no batch axis (result_pair, not dropdims)
    x, x_r = result_pair((n, N, B), (n, N, bt...)); u, u_r = result_pair((m, N, B), (m, N, bt...)); K, K_r = result_pair((m, n, N, B), (m, n, N, bt...))
    S, S_r = result_pair((m, m, N, B), (m, m, N, bt...)); Si, Si_r = result_pair((m, m, N, B), (m, m, N, bt...))
    Vx, Vx_r = result_pair((n, N, B), (n, N, bt...)); Vxx, Vxx_r = result_pair((n, n, N, B), (n, n, N, bt...))
    cnew, cnew_r = result_pair((CL, B), (CL, bt...)); dV = zeros(2, B); st = zeros(12, B)
    its = Ref{Cint}(0)
    GC.@preserve problem x0 c0 Kp u0 Sp Sip fxm R1 limsp etab x u K S Si Vx Vxx cnew dV st begin
        check(@ccall libddp.ddp_ilqgkl_f64(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, Ref(o)::Ptr{ILQGKLOpts}, x0::Ptr{Float64},
            c0::Ptr{Float64}, Kp::Ptr{Float64}, u0::Ptr{Float64}, Sp::Ptr{Float64}, Sip::Ptr{Float64}, fxm::Ptr{Float64},
            (ndims(fxm) == 4)::Cint, R1::Ptr{Float64}, _ptr_or_null(limsp)::Ptr{Float64}, etab::Ptr{Float64},
            x::Ptr{Float64}, u::Ptr{Float64}, K::Ptr{Float64}, S::Ptr{Float64}, Si::Ptr{Float64}, Vx::Ptr{Float64}, Vxx::Ptr{Float64},
            cnew::Ptr{Float64}, dV::Ptr{Float64}, st::Ptr{Float64}, its::Ptr{Cint})::Cint)
    end
    trace = Dict{Symbol,Any}(:status => Int.(st[1, :]), :iter => Int.(st[2, :]), :n_backpass => Int.(st[3, :]), :satisfied => st[4, :] .!= 0,
                             :η => etab, :divergence => st[8, :], :cost => st[9, :], :improvement => st[10, :],
                             :expected_reduction => st[11, :], :grad_norm => st[12, :], :dV => dV, :batch_iterations => Int(its[]))
    traj_new = policy(N, n, m, K_r, copy(u_r), S_r, Si_r)                                                     # traj_new.k = copy(u) (:239)
    return x_r, u_r, traj_new, Vx_r, Vxx_r, cnew_r, trace
end

end # module
