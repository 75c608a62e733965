# DDPAmd.jl — thin `@ccall` binding of libddp_amd.so (include/ddp_amd.h) that keeps the reference's
# call signatures for the hot path.  NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia
# toolchain; every call below is mirrored one-to-one by the ctypes host in ../__init__.py, which is what
# the tests drive.  See INTEGRATION.md for how a maintainer of DifferentialDynamicProgramming.jl wires it in.
#
#   DDPAmd.back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u)   ↔ src/backward_pass.jl:162-252
#   DDPAmd.boxQP(H,g,lower,upper,x0)                               ↔ src/boxQP.jl:29-188
#   DDPAmd.forward_pass(traj_new,x0,u,x,α,problem,lims)            ↔ src/forward_pass.jl:9-33
#   DDPAmd.iLQG(problem,x0,u0; lims, kwargs...)                    ↔ src/iLQG.jl:143-341
module DDPAmd

using LinearAlgebra

const libddp = get(ENV, "DDP_AMD_LIB", joinpath(@__DIR__, "..", "libddp_amd.so"))

# ---- C structs (field order = include/ddp_amd.h) -------------------------------------------------
struct BPDesc
    n::Cint; m::Cint; N::Cint; B::Cint
    fx_tv::Cint; fx_batched::Cint; cost_tv::Cint; cost_batched::Cint
    regType::Cint; has_lims::Cint
end

struct CProblem
    kind::Cint; n::Cint; m::Cint; N::Cint; B::Cint
    A::Ptr{Float64}; Bm::Ptr{Float64}; dyn_tv::Cint; dyn_batched::Cint
    Q::Ptr{Float64}; R::Ptr{Float64}
    g::Float64; l::Float64; h::Float64; d::Float64
    goal::NTuple{4,Float64}
end

struct ILQGOpts
    lambda::Float64; dlambda::Float64; lambda_factor::Float64; lambda_max::Float64; lambda_min::Float64
    tol_fun::Float64; tol_grad::Float64
    max_iter::Cint; regType::Cint
    reduce_ratio_min::Float64
    n_alpha::Cint
    alpha::NTuple{16,Float64}
end

mutable struct Handle
    ptr::Ptr{Cvoid}
    function Handle(device::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(@ccall libddp.ddp_create(device::Cint, r::Ptr{Ptr{Cvoid}})::Cint)
        h = new(r[])
        finalizer(h -> (@ccall libddp.ddp_destroy(h.ptr::Ptr{Cvoid})::Cint), h)
        h
    end
end

last_error() = unsafe_string(@ccall libddp.ddp_last_error()::Cstring)
check(rc) = rc == 0 ? nothing : error("libddp_amd: $(last_error()) (rc=$rc)")   # there is no CPU fallback

const _default = Ref{Union{Nothing,Handle}}(nothing)
default_handle() = (_default[] === nothing && (_default[] = Handle(0)); _default[])

# the reference's output container (src/iLQG.jl:39-53); redefine or reuse the package's own type
mutable struct GaussianPolicy{P}
    T::Int; n::Int; m::Int
    K::Array{P,3}; k::Array{P,2}; Σ::Array{P,3}; Σi::Array{P,3}
end

# ---- registered problem families (stand-ins for the closures f / costfun / df) --------------------
struct LQProblem
    A::Array{Float64}; B::Array{Float64}; Q::Matrix{Float64}; R::Matrix{Float64}
end
Base.@kwdef struct PendcartProblem
    g::Float64 = 9.82; l::Float64 = 0.35; h::Float64 = 0.01; d::Float64 = 0.99
    Q::Matrix{Float64} = Matrix(Diagonal([10.0, 1, 2, 1])); R::Matrix{Float64} = fill(1.0, 1, 1)
    goal::Vector{Float64} = [π, 0, 0, 0]
end

cproblem(p::LQProblem, N, B) = CProblem(0, size(p.A, 1), size(p.B, 2), N, B, pointer(p.A), pointer(p.B),
                                         ndims(p.A) == 3, 0, pointer(p.Q), pointer(p.R), 0, 0, 0, 0, (0.0, 0.0, 0.0, 0.0))
cproblem(p::PendcartProblem, N, B) = CProblem(1, 4, 1, N, B, C_NULL, C_NULL, 0, 0, pointer(p.Q), pointer(p.R),
                                               p.g, p.l, p.h, p.d, Tuple(p.goal))

_f64(a) = Array{Float64}(a)      # dense column-major copy (handles Diagonal cxx, Vector cuu of the demos)

"""
    back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u) -> diverge, GaussianPolicy, Vx, Vxx, dV

Same signature, dispatch (array rank selects LTI / LTV / time-varying cost) and return values as the
reference's linear-system `back_pass` methods.  One trajectory (B = 1) goes through the host-pointer
entry point; arrays are passed as they are (Julia memory layout is the ABI's layout).
"""
function back_pass(cx, cu, cxx, cxu, cuu, fx, fu, λ, regType, lims, x, u; handle=default_handle())
    n, N = size(cx); m = size(cu, 1)
    cx, cu, cxx, cxu, fx, fu, u = map(_f64, (cx, cu, cxx, cxu, fx, fu, u))
    cuu = reshape(_f64(cuu), m, m, :)
    has_lims = !isempty(lims)
    d = BPDesc(n, m, N, 1, ndims(fx) == 3, 0, ndims(cxx) == 3, 0, regType, has_lims)
    K = zeros(m, n, N); k = zeros(m, N); Quu = zeros(m, m, N); Vx = zeros(n, N); Vxx = zeros(n, n, N); dV = zeros(2)
    diverge = Ref{Int32}(0)
    lam = [Float64(λ)]
    limsp = has_lims ? _f64(lims) : Float64[]
    GC.@preserve cx cu cxx cxu cuu fx fu u lam limsp K k Quu Vx Vxx dV begin
        check(@ccall libddp.ddp_back_pass_f64(handle.ptr::Ptr{Cvoid}, Ref(d)::Ptr{BPDesc},
            cx::Ptr{Float64}, cu::Ptr{Float64}, cxx::Ptr{Float64}, cxu::Ptr{Float64}, cuu::Ptr{Float64},
            fx::Ptr{Float64}, fu::Ptr{Float64}, lam::Ptr{Float64},
            (has_lims ? pointer(limsp) : Ptr{Float64}(C_NULL))::Ptr{Float64},
            (has_lims ? pointer(u) : Ptr{Float64}(C_NULL))::Ptr{Float64},
            K::Ptr{Float64}, k::Ptr{Float64}, Quu::Ptr{Float64}, Vx::Ptr{Float64}, Vxx::Ptr{Float64},
            dV::Ptr{Float64}, diverge::Ptr{Int32})::Cint)
    end
    # Σ (= Quui) is never written by the reference's back_pass (`undef`, backward_pass.jl:231); zeros here
    return Int(diverge[]), GaussianPolicy(N, n, m, K, k, zeros(m, m, N), Quu), Vx, Vxx, dV
end

"""
    boxQP(H,g,lower,upper,x0; maxIter=100, ...) -> x, result, Hfree, free
"""
function boxQP(H, g, lower, upper, x0::AbstractVector; maxIter=100, minGrad=1e-8, minRelImprove=1e-8,
               stepDec=0.6, minStep=1e-22, Armijo=0.1, handle=default_handle())
    m = size(H, 1)
    H, g, lower, upper, x0 = map(_f64, (H, g, lower, upper, x0))
    x = zeros(m); Hf = zeros(m, m); res = Ref{Int32}(0); fr = zeros(UInt8, m)
    opts = (Cint(maxIter), minGrad, minRelImprove, stepDec, minStep, Armijo)
    GC.@preserve H g lower upper x0 x Hf fr begin
        check(@ccall libddp.ddp_boxqp_f64(handle.ptr::Ptr{Cvoid}, m::Cint, 1::Cint, H::Ptr{Float64}, g::Ptr{Float64},
            lower::Ptr{Float64}, upper::Ptr{Float64}, x0::Ptr{Float64}, Ref(opts)::Ptr{Cvoid},
            x::Ptr{Float64}, res::Ptr{Int32}, Hf::Ptr{Float64}, fr::Ptr{UInt8})::Cint)
    end
    free = BitVector(fr .!= 0); nf = count(free)
    return x, Int(res[]), UpperTriangular(Hf[1:nf, 1:nf]), free, nothing
end

"""
    forward_pass(traj_new, x0, u, x, α, problem, lims) -> xnew, unew, cnew

`problem` (LQProblem / PendcartProblem) replaces the closures `f`, `costfun`; `diff` is `-`.
"""
function forward_pass(traj_new, x0, u, x, α, problem, lims; handle=default_handle())
    m, N = size(u); n = length(x0)
    P = cproblem(problem, N, 1)
    CL = problem isa PendcartProblem ? N + 1 : N
    empty = traj_new === nothing || traj_new.T == 0
    xnew = zeros(n, N); unew = zeros(m, N); cnew = zeros(CL); csum = zeros(1)
    a = [Float64(α)]; x0 = _f64(x0); u = _f64(u)
    Kp = empty ? Ptr{Float64}(C_NULL) : pointer(traj_new.K); kp = empty ? Ptr{Float64}(C_NULL) : pointer(traj_new.k)
    xp = empty ? Ptr{Float64}(C_NULL) : pointer(x)
    lp = isempty(lims) ? Ptr{Float64}(C_NULL) : pointer(_f64(lims))
    GC.@preserve problem traj_new x0 u x a xnew unew cnew csum begin
        check(@ccall libddp.ddp_forward_pass_f64(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, Kp::Ptr{Float64}, kp::Ptr{Float64},
            x0::Ptr{Float64}, u::Ptr{Float64}, xp::Ptr{Float64}, a::Ptr{Float64}, 1::Cint, lp::Ptr{Float64},
            xnew::Ptr{Float64}, unew::Ptr{Float64}, cnew::Ptr{Float64}, csum::Ptr{Float64})::Cint)
    end
    return xnew, unew, cnew
end

"""
    iLQG(problem, x0, u0; lims=[], α=..., tol_fun=1e-7, ...) -> x, u, traj_new, Vx, Vxx, cost, trace

Device-resident solve for a registered problem family (same keyword arguments and defaults as
src/iLQG.jl:143-163).  Arbitrary closures `f/costfun/df` keep the reference's own `iLQG` loop and
offload only `back_pass` (see INTEGRATION.md).
"""
function iLQG(problem, x0, u0; lims=[], α=exp10.(range(0, stop=-3, length=11)), tol_fun=1e-7, tol_grad=1e-4,
              max_iter=500, λ=1.0, dλ=1.0, λfactor=1.6, λmax=1e10, λmin=1e-6, regType=1, reduce_ratio_min=0.0,
              handle=default_handle(), kwargs...)
    m, N = size(u0); n = size(x0, 1)
    P = cproblem(problem, N, 1)
    CL = problem isa PendcartProblem ? N + 1 : N
    al = ntuple(i -> i <= length(α) ? Float64(α[i]) : 0.0, 16)
    o = ILQGOpts(λ, dλ, λfactor, λmax, λmin, tol_fun, tol_grad, max_iter, regType, reduce_ratio_min, length(α), al)
    x = zeros(n, N); u = zeros(m, N); K = zeros(m, n, N); k = zeros(m, N); Quu = zeros(m, m, N)
    Vx = zeros(n, N); Vxx = zeros(n, n, N); cost = zeros(CL); stats = zeros(8)
    cap = 4max_iter + 64; tr = zeros(cap); git = Ref{Cint}(0)
    x0 = _f64(vec(x0)); u0 = _f64(u0)
    lp = isempty(lims) ? Ptr{Float64}(C_NULL) : pointer(_f64(lims))
    tcap = 4max_iter + 1000; timing = fill(NaN, tcap, 3)     # C layout [3, tcap]: column r of the Julia array is row r
    GC.@preserve problem x0 u0 timing begin
        check(@ccall libddp.ddp_ilqg_set_timing(handle.ptr::Ptr{Cvoid}, timing::Ptr{Float64}, tcap::Cint)::Cint)
        check(@ccall libddp.ddp_ilqg_f64(handle.ptr::Ptr{Cvoid}, Ref(P)::Ptr{CProblem}, Ref(o)::Ptr{ILQGOpts},
            x0::Ptr{Float64}, u0::Ptr{Float64}, lp::Ptr{Float64}, x::Ptr{Float64}, u::Ptr{Float64}, K::Ptr{Float64},
            k::Ptr{Float64}, Quu::Ptr{Float64}, Vx::Ptr{Float64}, Vxx::Ptr{Float64}, cost::Ptr{Float64},
            stats::Ptr{Float64}, cap::Cint, tr::Ptr{Float64}, git::Ptr{Cint})::Cint)
        @ccall libddp.ddp_ilqg_set_timing(handle.ptr::Ptr{Cvoid}, C_NULL::Ptr{Float64}, 0::Cint)::Cint
    end
    stats[1] == -1 && return nothing                       # EXIT: Initial control sequence caused divergence
    g = Int(git[])
    trace = Dict(:cost => tr[1:max(Int(stats[2]) - 1, 0)], :λ => stats[6], :grad_norm => stats[7], :status => Int(stats[1]),
                 :time_derivs => timing[1:g, 1], :time_backward => timing[1:g, 2], :time_forward => timing[1:g, 3])   # iLQG.jl:227,241,281
    return x, u, GaussianPolicy(N, n, m, K, k, zeros(m, m, N), Quu), Vx, Vxx, cost, trace
end

"""
    mpc_shift(a, shift=1; zero_tail=false)

Receding-horizon warm start between two solves: `out[:, i] = a[:, i+shift]` along the time axis (the last axis of `u[m,N]`,
`x[n,N]`, `K[m,n,N]`), the vacated tail repeats the last column or is zero.  Host arrays; device-resident loops call
`ddp_mpc_shift_f64_dev` on their buffers instead.
"""
function mpc_shift(a::AbstractArray, shift::Integer=1; zero_tail::Bool=false)
    N = size(a, ndims(a)); out = similar(a)
    for i in 1:N
        src = i + shift
        selectdim(out, ndims(a), i) .= src <= N ? selectdim(a, ndims(a), src) : (zero_tail ? zero(eltype(a)) : selectdim(a, ndims(a), N))
    end
    out
end

# ---- KL-constrained path (src/backward_pass.jl:259-350, src/klutils.jl, src/forward_pass.jl:37-56) -------------------
struct KLCostTerms
    cx::Ptr{Float64}; cu::Ptr{Float64}; cxx::Ptr{Float64}; cxu::Ptr{Float64}; cuu::Ptr{Float64}; eta::Ptr{Float64}; eta_tv::Cint
end

"""
    ∇kl(traj_prev) -> cx, cu, cxx, cxu, cuu          (klutils.jl:8-23; cxu is m×n×T like the reference)
"""
function ∇kl(traj_prev::GaussianPolicy; handle=default_handle())
    isempty(traj_prev) && return (0, 0, 0, 0, 0)
    m, n, T = traj_prev.m, traj_prev.n, traj_prev.T
    cx, cu, cxx, cxu, cuu = zeros(n, T), zeros(m, T), zeros(n, n, T), zeros(m, n, T), zeros(m, m, T)
    K, k, Σi = traj_prev.K, traj_prev.k, traj_prev.Σi
    GC.@preserve K k Σi cx cu cxx cxu cuu begin
        check(@ccall libddp.ddp_kl_terms_f64(handle.ptr::Ptr{Cvoid}, n::Cint, m::Cint, T::Cint, 1::Cint, K::Ptr{Float64},
            k::Ptr{Float64}, Σi::Ptr{Float64}, cx::Ptr{Float64}, cu::Ptr{Float64}, cxx::Ptr{Float64}, cxu::Ptr{Float64},
            cuu::Ptr{Float64})::Cint)
    end
    return cx, cu, cxx, cxu, cuu
end

"""
    back_pass_gps(cx,cu,cxx,cxu,cuu,fx,fu,lims,x,u,kl_cost_terms) -> diverge, GaussianPolicy(N,n,m,K,k,Quui,Quu), Vx, Vxx, dV

Same signature and return values as backward_pass.jl:259; `kl_cost_terms = (∇kl(traj_prev), ηbracket)` with `ηbracket`
a 3-vector or a 3×N matrix.
"""
function back_pass_gps(cx, cu, cxx, cxu, cuu, fx, fu, lims, x, u, kl_cost_terms; handle=default_handle())
    n, N = size(cx); m = size(cu, 1)
    cx, cu, cxx, cxu, cuu, fx, fu, u = map(_f64, (cx, cu, cxx, cxu, cuu, fx, fu, u))
    cxkl, cukl, cxxkl, cxukl, cuukl = map(_f64, kl_cost_terms[1])
    ηb = kl_cost_terms[2]
    η = isa(ηb, AbstractMatrix) ? _f64(ηb[2, :]) : [Float64(ηb[2])]
    has_lims = !isempty(lims)
    limsp = has_lims ? _f64(lims) : Float64[]
    d = BPDesc(n, m, N, 1, 1, 0, 1, 0, 1, has_lims)
    K = zeros(m, n, N); k = zeros(m, N); Quu = zeros(m, m, N); Quui = zeros(m, m, N); Vx = zeros(n, N); Vxx = zeros(n, n, N)
    dV = zeros(2); diverge = Ref{Int32}(0)
    GC.@preserve cx cu cxx cxu cuu fx fu u cxkl cukl cxxkl cxukl cuukl η limsp K k Quu Quui Vx Vxx dV begin
        t = KLCostTerms(pointer(cxkl), pointer(cukl), pointer(cxxkl), pointer(cxukl), pointer(cuukl), pointer(η), isa(ηb, AbstractMatrix))
        check(@ccall libddp.ddp_back_pass_gps_f64(handle.ptr::Ptr{Cvoid}, Ref(d)::Ptr{BPDesc}, cx::Ptr{Float64}, cu::Ptr{Float64},
            cxx::Ptr{Float64}, cxu::Ptr{Float64}, cuu::Ptr{Float64}, fx::Ptr{Float64}, fu::Ptr{Float64}, Ref(t)::Ptr{KLCostTerms},
            (has_lims ? pointer(limsp) : Ptr{Float64}(C_NULL))::Ptr{Float64}, (has_lims ? pointer(u) : Ptr{Float64}(C_NULL))::Ptr{Float64},
            K::Ptr{Float64}, k::Ptr{Float64}, Quu::Ptr{Float64}, Quui::Ptr{Float64}, Vx::Ptr{Float64}, Vxx::Ptr{Float64},
            dV::Ptr{Float64}, diverge::Ptr{Int32})::Cint)
    end
    return Int(diverge[]), GaussianPolicy(N, n, m, K, k, Quui, Quu), Vx, Vxx, dV
end

"""
    forward_covariance(fx, R1, traj) -> sigmanew       (forward_pass.jl:37-56 with `df(model,·)[1]`, `covariance(model,·)` passed in)
"""
function forward_covariance(fx::Array{Float64,3}, R1::Matrix{Float64}, traj::GaussianPolicy; handle=default_handle())
    n, m, N = traj.n, traj.m, traj.T
    S = zeros(n + m, n + m, N); K, Σ = traj.K, traj.Σ
    GC.@preserve fx R1 K Σ S begin
        check(@ccall libddp.ddp_forward_covariance_f64(handle.ptr::Ptr{Cvoid}, n::Cint, m::Cint, N::Cint, 1::Cint, fx::Ptr{Float64},
            0::Cint, R1::Ptr{Float64}, K::Ptr{Float64}, Σ::Ptr{Float64}, S::Ptr{Float64})::Cint)
    end
    return S
end

"""
    kl_div_wiki(xnew, xold, Σ_new, traj_new, traj_prev) -> kldiv (or Inf)      (klutils.jl:70-103)
"""
function kl_div_wiki(xnew, xold, Σ_new, traj_new::GaussianPolicy, traj_prev::GaussianPolicy; handle=default_handle())
    n, m, T = traj_new.n, traj_new.m, traj_new.T
    kld = zeros(T); mean_ = zeros(1)
    xnew, xold, Σ_new = map(_f64, (xnew, xold, Σ_new))
    Kn, kn, Σn, Kp, kp, Σp, Σip = traj_new.K, traj_new.k, traj_new.Σ, traj_prev.K, traj_prev.k, traj_prev.Σ, traj_prev.Σi
    GC.@preserve xnew xold Σ_new Kn kn Σn Kp kp Σp Σip kld mean_ begin
        check(@ccall libddp.ddp_kl_div_f64(handle.ptr::Ptr{Cvoid}, n::Cint, m::Cint, T::Cint, 1::Cint, xnew::Ptr{Float64},
            xold::Ptr{Float64}, Σ_new::Ptr{Float64}, Kn::Ptr{Float64}, kn::Ptr{Float64}, Σn::Ptr{Float64}, Kp::Ptr{Float64},
            kp::Ptr{Float64}, Σp::Ptr{Float64}, Σip::Ptr{Float64}, kld::Ptr{Float64}, mean_::Ptr{Float64})::Cint)
    end
    return isinf(mean_[1]) && all(isfinite, kld) ? Inf : kld
end
# calc_η, geom and the iLQGkl loop stay the reference's scalar Julia code (klutils.jl:112-155, iLQGkl.jl): with the four
# functions above rebound, src/iLQGkl.jl:90,100,133 and klutils.jl:114 run on the GPU unchanged.

end # module
