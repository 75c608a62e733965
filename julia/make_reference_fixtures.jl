#!/usr/bin/env julia
# make_reference_fixtures.jl — runs the REAL reference (baggepinnen/DifferentialDynamicProgramming.jl v0.5.0) on the inputs of
# tests/golden/*.npz and writes its outputs to tests/golden/julia/, which turns "parity unpinned" into pinned:
#
#     python tests/golden/rawio.py                                   # (already committed) inputs as raw f64: tests/golden/raw/
#     julia --project=/path/to/DifferentialDynamicProgramming.jl julia/make_reference_fixtures.jl
#     python -m pytest tests/test_julia_fixtures.py -q               # oracle vs Julia (CPU);  add  -m gpu  for HIP vs Julia
#
# NOT EXECUTED in this repository's build image (no Julia toolchain, no network): written against the reference's source,
# Julia >= 1.5, no packages beyond the reference's own dependencies.  File format: tests/golden/rawio.py (one <case>.bin of
# column-major little-endian arrays + a whitespace-separated manifest.txt).
#
# Calls made (the functions of SURVEY.md §8a):
#   bp_*        DifferentialDynamicProgramming.back_pass(cx,cu,cxx,cxu,cuu,fx,fu,λ,regType,lims,x,u)   src/backward_pass.jl:162,179,217
#   boxqp       boxQP(H,g,lower,upper,x0)                                                                src/boxQP.jl:29
#   fwd_*       forward_pass(traj_new,x0,u,x,α,f,costfun,lims,-)                                         src/forward_pass.jl:9
#   df_pendcart the `df` closure of demo_pendcart (restated below from src/system_pendcart.jl:125-154)
#   ilqg_*      iLQG(f,costfun,df,x0,u0; ...)                                                            src/iLQG.jl:143
#   kl_gps_*    ∇kl, back_pass_gps, forward_covariance, kl_div_wiki       src/klutils.jl:8,70; backward_pass.jl:259; forward_pass.jl:37
#               + calc_η on the fixture's own divergence (four step sizes around its mean)               src/klutils.jl:110-133
#   kl_ilqgkl_* iLQGkl(dynamics,costfun,derivs,x0,traj_prev,model; kl_step, cost, ...)                    src/iLQGkl.jl:25-178
#   ilqg_warm_* iLQG with a PRE-ROLLED x0[n,N] and its cost (the warm start of an MPC loop)               src/iLQG.jl:193-197
#   ilqg_trace_* iLQG + every per-iteration trace key (:λ :dλ :α :improvement :cost :reduce_ratio :grad_norm)  src/iLQG.jl:257,325-330
using LinearAlgebra, Printf
using DifferentialDynamicProgramming
const DDP = DifferentialDynamicProgramming

const ROOT = length(ARGS) >= 1 ? ARGS[1] : normpath(joinpath(@__DIR__, ".."))
const RAW = joinpath(ROOT, "tests", "golden", "raw")
const OUT = joinpath(ROOT, "tests", "golden", "julia")

# ------------------------------------------------------------------------------------------- raw format
struct Entry
    case::String
    key::String
    dtype::String
    dims::Vector{Int}
    offset::Int
end

function read_manifest(path)
    entries = Entry[]
    for line in eachline(path)
        s = strip(line)
        (isempty(s) || startswith(s, "#")) && continue
        t = split(s)
        nd = parse(Int, t[4])
        dims = Int[parse(Int, t[4+i]) for i in 1:nd]
        push!(entries, Entry(String(t[1]), String(t[2]), String(t[3]), dims, parse(Int, t[5+nd])))
    end
    return entries
end

function load_case(entries, case)
    d = Dict{String,Any}()
    open(joinpath(RAW, case * ".bin")) do io
        for e in entries
            e.case == case || continue
            seek(io, e.offset)
            T = e.dtype == "i64" ? Int64 : Float64
            a = Array{T}(undef, e.dims...)
            read!(io, a)
            d[e.key] = isempty(e.dims) ? a[] : a
        end
    end
    return d
end

mutable struct Writer
    io::IOStream
    man::IOStream
    case::String
    off::Int
end

function emit(w::Writer, key::String, a)
    arr = if a isa Integer || a isa Bool
        fill(Int64(a))
    elseif a isa Number
        fill(Float64(a))
    elseif eltype(a) <: Integer || eltype(a) == Bool
        Array{Int64}(a)
    else
        Array{Float64}(a)
    end
    dt = eltype(arr) == Int64 ? "i64" : "f64"
    println(w.man, join(Any[w.case, key, dt, ndims(arr), size(arr)..., w.off], " "))
    write(w.io, arr)
    w.off += sizeof(arr)
    return nothing
end

lims_of(c) = (haskey(c, "lims") && !isempty(c["lims"])) ? c["lims"] : []

# ------------------------------------------------------------------------------------------- problem closures
# LQ family: the closures of src/demo_linear.jl:30-50 / test/test_readme.jl:34-55
function lq_closures(A, B, Q, R)
    cxu = zeros(size(B))
    f(x, u, i) = (u[isnan.(u)] .= 0; A * x + B * u)
    costfun(x, u) = 0.5 * sum(x .* (Q * x)) + 0.5 * sum(u .* (R * u))
    costvec(x, u) = vec(0.5 * sum(x .* (Q * x), dims=1) + 0.5 * sum(u .* (R * u), dims=1))     # per-step terms of the same sum
    function df(x, u)
        u[isnan.(u)] .= 0
        return A, B, [], [], [], Q * x, R * u, Q, cxu, R
    end
    return f, costfun, costvec, df
end

# pendulum on a cart: src/system_pendcart.jl:42-59 (parameters), :83-89 (Euler step), :92-116 (cost), :125-154 (derivatives)
function pendcart_closures(T; g=9.82, l=0.35, h=0.01, d=0.99, Q=Matrix(Diagonal([10.0, 1, 2, 1])), R=1.0, goal=[π, 0, 0, 0])
    function f(x, u, i)
        u[isnan.(u)] .= 0
        return [x[1] + h * x[2],
                x[2] + h * (-g / l * sin(x[1]) + u[1] / l * cos(x[1]) - d * x[2]),
                x[3] + h * x[4],
                x[4] + h * u[1]]
    end
    function costfun(x::AbstractMatrix, u)
        dx = x .- goal
        N = size(u, 2)
        c = Vector{Float64}(undef, N + 1)
        for t in 1:N
            c[t] = 0.5 * (dx[:, t]' * Q * dx[:, t] + u[:, t]' * R * u[:, t])[1]
        end
        c[end] = 0.5 * (dx[:, end]' * Q * dx[:, end])[1]
        return c
    end
    function df(x, u)
        u[isnan.(u)] .= 0
        n, N = size(x, 1), size(u, 2)
        cx = Q * (x .- goal)
        cu = R .* u
        fxd = Array{Float64}(undef, n, n, N)
        fud = Array{Float64}(undef, n, 1, N)
        for i in 1:N
            Ac = [0 1 0 0; (-g / l * cos(x[1, i]) - u[i] / l * sin(x[1, i])) -d 0 0; 0 0 0 1; 0 0 0 0]
            Bc = [0, cos(x[1, i]) / l, 0, 1]
            ABd = exp([Ac * h  Bc * h; zeros(1, n + 1)])            # zero-order-hold sampling
            fxd[:, :, i] = ABd[1:n, 1:n]
            fud[:, :, i] = ABd[1:n, n+1:n+1]
        end
        return fxd, fud, [], [], [], cx, cu, Q, zeros(n, 1), fill(R, 1, 1)
    end
    return f, costfun, df
end

policy(K, k) = GaussianPolicy(size(k, 2), size(K, 2), size(K, 1), K, k, zeros(size(K, 1), size(K, 1), size(k, 2)), zeros(size(K, 1), size(K, 1), size(k, 2)))

# ------------------------------------------------------------------------------------------- families
function run_bp(w, c)
    diverge, traj, Vx, Vxx, dV = DDP.back_pass(c["cx"], c["cu"], c["cxx"], c["cxu"], c["cuu"], c["fx"], c["fu"], c["lam"],
                                               Int(c["regType"]), lims_of(c), c["x"], c["u"])
    emit(w, "diverge", Int(diverge)); emit(w, "K", traj.K); emit(w, "k", traj.k)
    emit(w, "Quu", traj.Σi)                     # entries before a failing step are uninitialised memory upstream (`undef`)
    emit(w, "Vx", Vx); emit(w, "Vxx", Vxx); emit(w, "dV", dV)
end

function run_boxqp(w, c)
    cnt = length(c["m"])
    X = zeros(cnt, 8); res = zeros(Int, cnt); FR = zeros(cnt, 8); HF = zeros(cnt, 8, 8)
    for t in 1:cnt
        m = Int(c["m"][t])
        x, result, Hfree, free, _ = boxQP(c["H"][t, 1:m, 1:m], c["g"][t, 1:m], c["lower"][t, 1:m], c["upper"][t, 1:m], c["x0"][t, 1:m])
        X[t, 1:m] = x; res[t] = result; FR[t, 1:m] = free
        HF[t, 1:size(Hfree, 1), 1:size(Hfree, 2)] = Hfree
    end
    emit(w, "x", X); emit(w, "result", res); emit(w, "free", FR); emit(w, "Hfree", HF)
end

function run_fwd(w, c, f, costfun)
    al = c["alphas"]
    n, N = size(c["x"]); m = size(c["u"], 1)
    first_c = costfun(c["x"], c["u"])
    xs = zeros(n, N, length(al)); us = zeros(m, N, length(al)); cs = zeros(length(first_c), length(al))
    for (j, a) in enumerate(al)
        xn, un, cn = DDP.forward_pass(policy(c["K"], c["k"]), vec(c["x0"]), copy(c["u"]), c["x"], a, f, costfun, lims_of(c), -)
        xs[:, :, j] = xn; us[:, :, j] = un; cs[:, j] .= cn
    end
    emit(w, "xnew", xs); emit(w, "unew", us); emit(w, "cnew", cs)
end

function run_ilqg(w, f, costfun, df, x0, u0; kwargs...)
    r = iLQG(f, costfun, df, x0, u0; verbosity=0, plot=0, kwargs...)
    if r === nothing
        emit(w, "status", -1)
        return
    end
    x, u, L, Vx, Vxx, cost, trace = r
    emit(w, "x", x); emit(w, "u", u); emit(w, "K", L.K); emit(w, "k", L.k); emit(w, "Quu", L.Σi)
    emit(w, "Vx", Vx); emit(w, "Vxx", Vxx); emit(w, "cost", cost isa Number ? [cost] : vec(cost))
    its, tc = get(trace, :cost)
    emit(w, "tr_cost", collect(Float64, tc)); emit(w, "iter", length(tc) + 1)
    # the other per-iteration keys (iLQG.jl:257 :grad_norm, :325-330); a key that a given version of the reference does not push is skipped
    for (sym, key) in ((:λ, "tr_lambda"), (:dλ, "tr_dlambda"), (:α, "tr_alpha"), (:improvement, "tr_improvement"),
                       (:reduce_ratio, "tr_reduce_ratio"), (:grad_norm, "tr_grad_norm"))
        try
            _, v = get(trace, sym)
            emit(w, key, collect(Float64, v))
        catch err
            @warn "trace key $sym not recorded" err
        end
    end
end

# the model argument of iLQGkl / forward_covariance: LinearTimeVaryingModelsBase is un-vendored (SURVEY §8c); a fixture model that hands
# back given arrays pins the arithmetic of everything downstream of it
function define_fixture_model()
    @eval import LinearTimeVaryingModelsBase
    @eval LinearTimeVaryingModelsBase.df(mo::FixtureModel, x, u) = (mo.fx, mo.fu, [], [], [])
    @eval LinearTimeVaryingModelsBase.covariance(mo::FixtureModel, x, u) = mo.R1
end

function run_ilqgkl(w, c)
    A, B, Q, R = c["A"], c["B"], c["Q"], c["R"]
    n, m = size(B); T = size(c["u"], 2)
    f, costfun, costvec, _ = lq_closures(A, B, Q, R)
    fx = repeat(A, 1, 1, T); fu = repeat(B, 1, 1, T)
    cxx = repeat(Q, 1, 1, T); cuu = repeat(R, 1, 1, T); cxu = zeros(n, m, T)
    derivs(x, u) = (fx, fu, [], [], [], Q * x, R * u, cxx, cxu, cuu)                 # the 10-tuple of iLQGkl.jl:88
    eyeT = repeat(Matrix{Float64}(I, m, m), 1, 1, T)
    prev = GaussianPolicy(T, n, m, zeros(m, n, T), copy(c["u"]), copy(eyeT), copy(eyeT))
    define_fixture_model()
    model = FixtureModel(fx, fu, c["R1"])
    r = Base.invokelatest(iLQGkl, f, costvec, derivs, c["x"], prev, model; kl_step=c["kl_step"], cost=c["cost0"], max_iter=50, verbosity=0)
    x, u, L, Vx, Vxx, cost, trace = r
    emit(w, "xnew", x); emit(w, "unew", u); emit(w, "K", L.K); emit(w, "S", L.Σ); emit(w, "Si", L.Σi)
    emit(w, "Vx", Vx); emit(w, "Vxx", Vxx); emit(w, "cost", cost isa Number ? [cost] : vec(cost))
    for (sym, key) in ((:η, "eta_trace"), (:divergence, "divergence_trace"))
        try
            _, v = get(trace, sym)
            emit(w, key, collect(Float64, v))
        catch err
            @warn "trace key $sym not recorded" err
        end
    end
end

# forward_covariance asks the model for df(model,x,u) and covariance(model,x,u) (LinearTimeVaryingModelsBase, un-vendored): a
# fixture model that returns given arrays
struct FixtureModel
    fx::Array{Float64,3}
    fu::Array{Float64,3}
    R1::Matrix{Float64}
end

function run_gps(w, c)
    Kp, kp, Sip, Sp = c["Kp"], c["kp"], c["Sip"], c["Sp"]
    N, n, m = size(kp, 2), size(Kp, 2), size(Kp, 1)
    prev = GaussianPolicy(N, n, m, Kp, kp, Sp, Sip)
    terms = DDP.∇kl(prev)
    for (key, a) in zip(("cxkl", "cukl", "cxxkl", "cxukl", "cuukl"), terms)
        emit(w, key, a)
    end
    etab = c["etab"]
    diverge, traj, Vx, Vxx, dV = DDP.back_pass_gps(c["cx"], c["cu"], c["cxx"], c["cxu"], c["cuu"], c["fx"], c["fu"], lims_of(c), c["x"],
                                                   c["u"], (terms, etab))
    emit(w, "diverge", Int(diverge)); emit(w, "K", traj.K); emit(w, "k", traj.k); emit(w, "Quui", traj.Σ); emit(w, "Quu", traj.Σi)
    emit(w, "Vx", Vx); emit(w, "Vxx", Vxx); emit(w, "dV", dV)
    diverge == 0 || return
    try
        @eval import LinearTimeVaryingModelsBase
        @eval LinearTimeVaryingModelsBase.df(mo::FixtureModel, x, u) = (mo.fx, mo.fu, [], [], [])
        @eval LinearTimeVaryingModelsBase.covariance(mo::FixtureModel, x, u) = mo.R1
        sig = Base.invokelatest(DDP.forward_covariance, FixtureModel(c["fx"], c["fu"], c["R1"]), c["x"], c["u"], traj)
        sig[isnan.(sig)] .= 0                   # the u-blocks of the last step are never written upstream
        emit(w, "sigmanew", sig)
        kld = DDP.kl_div_wiki(c["xnew"], c["x"], sig, traj, prev)
        emit(w, "kldiv", kld isa Number ? fill(Float64(kld), N) : kld)
        # calc_η (klutils.jl:110-133), scalar kl_step, on this divergence: η too big / converged / η too small / just outside the 10 % band
        if ndims(etab) == 1
            dbar = sum(kld) / length(kld)
            steps = [2.0, 1.0, 0.5, 1.0 / 0.85] .* dbar
            eo = zeros(3, length(steps)); sat = zeros(Int, length(steps)); dv = zeros(length(steps))
            for (j, st) in enumerate(steps)
                e2, s2, d2 = DDP.calc_η(c["xnew"], c["x"], sig, copy(vec(etab)), traj, prev, st)
                eo[:, j] = e2; sat[j] = s2 ? 1 : 0; dv[j] = d2
            end
            emit(w, "eta_kl_steps", steps); emit(w, "eta_out", eo); emit(w, "eta_satisfied", sat); emit(w, "eta_divergence", dv)
        end
    catch err
        @warn "forward_covariance / kl_div_wiki skipped" err
    end
end

# ------------------------------------------------------------------------------------------- main
function main()
    isfile(joinpath(RAW, "manifest.txt")) || error("run `python tests/golden/rawio.py` first (no $(RAW)/manifest.txt)")
    entries = read_manifest(joinpath(RAW, "manifest.txt"))
    cases = unique(e.case for e in entries)
    mkpath(OUT)
    man = open(joinpath(OUT, "manifest.txt"), "w")
    println(man, "# outputs of DifferentialDynamicProgramming.jl (julia $(VERSION)) on tests/golden/raw; format: tests/golden/rawio.py")
    for case in cases
        c = load_case(entries, case)
        io = open(joinpath(OUT, case * ".bin"), "w")
        w = Writer(io, man, case, 0)
        try
            if startswith(case, "bp_")
                run_bp(w, c)
            elseif case == "boxqp"
                run_boxqp(w, c)
            elseif startswith(case, "fwd_lq")
                f, costfun, costvec, _ = lq_closures(c["A"], c["B"], c["Q"], c["R"])
                run_fwd(w, c, f, costvec)           # per-step cost terms; their sum is the reference's scalar (quirk Q15)
            elseif case == "fwd_pendcart"
                f, costfun, _ = pendcart_closures(size(c["u"], 2))
                run_fwd(w, c, f, costfun)
            elseif case == "df_pendcart"
                _, _, df = pendcart_closures(size(c["u"], 2))
                fx, fu, _, _, _, cx, cu, _, _, _ = df(c["x"], copy(c["u"]))
                emit(w, "fx", fx); emit(w, "fu", fu); emit(w, "cx", cx); emit(w, "cu", cu)
            elseif case == "ilqg_lq_n10m2"
                f, costfun, costvec, df = lq_closures(c["A"], c["B"], c["Q"], c["R"])
                run_ilqg(w, f, costvec, df, reshape(c["x0"], :, 1), c["u0"])
            elseif case == "ilqg_pendcart"
                T = Int(c["T"])
                f, costfun, df = pendcart_closures(T)
                run_ilqg(w, f, costfun, df, reshape(c["x0"], :, 1), zeros(1, T); lims=5.0 * [-1 1], regType=2,
                         α=exp10.(range(0.2, stop=-3, length=6)), λmax=1e15, tol_fun=1e-8, tol_grad=1e-8, max_iter=1000)   # system_pendcart.jl:197-206
            elseif startswith(case, "kl_gps_")
                run_gps(w, c)
            elseif startswith(case, "kl_ilqgkl_")
                run_ilqgkl(w, c)
            elseif case == "ilqg_trace_lq"
                f, costfun, costvec, df = lq_closures(c["A"], c["B"], c["Q"], c["R"])
                run_ilqg(w, f, costvec, df, reshape(c["x0"], :, 1), c["u0"])
            elseif case == "ilqg_warm_lq"
                f, costfun, costvec, df = lq_closures(c["A"], c["B"], c["Q"], c["R"])
                run_ilqg(w, f, costvec, df, c["x0"], c["u0"]; cost=c["cost0"])          # size(x0, 2) == N: pre-rolled (iLQG.jl:193-197)
            else
                @warn "no runner for case $case"
            end
            @printf("%-32s ok\n", case)
        catch err
            @printf("%-32s FAILED: %s\n", case, sprint(showerror, err))
        end
        close(io)
    end
    close(man)
    println("wrote ", OUT)
end

main()
