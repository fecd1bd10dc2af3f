# BLRMI355X.jl -- the Julia side of the drop-in boundary (NOT runnable in the build image: no `julia`).
#
# Kept deliberately thin so it is correct by inspection: every behaviour it relies on is exercised through the same C ABI
# by the Python/ctypes harness (tests/test_gpu_parity.py), and tests/test_abi_cpu.py keeps every ccall below in step with
# include/blr_mi355x.h (argument count and kind per position).
#
# What it offloads, for FiniteGP{<:BayesianLinearRegressor} / FiniteGP{<:BasisFunctionRegressor} with Float64 / Float32 data,
# ColVecs / RowVecs inputs, Diagonal (incl. Fill) or dense observation noise:
#     logpdf(fx, y), logpdf(fx, Y::Matrix), posterior(fx, y), mean, var, cov, mean_and_var, mean_and_cov, marginals,
#     rand(rng, fx[, S]), rand(rng, f[, dims]), rand!(rng, A, f), Random.Sampler          (reference src/*.jl, cited per function)
# plus what the reference gets from Zygote and a ccall cannot give by itself:
#     ChainRulesCore.rrule(logpdf, fx, y)                                                     (README.md:56-71, examples/nn-blr.jl:35-37)
# and what only a device library can offer:
#     maps over collections of problems in one call (logpdf_map, posterior_map), device-resident batches (posterior_batched!), the RCCL exchange for one Julia process per GPU (comm_init!, logpdf_allgather_sum!).
# Anything else (exotic element types, unknown containers) falls back to the reference's own CPU methods.
#
# Two ways to use it -- both spelled out in INTEGRATION.md:
#   opt-in, side by side :  using .BLRMI355X;  BLRMI355X.logpdf(fx, y)
#   drop-in              :  BLRMI355X.install_overrides!()  replaces the reference's methods on FiniteGP{<:BayesianLinearRegressor}
#                           (src/bayesian_linear_regression.jl:33-69) by the offloaded ones for the whole session.
module BLRMI355X

using AbstractGPs, LinearAlgebra, PDMats, Random
using AbstractGPs: FiniteGP
using BayesianLinearRegressors: BayesianLinearRegressor, BasisFunctionRegressor, BLRFunctionSample
import BayesianLinearRegressors as REF
import ChainRulesCore
using ChainRulesCore: NoTangent, Tangent, ZeroTangent

const LIB = get(ENV, "BLR_MI355X_LIB", "libblr_mi355x")

# enums of include/blr_mi355x.h
const COLVECS, ROWVECS = Cint(0), Cint(1)
const ISOTROPIC, DIAGONALN, DENSEN = Cint(0), Cint(1), Cint(2)
const P_DENSE, P_UPPER, P_DIAG = Cint(0), Cint(1), Cint(2)
const MEM_HOST, MEM_DEVICE = Cint(0), Cint(1)

const Elt = Union{Float32,Float64}
const FiniteBLR = FiniteGP{<:Union{BayesianLinearRegressor,BasisFunctionRegressor}}

# ---- handle: one per task ------------------------------------------------------------------------------
const HANDLE_KEY = :blr_mi355x_handle
function handle()
    get!(task_local_storage(), HANDLE_KEY) do
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:blr_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), parse(Cint, get(ENV, "BLR_MI355X_DEVICE", "0")), h)
        rc == 0 || error("blr_create failed with code $rc (no MI355X visible?)")
        h[]
    end::Ptr{Cvoid}
end

function check(h, rc)
    rc == 0 && return nothing
    rc > 0 && throw(PosDefException(rc))                       # what cholesky at reference :78/:79/:86 throws
    msg = unsafe_string(ccall((:blr_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    rc > -1000 ? throw(DimensionMismatch("libblr_mi355x: $msg")) : error("libblr_mi355x (HIP/RCCL): $msg")
end

# ---- x_as_colvecs (reference :20-31) as (array, layout, ld, D, N): zero copies -----------------------------
xlayout(x::ColVecs{T,<:StridedMatrix{T}}) where {T<:Elt} = (x.X, COLVECS, stride(x.X, 2), size(x.X, 1), size(x.X, 2))
xlayout(x::RowVecs{T,<:StridedMatrix{T}}) where {T<:Elt} = (x.X, ROWVECS, stride(x.X, 2), size(x.X, 2), size(x.X, 1))
xlayout(x) = nothing  # anything else: not offloadable (the reference raises its own error for unknown containers)

# observation noise -> (buffer, kind, leading dimension)
noise(Σ::Diagonal{T,<:AbstractGPs.FillArrays.Fill}) where {T<:Elt} = (T[Σ.diag.value], ISOTROPIC, 1)
noise(Σ::Diagonal{T,<:StridedVector{T}}) where {T<:Elt} = (Σ.diag, DIAGONALN, 1)
noise(Σ::StridedMatrix{T}) where {T<:Elt} = (Σ, DENSEN, stride(Σ, 2))                  # reference :79-82 general branch
noise(Σ::Symmetric{T,<:StridedMatrix{T}}) where {T<:Elt} = (Matrix(Σ), DENSEN, size(Σ, 1))
noise(Σ) = nothing

prior(Λ::Diagonal{T}) where {T<:Elt} = (collect(Λ.diag), P_DIAG, 1)
prior(Λ::PDMat{T}) where {T<:Elt} = (Matrix(Λ.chol.U), P_UPPER, size(Λ, 1))          # factor carried forward (:93)
prior(Λ::Symmetric{T,<:StridedMatrix{T}}) where {T<:Elt} = (Matrix(Λ), P_DENSE, size(Λ, 1))
prior(Λ::StridedMatrix{T}) where {T<:Elt} = (Λ, P_DENSE, stride(Λ, 2))
prior(Λ) = nothing

build_Λ(::Type{<:AbstractPDMat}, T, A) = PDMat(Cholesky(UpperTriangular(T)))           # reference :93
build_Λ(_, T, A) = Symmetric(A)                                                          # reference :92 (A == T'T)

to_blr(fx::FiniteGP{<:BayesianLinearRegressor}) = fx
to_blr(fx::FiniteGP{<:BasisFunctionRegressor}) = fx.f.blr(fx.f.ϕ(fx.x), fx.Σy)         # basis_function_regression.jl:41

# ---- random-Fourier basis on the device (BASELINE config 5) -----------------------------------------
# A BasisFunctionRegressor whose ϕ is  x -> sqrt(2/D) cos.(Ω'x .+ β)  (reference basis_function_regression.jl:41,62-65:
# bfr(x) = blr(ϕ(x))) hands the raw inputs to the device: the feature matrix is generated there and consumed by the same
# fused posterior/logpdf path, never crossing PCIe.
struct RandomFourierFeatures{T<:Elt}
    Ω::Matrix{T}      # Din x D
    β::Vector{T}      # D
end
(r::RandomFourierFeatures{T})(x::ColVecs) where {T} = ColVecs(convert(T, sqrt(2 / length(r.β))) .* cos.(r.Ω' * x.X .+ r.β))
(r::RandomFourierFeatures)(x::RowVecs) = r(ColVecs(permutedims(x.X)))

function fused_rff(blr::BayesianLinearRegressor, ϕ::RandomFourierFeatures{T}, x::ColVecs, Σy, y::AbstractVector, want_posterior::Bool) where {T}
    pr = prior(blr.Λw); nz = noise(Σy)
    (pr === nothing || nz === nothing || nz[2] == DENSEN) && return nothing
    Lw, pk, ldl = pr; s, nk, _ = nz
    Din, D = size(ϕ.Ω); N = length(y)
    Xin = convert(Matrix{T}, x.X); yv = convert(Vector{T}, y); mw = convert(Vector{T}, blr.mw)
    mw′ = want_posterior ? Vector{T}(undef, D) : Ptr{T}(C_NULL)
    Tm = want_posterior ? Matrix{T}(undef, D, D) : Ptr{T}(C_NULL)
    A = (want_posterior && pk != P_UPPER) ? Matrix{T}(undef, D, D) : Ptr{T}(C_NULL)
    lp = Ref{Cdouble}(0.0); info = Ref{Int32}(0)
    h = handle()
    rc = GC.@preserve Xin yv s Lw mw mw′ Tm A begin
        if T === Float64
            ccall((:blr_posterior_rff_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, T, Ptr{T}, Cint, Ptr{T},
                   Cint, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ref{Cdouble}, Ref{Int32}),
                  h, MEM_HOST, Din, D, N, Xin, Din, ϕ.Ω, Din, ϕ.β, sqrt(T(2) / D), yv, nk, s, pk, mw, Lw, ldl, mw′, Tm, D, A, D,
                  lp, info)
        else
            ccall((:blr_posterior_rff_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, T, Ptr{T}, Cint, Ptr{T},
                   Cint, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ref{Cdouble}, Ref{Int32}),
                  h, MEM_HOST, Din, D, N, Xin, Din, ϕ.Ω, Din, ϕ.β, sqrt(T(2) / D), yv, nk, s, pk, mw, Lw, ldl, mw′, Tm, D, A, D,
                  lp, info)
        end
    end
    check(h, rc)
    info[] > 0 && throw(PosDefException(info[]))
    return lp[], mw′, Tm, A
end

# ---- fused inference: reference :55-58, :60-69, :72-89 ------------------------------------------------------
function fused(fx::FiniteGP, y::AbstractVector{<:Real}, want_posterior::Bool)
    # BasisFunctionRegressor with the random-Fourier ϕ: features + inference in one library call
    if fx.f isa BasisFunctionRegressor && fx.f.ϕ isa RandomFourierFeatures && fx.x isa ColVecs
        r = fused_rff(fx.f.blr, fx.f.ϕ, fx.x, fx.Σy, y, want_posterior)
        r === nothing || return r
    end
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && return nothing
    X, layout, ldx, D, N = xl
    T = eltype(X)
    length(y) == N || throw(error("length(y) != size(fx.x.X, 2)"))                        # reference :74
    length(fb.f.mw) == D || throw(DimensionMismatch("length(mw) != dimension of the inputs"))
    yv, mw = convert(Vector{T}, y), convert(Vector{T}, fb.f.mw)
    s, nk, lds = nz
    Lw, pk, ldl = pr
    mw_post = want_posterior ? Vector{T}(undef, D) : Ptr{T}(C_NULL)
    Tp = want_posterior ? Matrix{T}(undef, D, D) : Ptr{T}(C_NULL)
    Ap = (want_posterior && pk != P_UPPER) ? Matrix{T}(undef, D, D) : Ptr{T}(C_NULL)
    lp = Ref{Cdouble}(0.0)
    h = handle()
    if nk == DENSEN                                                                        # reference :79-82, general Sigma_y
        info = Ref{Int32}(0)
        rc = GC.@preserve X yv mw s Lw mw_post Tp Ap begin
            if T === Float64
                ccall((:blr_posterior_dense_noise_f64, LIB), Cint,
                      (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Ptr{T}, Int64, Cint, Ptr{T}, Ptr{T}, Int64,
                       Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ref{Cdouble}, Ref{Int32}),
                      h, MEM_HOST, layout, D, N, X, ldx, yv, s, lds, pk, mw, Lw, ldl, mw_post, Tp, D, Ap, D, lp, info)
            else
                ccall((:blr_posterior_dense_noise_f32, LIB), Cint,
                      (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Ptr{T}, Int64, Cint, Ptr{T}, Ptr{T}, Int64,
                       Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ref{Cdouble}, Ref{Int32}),
                      h, MEM_HOST, layout, D, N, X, ldx, yv, s, lds, pk, mw, Lw, ldl, mw_post, Tp, D, Ap, D, lp, info)
            end
        end
        check(h, rc)
        check(h, info[])
        return lp[], mw_post, Tp, Ap
    end
    rc = GC.@preserve X yv mw s Lw mw_post Tp Ap begin
        if T === Float64
            ccall((:blr_posterior_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Cint, Ptr{T}, Cint, Ptr{T}, Ptr{T}, Int64,
                   Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ref{Cdouble}),
                  h, layout, D, N, X, ldx, yv, nk, s, pk, mw, Lw, ldl, mw_post, Tp, D, Ap, D, lp)
        else
            ccall((:blr_posterior_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Cint, Ptr{T}, Cint, Ptr{T}, Ptr{T}, Int64,
                   Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ref{Cdouble}),
                  h, layout, D, N, X, ldx, yv, nk, s, pk, mw, Lw, ldl, mw_post, Tp, D, Ap, D, lp)
        end
    end
    check(h, rc)
    return lp[], mw_post, Tp, Ap
end

function logpdf(fx::FiniteGP, y::AbstractVector{<:Real})
    r = fused(fx, y, false)
    r === nothing ? REF_logpdf(fx, y) : r[1]
end

function posterior(fx::FiniteGP, y::AbstractVector{<:Real})
    r = fused(fx, y, true)
    r === nothing && return REF_posterior(fx, y)
    _, mw_post, Tp, Ap = r
    blr0 = fx.f isa BasisFunctionRegressor ? fx.f.blr : fx.f
    post = BayesianLinearRegressor(mw_post, build_Λ(typeof(blr0.Λw), Tp, Ap))
    fx.f isa BasisFunctionRegressor ? BasisFunctionRegressor(post, fx.f.ϕ) : post           # :62-65
end

# ---- many equally shaped problems in ONE library call: `map(posterior, fxs, ys)` / `logpdf.(fxs, ys)` -------------------
# (BASELINE config 4 is this with 8192 regressors; at D > 128 the regressors share every launch of the update.)  Host arrays
# are packed side by side -- problem b at b * stride -- and handed to blr_posterior_batched_* with BLR_MEM_HOST.  Collections
# the batched entry point does not take (mixed shapes / layouts / kinds, dense noise, a fused random-Fourier basis) are
# mapped one by one.  The first problem that is not positive definite throws PosDefException, as the map would.
function fused_many(fxs::AbstractVector{<:FiniteGP}, ys::AbstractVector{<:AbstractVector{<:Real}}, want_posterior::Bool)
    length(fxs) == length(ys) || throw(DimensionMismatch("as many observation vectors as finite regressors are needed"))
    B = length(fxs)
    B == 0 && return Tuple[]
    any(fx -> fx.f isa BasisFunctionRegressor && fx.f.ϕ isa RandomFourierFeatures, fxs) && return nothing
    fbs = map(to_blr, fxs)
    xls, nzs, prs = map(fb -> xlayout(fb.x), fbs), map(fb -> noise(fb.Σy), fbs), map(fb -> prior(fb.f.Λw), fbs)
    (any(isnothing, xls) || any(isnothing, nzs) || any(isnothing, prs)) && return nothing
    X1, layout, _, D, N = xls[1]
    T = eltype(X1)
    nk, pk = nzs[1][2], prs[1][2]
    (nk == DENSEN || D == 0 || N == 0) && return nothing
    same = all(b -> eltype(xls[b][1]) === T && xls[b][2] == layout && xls[b][4] == D && xls[b][5] == N && nzs[b][2] == nk &&
                    prs[b][2] == pk && length(ys[b]) == N && length(fbs[b].f.mw) == D, 1:B)
    same || return nothing
    rows, cols = layout == COLVECS ? (D, N) : (N, D)
    Xb = Array{T}(undef, rows, cols, B)
    for b in 1:B
        copyto!(view(Xb, :, :, b), xls[b][1])
    end
    yb = Matrix{T}(undef, N, B)
    for b in 1:B
        yb[:, b] .= ys[b]
    end
    ns = nk == ISOTROPIC ? 1 : N
    sb = Matrix{T}(undef, ns, B)
    for b in 1:B
        sb[:, b] .= view(nzs[b][1], 1:ns)
    end
    mwb = Matrix{T}(undef, D, B)
    for b in 1:B
        mwb[:, b] .= fbs[b].f.mw
    end
    Lb = pk == P_DIAG ? Matrix{T}(undef, D, B) : Array{T}(undef, D, D, B)
    for b in 1:B
        pk == P_DIAG ? (Lb[:, b] .= prs[b][1]) : copyto!(view(Lb, :, :, b), prs[b][1])
    end
    ldl, strideL = pk == P_DIAG ? (1, D) : (D, D * D)
    mw_post = want_posterior ? Matrix{T}(undef, D, B) : Ptr{T}(C_NULL)
    Tp = want_posterior ? Array{T}(undef, D, D, B) : Ptr{T}(C_NULL)
    Ap = (want_posterior && pk != P_UPPER) ? Array{T}(undef, D, D, B) : Ptr{T}(C_NULL)
    lp = zeros(Cdouble, B)
    info = zeros(Int32, B)
    h = handle()
    rc = GC.@preserve Xb yb sb mwb Lb mw_post Tp Ap lp info begin
        if T === Float64
            ccall((:blr_posterior_batched_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Cint, Ptr{T}, Int64,
                   Ptr{T}, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{Cdouble}, Ptr{Int32}),
                  h, MEM_HOST, layout, B, D, N, Xb, rows, rows * cols, yb, N, nk, sb, ns, pk, mwb, D, Lb, ldl, strideL,
                  mw_post, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        else
            ccall((:blr_posterior_batched_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Cint, Ptr{T}, Int64,
                   Ptr{T}, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{Cdouble}, Ptr{Int32}),
                  h, MEM_HOST, layout, B, D, N, Xb, rows, rows * cols, yb, N, nk, sb, ns, pk, mwb, D, Lb, ldl, strideL,
                  mw_post, D, Tp, D, D * D, Ap, D, D * D, lp, info)
        end
    end
    check(h, rc)
    bad = findfirst(>(0), info)
    bad === nothing || throw(PosDefException(Int(info[bad])))
    want_posterior || return [(lp[b], nothing, nothing, nothing) for b in 1:B]
    return [(lp[b], mw_post[:, b], Tp[:, :, b], Ap isa Ptr ? Ap : Ap[:, :, b]) for b in 1:B]
end

function logpdf_map(fxs::AbstractVector{<:FiniteGP}, ys::AbstractVector{<:AbstractVector{<:Real}})
    r = fused_many(fxs, ys, false)
    r === nothing ? map(logpdf, fxs, ys) : Float64[q[1] for q in r]
end

function posterior_map(fxs::AbstractVector{<:FiniteGP}, ys::AbstractVector{<:AbstractVector{<:Real}})
    r = fused_many(fxs, ys, true)
    r === nothing && return map(posterior, fxs, ys)
    map(fxs, r) do fx, (_, mw_post, Tp, Ap)
        blr0 = fx.f isa BasisFunctionRegressor ? fx.f.blr : fx.f
        post = BayesianLinearRegressor(mw_post, build_Λ(typeof(blr0.Λw), Tp, Ap))
        fx.f isa BasisFunctionRegressor ? BasisFunctionRegressor(post, fx.f.ϕ) : post
    end
end

# The reference's own methods stay reachable after install_overrides! has replaced them: the fallbacks run in the world age
# recorded when this module was loaded (i.e. against the method tables as the reference defined them) -- no recursion.
const REF_WORLD = Ref{UInt}(0)
__init__() = (REF_WORLD[] = Base.get_world_counter(); nothing)
ref_call(f, args...) = Base.invoke_in_world(REF_WORLD[], f, args...)
REF_logpdf(fx, y) = ref_call(AbstractGPs.logpdf, fx, y)
REF_posterior(fx, y) = ref_call(AbstractGPs.posterior, fx, y)

# ---- marginal stream: reference :33, :40-43, :47 --------------------------------------------------------------
function mean_and_var(fx::FiniteGP; want_mean::Bool=true, want_var::Bool=true)
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && return ref_call(AbstractGPs.mean_and_var, fx)
    X, layout, ldx, D, N = xl
    T = eltype(X)
    mw = convert(Vector{T}, fb.f.mw)
    s, nk, _ = nz
    nk == DENSEN && ((s, nk) = (collect(diag(s)), DIAGONALN))                              # var adds diag(Σy) only (:43)
    Lw, pk, ldl = pr
    m = want_mean ? Vector{T}(undef, N) : Ptr{T}(C_NULL)
    v = want_var ? Vector{T}(undef, N) : Ptr{T}(C_NULL)
    info = Ref{Int32}(0)
    h = handle()
    rc = GC.@preserve X mw s Lw m v begin
        if T === Float64
            ccall((:blr_marginals_batched_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Cint, Ptr{T}, Int64, Cint, Ptr{T},
                   Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Ref{Int32}),
                  h, MEM_HOST, layout, 1, D, N, X, ldx, 0, nk, s, 0, pk, mw, 0, Lw, ldl, 0, m, N, v, N, info)
        else
            ccall((:blr_marginals_batched_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Cint, Ptr{T}, Int64, Cint, Ptr{T},
                   Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Ref{Int32}),
                  h, MEM_HOST, layout, 1, D, N, X, ldx, 0, nk, s, 0, pk, mw, 0, Lw, ldl, 0, m, N, v, N, info)
        end
    end
    check(h, rc)
    check(h, info[])
    return m, v
end
mean(fx::FiniteGP) = mean_and_var(fx; want_var=false)[1]
var(fx::FiniteGP) = mean_and_var(fx; want_mean=false)[2]
marginals(fx::FiniteGP) = ((m, v) = mean_and_var(fx); AbstractGPs.Normal.(m, sqrt.(v)))

# ---- full covariance: reference :35-38, :45 ----------------------------------------------------------------------
function mean_and_cov(fx::FiniteGP; want_mean::Bool=true)
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && return ref_call(AbstractGPs.mean_and_cov, fx)
    X, layout, ldx, D, N = xl
    T = eltype(X)
    mw = convert(Vector{T}, fb.f.mw)
    s, nk, lds = nz
    Lw, pk, ldl = pr
    m = want_mean ? Vector{T}(undef, N) : Ptr{T}(C_NULL)
    C = Matrix{T}(undef, N, N)
    info = Ref{Int32}(0)
    h = handle()
    rc = GC.@preserve X mw s Lw m C begin
        if T === Float64
            ccall((:blr_mean_and_cov_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Cint, Ptr{T}, Ptr{T}, Int64,
                   Ptr{T}, Ptr{T}, Int64, Ref{Int32}),
                  h, MEM_HOST, layout, D, N, X, ldx, nk, s, lds, pk, mw, Lw, ldl, m, C, N, info)
        else
            ccall((:blr_mean_and_cov_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Cint, Ptr{T}, Ptr{T}, Int64,
                   Ptr{T}, Ptr{T}, Int64, Ref{Int32}),
                  h, MEM_HOST, layout, D, N, X, ldx, nk, s, lds, pk, mw, Lw, ldl, m, C, N, info)
        end
    end
    check(h, rc)
    check(h, info[])
    return m, Symmetric(C)                                                                   # :37 Symmetric(α'α + Σy)
end
cov(fx::FiniteGP) = mean_and_cov(fx; want_mean=false)[2]

# ---- draws: reference :49-53 -- the RNG stream stays Julia's: Z1 = randn(rng, D, S) FIRST, then Z2 -------------
function rand(rng::AbstractRNG, fx::FiniteGP, samples::Int)
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && return ref_call(AbstractGPs.rand, rng, fx, samples)
    X, layout, ldx, D, N = xl
    T = eltype(X)
    mw = convert(Vector{T}, fb.f.mw)
    s, nk, lds = nz
    Lw, pk, ldl = pr
    Z1 = randn(rng, T, D, samples)       # reference :51
    Z2 = randn(rng, T, N, samples)       # reference :52
    Y = Matrix{T}(undef, N, samples)
    h = handle()
    rc = GC.@preserve X mw s Lw Z1 Z2 Y begin
        if nk == DENSEN
            if T === Float64
                ccall((:blr_rand_dense_noise_f64, LIB), Cint,
                      (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Ptr{T}, Int64,
                       Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64),
                      h, MEM_HOST, layout, D, N, samples, X, ldx, s, lds, pk, mw, Lw, ldl, Z1, D, Z2, N, Y, N)
            else
                ccall((:blr_rand_dense_noise_f32, LIB), Cint,
                      (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Ptr{T}, Int64,
                       Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64),
                      h, MEM_HOST, layout, D, N, samples, X, ldx, s, lds, pk, mw, Lw, ldl, Z1, D, Z2, N, Y, N)
            end
        elseif T === Float64
            ccall((:blr_rand_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Cint, Ptr{T}, Ptr{T}, Int64,
                   Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64),
                  h, MEM_HOST, layout, D, N, samples, X, ldx, nk, s, pk, mw, Lw, ldl, Z1, D, Z2, N, Y, N)
        else
            ccall((:blr_rand_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Cint, Ptr{T}, Ptr{T}, Int64,
                   Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64),
                  h, MEM_HOST, layout, D, N, samples, X, ldx, nk, s, pk, mw, Lw, ldl, Z1, D, Z2, N, Y, N)
        end
    end
    check(h, rc)
    return Y
end
rand(rng::AbstractRNG, fx::FiniteGP) = vec(rand(rng, fx, 1))

# ---- function-space samples: src/sampling_functions.jl:12-52 ------------------------------------------------------
function sample_weights(rng::AbstractRNG, blr::BayesianLinearRegressor, S::Int)
    pr = prior(blr.Λw)
    T = eltype(blr.mw)
    (pr === nothing || !(T <: Elt)) && return blr.mw .+ AbstractGPs._cholesky(blr.Λw).U \ randn(rng, length(blr.mw), S)
    D = length(blr.mw)
    Lw, pk, ldl = pr
    mw = convert(Vector{T}, blr.mw)
    Z = randn(rng, T, D, S)
    W = Matrix{T}(undef, D, S)
    h = handle()
    rc = GC.@preserve mw Lw Z W begin
        if T === Float64
            ccall((:blr_sample_weights_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Int64, Int64, Cint, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64),
                  h, MEM_HOST, D, S, pk, mw, Lw, ldl, Z, D, W, D)
        else
            ccall((:blr_sample_weights_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Int64, Int64, Cint, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64),
                  h, MEM_HOST, D, S, pk, mw, Lw, ldl, Z, D, W, D)
        end
    end
    check(h, rc)
    return W
end
const BLRLike = Union{BayesianLinearRegressor,BasisFunctionRegressor}
blr_and_mapping(b::BayesianLinearRegressor) = (b, identity)                               # sampling_functions.jl:51
blr_and_mapping(b::BasisFunctionRegressor) = (b.blr, b.ϕ)                                 # sampling_functions.jl:52
function rand(rng::AbstractRNG, b::BLRLike)                                               # sampling_functions.jl:27-31
    blr, ϕ = blr_and_mapping(b)
    BLRFunctionSample(vec(sample_weights(rng, blr, 1)), ϕ)
end
function rand(rng::AbstractRNG, b::BLRLike, dims::Dims)                                   # sampling_functions.jl:33-38
    blr, ϕ = blr_and_mapping(b)
    ws = sample_weights(rng, blr, prod(dims))
    reshape([BLRFunctionSample(collect(w), ϕ) for w in eachcol(ws)], dims)
end
rand(rng::AbstractRNG, b::BLRLike, d1::Integer, dims::Integer...) = rand(rng, b, Dims((d1, dims...)))
function rand!(rng::AbstractRNG, A::AbstractArray{<:BLRFunctionSample}, b::BLRLike)       # sampling_functions.jl:40-49
    blr, ϕ = blr_and_mapping(b)
    ws = sample_weights(rng, blr, length(A))
    for (i, w) in zip(eachindex(A), eachcol(ws))
        A[i] = BLRFunctionSample(collect(w), ϕ)
    end
    return A
end

# ---- Y = X'W for S given weight vectors (blr_apply_weights_*): a batch of function samples evaluated at the inputs,
#      sampling_functions.jl:16-18; `layout`, `D`, `N` describe X as the library reads it (see xlayout)
function apply_weights!(Y::Matrix{T}, X::Array{T}, layout::Integer, ldx::Integer, D::Integer, N::Integer, W::Matrix{T}) where {T<:Elt}
    S = size(W, 2)
    h = handle()
    rc = GC.@preserve X W Y begin
        if T === Float64
            ccall((:blr_apply_weights_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64),
                  h, MEM_HOST, layout, D, N, S, X, ldx, W, size(W, 1), Y, size(Y, 1))
        else
            ccall((:blr_apply_weights_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64),
                  h, MEM_HOST, layout, D, N, S, X, ldx, W, size(W, 1), Y, size(Y, 1))
        end
    end
    check(h, rc)
    return Y
end
"""
    evaluate(samples, x) -> Matrix (N x S)

Every function sample of `samples` (an array of `BLRFunctionSample`s drawn from ONE regressor, e.g. `rand(rng, f, S)`) at the
inputs `x`, in one pass over ϕ(x): column j is `samples[j](x)` (reference sampling_functions.jl:16-18).
"""
function evaluate(samples::AbstractArray{<:BLRFunctionSample}, x::AbstractVector)
    ϕx = first(samples).ϕ(x)
    xl = xlayout(ϕx)
    xl === nothing && return reduce(hcat, [smp(x) for smp in samples])
    X, layout, ldx, D, N = xl
    T = eltype(X)
    W = Matrix{T}(undef, D, length(samples))
    for (j, smp) in enumerate(samples)
        W[:, j] .= smp.w
    end
    return apply_weights!(Matrix{T}(undef, N, length(samples)), X, layout, ldx, D, N, W)
end

# ---- reverse-mode rule of rand(rng, fx, S) (README.md:56-60 differentiates it w.r.t. X, Σ, mw, Λw through Zygote) -------------
#   Y = X'W .+ sqrt.(s) .* Z2,  W = mw .+ U \ Z1   (reference :49-53), the draws held fixed:
#   W̄ = X Ȳ,  X̄ = W Ȳ',  m̄w = W̄ 1,  Ū = -triu(U'^-1 W̄ V'), V = U \ Z1,  s̄_n = Σ_s Ȳ[n,s] Z2[n,s] / (2 sqrt(s_n)).
# The two O(D N S) products are blr_apply_weights_* calls on re-interpreted layouts (the design matrix read with rows and
# columns swapped / the weights as the design matrix of an S-feature problem); the rest is D x D bookkeeping.
function ChainRulesCore.rrule(::typeof(rand), rng::AbstractRNG, fx::FiniteGP{<:BayesianLinearRegressor}, S::Int)
    xl, nz, pr = xlayout(fx.x), noise(fx.Σy), prior(fx.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing || nz[2] == DENSEN) && error("rrule(rand): input types outside the device path")
    X, layout, ldx, D, N = xl
    T = eltype(X)
    mw = convert(Vector{T}, fx.f.mw)
    s, nk, _ = nz
    Lw, pk, ldl = pr
    rng0 = copy(rng)
    Y = rand(rng, fx, S)                        # consumes Z1 = randn(rng, T, D, S), then Z2 = randn(rng, T, N, S)
    Z1 = randn(rng0, T, D, S); Z2 = randn(rng0, T, N, S)
    U = UpperTriangular(Matrix{T}(AbstractGPs._cholesky(fx.f.Λw).U))
    V = U \ Z1
    W = mw .+ V
    function rand_pullback(Ȳ0)
        Ȳ = convert(Matrix{T}, ChainRulesCore.unthunk(Ȳ0))
        flip = layout == COLVECS ? ROWVECS : COLVECS
        W̄ = apply_weights!(Matrix{T}(undef, D, S), X, flip, ldx, N, D, Ȳ)                              # X Ȳ
        X̄ = if layout == COLVECS                                                                       # W Ȳ' as D x N
            apply_weights!(Matrix{T}(undef, D, N), W, ROWVECS, D, S, D, Matrix{T}(Ȳ'))
        else                                                                                            # ... as N x D
            apply_weights!(Matrix{T}(undef, N, D), Ȳ, ROWVECS, N, S, N, Matrix{T}(W'))
        end
        Ū = -triu(U' \ (W̄ * V'))
        Λ = fx.f.Λw
        dΛ = if Λ isa Diagonal
            Diagonal(diag(Ū) ./ (2 .* sqrt.(Λ.diag)))
        elseif Λ isa AbstractPDMat
            Tangent{typeof(Λ)}(chol = Tangent{typeof(Λ.chol)}(factors = UpperTriangular(Ū)))
        else
            M = tril(U * Ū'); M[diagind(M)] ./= 2                                                       # Φ(L' L̄), L = U'
            Ab = U \ (U \ M')'
            (Ab .+ Ab') ./ 2
        end
        sv = nk == DIAGONALN ? s : fill(s[1], N)
        s̄ = vec(sum(Ȳ .* Z2; dims=2)) ./ (2 .* sqrt.(sv))
        dΣ = fx.Σy isa Diagonal{<:Any,<:AbstractGPs.FillArrays.Fill} ?
             Tangent{typeof(fx.Σy)}(diag = Tangent{typeof(fx.Σy.diag)}(value = sum(s̄))) : Diagonal(s̄)
        df = Tangent{typeof(fx.f)}(mw = vec(sum(W̄; dims=2)), Λw = dΛ)
        dx = Tangent{typeof(fx.x)}(X = X̄)
        return NoTangent(), NoTangent(), Tangent{typeof(fx)}(f = df, x = dx, Σy = dΣ), NoTangent()
    end
    return Y, rand_pullback
end
ChainRulesCore.rrule(::typeof(AbstractGPs.rand), rng::AbstractRNG, fx::FiniteGP{<:BayesianLinearRegressor}, S::Int) =
    ChainRulesCore.rrule(rand, rng, fx, S)

# ---- value + gradient of the log marginal likelihood (the rule behind the ccall; SURVEY.md 8f rank 1) ---------------
# Returns (lp, dX, dy, ds, dmw, mw_post, Ainv).
function logpdf_grad(fb::FiniteGP{<:BayesianLinearRegressor}, y::AbstractVector{<:Real})
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing || nz[2] == DENSEN) && error("logpdf_grad: input types outside the device path")
    X, layout, ldx, D, N = xl
    T = eltype(X)
    s, nk, _ = nz
    Lw, pk, ldl = pr
    yv = convert(Vector{T}, y); mw = convert(Vector{T}, fb.f.mw)
    dX = similar(X); dy = Vector{T}(undef, N); ds = Vector{T}(undef, N); dmw = Vector{T}(undef, D)
    mw′ = Vector{T}(undef, D); Ai = Matrix{T}(undef, D, D)
    lp = Ref{Cdouble}(0.0); info = Ref{Int32}(0)
    h = handle()
    rc = GC.@preserve X yv s Lw mw dX dy ds dmw mw′ Ai begin
        if T === Float64
            ccall((:blr_logpdf_grad_batched_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Cint,
                   Ptr{T}, Int64, Ptr{T}, Int64, Int64, Ref{Cdouble}, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64,
                   Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Int64, Ref{Int32}),
                  h, MEM_HOST, layout, 1, D, N, X, ldx, 0, yv, 0, nk, s, 0, pk, mw, 0, Lw, ldl, 0, lp, dX, ldx, 0, dy, 0, ds, 0,
                  dmw, 0, mw′, 0, Ai, D, 0, info)
        else
            ccall((:blr_logpdf_grad_batched_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Cint,
                   Ptr{T}, Int64, Ptr{T}, Int64, Int64, Ref{Cdouble}, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64,
                   Ptr{T}, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Int64, Ref{Int32}),
                  h, MEM_HOST, layout, 1, D, N, X, ldx, 0, yv, 0, nk, s, 0, pk, mw, 0, Lw, ldl, 0, lp, dX, ldx, 0, dy, 0, ds, 0,
                  dmw, 0, mw′, 0, Ai, D, 0, info)
        end
    end
    check(h, rc)
    info[] > 0 && throw(PosDefException(info[]))
    return lp[], dX, dy, ds, dmw, mw′, Ai
end

# The reverse-mode rule (README.md:56-71: Zygote differentiates logpdf through the reference's Julia code; a ccall is opaque).
#   dL/dy = -S r, dL/dmw = X S r, dL/dX = (mw' r' - A^-1 X) S, dL/ds_n = -(s_n - r_n^2 - x_n'A^-1 x_n) / (2 s_n^2),
#   dL/dΛw = -(m m' + A^-1 - Λw^-1) / 2 with m = mw' - mw;  r = y - X'mw' (one call: blr_logpdf_grad_batched_*).
# Tangents mirror the primal structs: FiniteGP(f = BLR(mw, Λw), x = ColVecs/RowVecs(X), Σy).
function ChainRulesCore.rrule(::typeof(logpdf), fx::FiniteGP{<:BayesianLinearRegressor}, y::AbstractVector{<:Real})
    lp, dX, dy, ds, dmw, mw′, Ai = logpdf_grad(fx, y)
    function logpdf_pullback(Δ)
        δ = ChainRulesCore.unthunk(Δ)
        m = mw′ .- fx.f.mw
        Λ = fx.f.Λw
        dΛ = if Λ isa Diagonal
            Diagonal(δ .* (-(m .^ 2 .+ diag(Ai) .- inv.(Λ.diag)) ./ 2))
        elseif Λ isa AbstractPDMat
            # the reference reads a PDMat prior through its factor (_cholesky(Λw) = Λw.chol, :78), so the cotangent belongs to
            # chol.factors: Λw = U'U  =>  dU = U (G + G') with G = dL/dΛw = -(m m' + A^-1 - Λw^-1) / 2 (symmetric), upper part
            G = -(m * m' .+ Ai .- inv(Λ.chol)) ./ 2
            dU = UpperTriangular(δ .* (Matrix(Λ.chol.U) * (G .+ G')))
            Tangent{typeof(Λ)}(chol = Tangent{typeof(Λ.chol)}(factors = dU))
        else
            δ .* (-(m * m' .+ Ai .- inv(Matrix(Λ))) ./ 2)
        end
        dΣ = fx.Σy isa Diagonal{<:Any,<:AbstractGPs.FillArrays.Fill} ?
             Tangent{typeof(fx.Σy)}(diag = Tangent{typeof(fx.Σy.diag)}(value = δ * sum(ds))) : Diagonal(δ .* ds)
        df = Tangent{typeof(fx.f)}(mw = δ .* dmw, Λw = dΛ)
        dx = Tangent{typeof(fx.x)}(X = δ .* dX)                       # same container layout as the primal (D x N or N x D)
        return NoTangent(), Tangent{typeof(fx)}(f = df, x = dx, Σy = dΣ), δ .* dy
    end
    return lp, logpdf_pullback
end
ChainRulesCore.rrule(::typeof(AbstractGPs.logpdf), fx::FiniteGP{<:BayesianLinearRegressor}, y::AbstractVector{<:Real}) =
    ChainRulesCore.rrule(logpdf, fx, y)

# ---- logpdf(fx, Y::AbstractMatrix): shared-X multi-output evidence (AbstractGPs' column-wise fallback) ----------------
function logpdf(fx::FiniteBLR, Y::AbstractMatrix{<:Real})
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing || nz[2] == DENSEN) && return [logpdf(fb, y) for y in eachcol(Y)]
    X, layout, ldx, D, N = xl
    T = eltype(X)
    s, nk, _ = nz
    Lw, pk, ldl = pr
    size(Y, 1) == N || throw(DimensionMismatch("length(y) != size(fx.x.X, 2)"))
    S = size(Y, 2)
    Ym = convert(Matrix{T}, Y); mw = convert(Vector{T}, fb.f.mw)
    lp = Vector{Float64}(undef, S); info = Ref{Int32}(0)
    h = handle()
    rc = GC.@preserve X Ym s Lw mw lp begin
        if T === Float64
            ccall((:blr_logpdf_multi_f64, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Cint, Ptr{T}, Ptr{T},
                   Int64, Ptr{Cdouble}, Ptr{T}, Int64, Ref{Int32}),
                  h, MEM_HOST, layout, D, N, S, X, ldx, Ym, N, nk, s, pk, mw, Lw, ldl, lp, C_NULL, D, info)
        else
            ccall((:blr_logpdf_multi_f32, LIB), Cint,
                  (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Cint, Ptr{T}, Ptr{T},
                   Int64, Ptr{Cdouble}, Ptr{T}, Int64, Ref{Int32}),
                  h, MEM_HOST, layout, D, N, S, X, ldx, Ym, N, nk, s, pk, mw, Lw, ldl, lp, C_NULL, D, info)
        end
    end
    check(h, rc)
    info[] > 0 && throw(PosDefException(info[]))
    return lp
end

# ---- device-resident batches: the form every throughput number is measured on ------------------------------------------
# A DeviceArray owns hipMalloc'd memory through the library's own helpers (no AMDGPU.jl needed); AMDGPU.jl users pass
# `Ptr{T}(pointer(roc_array))` instead.  Layout = Julia's: X is D x N x B column-major (regressor b at offset (b-1) D N).
mutable struct DeviceArray{T}
    ptr::Ptr{T}
    len::Int
    function DeviceArray{T}(len::Integer) where {T}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        h = handle()
        check(h, ccall((:blr_device_alloc, LIB), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), h, len * sizeof(T), p))
        a = new{T}(Ptr{T}(p[]), len)
        finalizer(a -> ccall((:blr_device_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), handle(), a.ptr), a)
    end
end
# Run-time switches of this task's handle (A/B measurements and tests; the defaults are the measured best).  `key` with or
# without the BLR_MI355X_ prefix (include/blr_mi355x.h lists them); `value === nothing` restores the default.
function set_option!(key::AbstractString, value::Union{Nothing,AbstractString,Integer}=nothing)
    h = handle()
    v = value === nothing ? C_NULL : string(value)
    rc = ccall((:blr_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Cstring), h, key, v)
    rc == 0 || throw(ArgumentError("blr_set_option: unknown key or malformed value: $key = $value"))
    return nothing
end
# Which kernel family this task's most recent posterior / logpdf call ran on ("fused_i8_kernel", "fused_small_kernel<double, 8, 4>", ...).
last_route() = unsafe_string(ccall((:blr_last_route, LIB), Cstring, (Ptr{Cvoid},), handle()))
# Counters of this task's handle: "i8_regressors" (sent down the int8-sliced Gram route), "i8_handed_back" (of those, redone by the fp64
# kernel inside the same call: each cost two passes over its data -- heavy-tailed design matrices), "planes_redone" (fp32 updates at
# D > 128 whose sampled row scales did not hold: exact row maxima + the operand planes a second time), "workspace_bytes".
function get_stat(key::AbstractString)
    h = handle()
    v = Ref{Int64}(0)
    check(h, ccall((:blr_get_stat, LIB), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), h, key, v))
    return v[]
end
reset_stats!() = (h = handle(); check(h, ccall((:blr_reset_stats, LIB), Cint, (Ptr{Cvoid},), h)); nothing)
# Give the handle's workspace, feature and side buffers back to the allocator (they are re-created by the next call that needs them).
release_workspace!() = (h = handle(); check(h, ccall((:blr_release_workspace, LIB), Cint, (Ptr{Cvoid},), h)); nothing)

function upload(a::Array{T}) where {T}
    d = DeviceArray{T}(length(a))
    h = handle()
    GC.@preserve a check(h, ccall((:blr_memcpy_h2d, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), h, d.ptr, a, sizeof(a)))
    d
end
function download!(a::Array{T}, d::DeviceArray{T}) where {T}
    h = handle()
    GC.@preserve a check(h, ccall((:blr_memcpy_d2h, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), h, a, d.ptr, sizeof(a)))
    a
end

"""
    posterior_batched!(mw_post, T_post, logpdf, info, X, y, s, mw, Λdiag; D, N, B, isotropic)

B independent regressors in ONE launch, everything device resident (`DeviceArray`s): X is D×N×B, y N×B, s one variance
(`isotropic`) or N×B, mw D×B, Λdiag the D diagonal entries of a prior precision shared by the batch; outputs mw_post D×B,
T_post D×D×B (upper factors), logpdf B (Float64), info B (Int32).  Reference semantics per regressor: `:55-69`.
"""
function posterior_batched!(mw_post::DeviceArray{T}, T_post::DeviceArray{T}, lp::DeviceArray{Float64}, info::DeviceArray{Int32},
                            X::DeviceArray{T}, y::DeviceArray{T}, s::DeviceArray{T}, mw::DeviceArray{T}, Λdiag::DeviceArray{T};
                            D::Int, N::Int, B::Int, isotropic::Bool) where {T<:Elt}
    h = handle()
    nk = isotropic ? ISOTROPIC : DIAGONALN
    rc = if T === Float64
        ccall((:blr_posterior_batched_f64, LIB), Cint,
              (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Cint, Ptr{T},
               Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{Cdouble}, Ptr{Int32}),
              h, MEM_DEVICE, COLVECS, B, D, N, X.ptr, D, D * N, y.ptr, N, nk, s.ptr, isotropic ? 0 : N, P_DIAG, mw.ptr, D,
              Λdiag.ptr, 1, 0, mw_post.ptr, D, T_post.ptr, D, D * D, Ptr{T}(C_NULL), D, D * D, lp.ptr, info.ptr)
    else
        ccall((:blr_posterior_batched_f32, LIB), Cint,
              (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Cint, Ptr{T},
               Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{Cdouble}, Ptr{Int32}),
              h, MEM_DEVICE, COLVECS, B, D, N, X.ptr, D, D * N, y.ptr, N, nk, s.ptr, isotropic ? 0 : N, P_DIAG, mw.ptr, D,
              Λdiag.ptr, 1, 0, mw_post.ptr, D, T_post.ptr, D, D * D, Ptr{T}(C_NULL), D, D * D, lp.ptr, info.ptr)
    end
    check(h, rc)
    return nothing
end

"""
    update_factor!(mw, T, logpdf, info, X, y, s; D, k, B, isotropic)

Condition B device-resident posterior states IN PLACE on k further observations each -- the "repeated conditioning" of
reference `test/bayesian_linear_regression.jl:49-70` without going back through a D×D precision (`:93`, `:72-89`):
mw is D×B, T D×D×B (upper factors, exactly what `posterior_batched!` wrote), X D×k×B, y k×B, s one variance (`isotropic`) or
k×B.  `logpdf[b]` = log p(y_b | state before the call) -- the evidence increment of the chain rule.
"""
function update_factor!(mw::DeviceArray{T}, Tf::DeviceArray{T}, lp::DeviceArray{Float64}, info::DeviceArray{Int32},
                        X::DeviceArray{T}, y::DeviceArray{T}, s::DeviceArray{T}; D::Int, k::Int, B::Int, isotropic::Bool) where {T<:Elt}
    h = handle()
    nk = isotropic ? ISOTROPIC : DIAGONALN
    rc = if T === Float64
        ccall((:blr_update_factor_f64, LIB), Cint,
              (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Ptr{T}, Int64,
               Ptr{T}, Int64, Int64, Ptr{Cdouble}, Ptr{Int32}),
              h, MEM_DEVICE, COLVECS, B, D, k, X.ptr, D, D * k, y.ptr, k, nk, s.ptr, isotropic ? 0 : k, mw.ptr, D, Tf.ptr, D, D * D,
              lp.ptr, info.ptr)
    else
        ccall((:blr_update_factor_f32, LIB), Cint,
              (Ptr{Cvoid}, Cint, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Int64, Ptr{T}, Int64, Cint, Ptr{T}, Int64, Ptr{T}, Int64,
               Ptr{T}, Int64, Int64, Ptr{Cdouble}, Ptr{Int32}),
              h, MEM_DEVICE, COLVECS, B, D, k, X.ptr, D, D * k, y.ptr, k, nk, s.ptr, isotropic ? 0 : k, mw.ptr, D, Tf.ptr, D, D * D,
              lp.ptr, info.ptr)
    end
    check(h, rc)
    return nothing
end

# ---- one Julia process per GPU: the path's single exchange through RCCL, called directly (SURVEY.md 8e) ------------------
"rank 0: 128 opaque bytes to ship to the other ranks (Distributed.jl, a file, MPI ...)"
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = GC.@preserve id ccall((:blr_comm_unique_id, LIB), Cint, (Ptr{Cvoid},), id)
    rc == 0 || error("blr_comm_unique_id failed with code $rc (librccl missing?)")
    id
end
function comm_init!(nranks::Integer, rank::Integer, id::Vector{UInt8})
    h = handle()
    GC.@preserve id check(h, ccall((:blr_comm_init, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), h, nranks, rank, id))
end
comm_destroy!() = (h = handle(); check(h, ccall((:blr_comm_destroy, LIB), Cint, (Ptr{Cvoid},), h)))

"""
    posterior_nsharded!(mw_post, T_post, logpdf, info, stats, scal, X, y, s, mw, Λdiag; D, N_local, N_total)

ONE regressor whose observations are split over the ranks of `comm_init!` (SURVEY §8e): this rank's `N_local` columns of X
(D×N_local, device), diagonal noise `s`, diagonal prior `Λdiag`; `stats` ((128⌈D/128⌉+128) × 128⌈D/128⌉) and `scal` (2 Float64)
are device scratch.  Every rank ends with the same posterior (`mw_post`, upper factor `T_post`) and evidence of all `N_total`
observations: local statistics, one in-place all-reduce, redundant D×D finish.
"""
function posterior_nsharded!(mw_post::DeviceArray{T}, T_post::DeviceArray{T}, lp::DeviceArray{Float64}, info::DeviceArray{Int32},
                             stats::DeviceArray{T}, scal::DeviceArray{Float64}, X::DeviceArray{T}, y::DeviceArray{T},
                             s::DeviceArray{T}, mw::DeviceArray{T}, Λdiag::DeviceArray{T}; D::Int, N_local::Int, N_total::Int) where {T<:Elt}
    h = handle()
    lds = 128 * cld(D, 128) + 128
    rc = if T === Float64
        ccall((:blr_posterior_nsharded_f64, LIB), Cint,
              (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Cint, Ptr{T}, Cint, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64,
               Ptr{Cdouble}, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ptr{Cdouble}, Ptr{Int32}),
              h, COLVECS, D, N_local, N_total, X.ptr, D, y.ptr, DIAGONALN, s.ptr, P_DIAG, mw.ptr, Λdiag.ptr, 1, stats.ptr, lds,
              scal.ptr, mw_post.ptr, T_post.ptr, D, Ptr{T}(C_NULL), D, lp.ptr, info.ptr)
    else
        ccall((:blr_posterior_nsharded_f32, LIB), Cint,
              (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{T}, Int64, Ptr{T}, Cint, Ptr{T}, Cint, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64,
               Ptr{Cdouble}, Ptr{T}, Ptr{T}, Int64, Ptr{T}, Int64, Ptr{Cdouble}, Ptr{Int32}),
              h, COLVECS, D, N_local, N_total, X.ptr, D, y.ptr, DIAGONALN, s.ptr, P_DIAG, mw.ptr, Λdiag.ptr, 1, stats.ptr, lds,
              scal.ptr, mw_post.ptr, T_post.ptr, D, Ptr{T}(C_NULL), D, lp.ptr, info.ptr)
    end
    check(h, rc)
    return nothing
end

comm_size() = ccall((:blr_comm_size, LIB), Cint, (Ptr{Cvoid},), handle())
comm_rank() = ccall((:blr_comm_rank, LIB), Cint, (Ptr{Cvoid},), handle())
"all-gather of every rank's `count` log evidences + the same fixed-order sum on every rank; returns the total"
function logpdf_allgather_sum!(lp_all::DeviceArray{Float64}, lp_local::DeviceArray{Float64}, count::Integer)
    tot = DeviceArray{Float64}(1)
    h = handle()
    check(h, ccall((:blr_logpdf_allgather_sum, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                   h, count, lp_local.ptr, lp_all.ptr, tot.ptr))
    download!(Vector{Float64}(undef, 1), tot)[1]
end
"in-place sum over ranks of a device buffer (the `stats` exchange of the N-sharded single regressor)"
function allreduce_sum!(buf::DeviceArray{T}) where {T<:Elt}
    h = handle()
    check(h, ccall((:blr_allreduce_sum, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64), h, T === Float64 ? 1 : 0, buf.ptr, buf.len))
    buf
end

# ---- drop-in: replace the reference's methods for the session ------------------------------------------------------------
"""
    install_overrides!()

Redefine `AbstractGPs.logpdf / posterior / mean / var / cov / mean_and_var / mean_and_cov / rand` on
`FiniteGP{<:BayesianLinearRegressor}` and `FiniteGP{<:BasisFunctionRegressor}` (reference
src/bayesian_linear_regression.jl:33-69, src/basis_function_regression.jl:46-65) and the `Random` hooks of
src/sampling_functions.jl:23-49 so that unmodified user code runs on the MI355X path.  Method overwriting is deliberate and
explicit (Julia prints one warning per method); inputs the device path does not take still reach the reference's code.
"""
function install_overrides!()
    # the EXACT signatures of the reference (one per regressor type), so these definitions replace its methods
    for R in (:BayesianLinearRegressor, :BasisFunctionRegressor)
        @eval begin
            AbstractGPs.logpdf(fx::FiniteGP{<:$R}, y::AbstractVector{<:Real}) = BLRMI355X.logpdf(fx, y)
            AbstractGPs.logpdf(fx::FiniteGP{<:$R}, Y::AbstractMatrix{<:Real}) = BLRMI355X.logpdf(fx, Y)
            AbstractGPs.posterior(fx::FiniteGP{<:$R}, y::AbstractVector{<:Real}) = BLRMI355X.posterior(fx, y)
            AbstractGPs.mean(fx::FiniteGP{<:$R}) = BLRMI355X.mean(fx)
            AbstractGPs.var(fx::FiniteGP{<:$R}) = BLRMI355X.var(fx)
            AbstractGPs.cov(fx::FiniteGP{<:$R}) = BLRMI355X.cov(fx)
            AbstractGPs.mean_and_var(fx::FiniteGP{<:$R}) = BLRMI355X.mean_and_var(fx)
            AbstractGPs.mean_and_cov(fx::FiniteGP{<:$R}) = BLRMI355X.mean_and_cov(fx)
            AbstractGPs.rand(rng::AbstractRNG, fx::FiniteGP{<:$R}, S::Int) = BLRMI355X.rand(rng, fx, S)
        end
    end
    @eval begin
        # the reference's Sampler hook returns the regressor ITSELF (sampling_functions.jl:23-25), so `rand(rng, f)` dispatches
        # on the regressor type (:27), not on a SamplerTrivial
        Random.rand(rng::AbstractRNG, b::BLRLike) = BLRMI355X.rand(rng, b)
        Random.rand(rng::AbstractRNG, b::BLRLike, dims::Dims) = BLRMI355X.rand(rng, b, dims)
        Random.rand!(rng::AbstractRNG, A::AbstractArray{<:BLRFunctionSample}, b::BLRLike) = BLRMI355X.rand!(rng, A, b)
    end
    return nothing
end

end # module
