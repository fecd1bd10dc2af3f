# BLRMI355X.jl -- the Julia side of the drop-in boundary (NOT runnable in the build image: no `julia`).
#
# Kept deliberately thin so it is correct by inspection: every behaviour it relies on is exercised through
# the same C ABI by the Python/ctypes harness (tests/test_gpu_parity.py).  It offloads
#     logpdf(fx, y), posterior(fx, y), mean(fx), var(fx), mean_and_var(fx), rand(rng, fx, S), rand(rng, f[, dims])
# for FiniteGP{<:BayesianLinearRegressor} / FiniteGP{<:BasisFunctionRegressor} with Float64/Float32 data,
# ColVecs / RowVecs inputs and Diagonal (incl. Fill) observation noise, and falls back to the reference's own
# CPU methods for everything else (dense Sigma_y, cov, exotic element types).
#
# Usage:   using AbstractGPs, BayesianLinearRegressors; include("BLRMI355X.jl"); using .BLRMI355X
#          f = BayesianLinearRegressor(mw, Lw); fx = f(ColVecs(X), Diagonal(s))
#          BLRMI355X.logpdf(fx, y); BLRMI355X.posterior(fx, y); BLRMI355X.mean_and_var(fx)
# A maintainer who wants the offload to be the default replaces the bodies of the reference methods at
# src/bayesian_linear_regression.jl:33,40,47,49,55,60 with calls to the functions below (INTEGRATION.md).
module BLRMI355X

using AbstractGPs, LinearAlgebra, PDMats, Random
using AbstractGPs: FiniteGP
using BayesianLinearRegressors: BayesianLinearRegressor, BasisFunctionRegressor, BLRFunctionSample
import BayesianLinearRegressors as REF

const LIB = get(ENV, "BLR_MI355X_LIB", "libblr_mi355x")

# enums of include/blr_mi355x.h
const COLVECS, ROWVECS = Cint(0), Cint(1)
const ISOTROPIC, DIAGONALN = Cint(0), Cint(1)
const P_DENSE, P_UPPER, P_DIAG = Cint(0), Cint(1), Cint(2)
const MEM_HOST = Cint(0)

const Elt = Union{Float32,Float64}
sfx(::Type{Float64}) = "f64"
sfx(::Type{Float32}) = "f32"

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
    rc > 0 && throw(PosDefException(rc))                       # what cholesky at reference :78/:86 throws
    msg = unsafe_string(ccall((:blr_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    rc > -1000 ? throw(DimensionMismatch("libblr_mi355x: $msg")) : error("libblr_mi355x (HIP): $msg")
end

# ---- x_as_colvecs (reference :20-31) as (array, layout, ld, D, N): zero copies -----------------------------
xlayout(x::ColVecs{T,<:StridedMatrix{T}}) where {T<:Elt} = (x.X, COLVECS, stride(x.X, 2), size(x.X, 1), size(x.X, 2))
xlayout(x::RowVecs{T,<:StridedMatrix{T}}) where {T<:Elt} = (x.X, ROWVECS, stride(x.X, 2), size(x.X, 2), size(x.X, 1))
xlayout(x) = nothing  # anything else: not offloadable (the reference raises its own error for unknown containers)

noise(Σ::Diagonal{T,<:AbstractGPs.FillArrays.Fill}) where {T<:Elt} = (T[Σ.diag.value], ISOTROPIC)
noise(Σ::Diagonal{T,<:StridedVector{T}}) where {T<:Elt} = (Σ.diag, DIAGONALN)
noise(Σ) = nothing    # dense Sigma_y stays on the CPU path (SURVEY.md 2 #19)

prior(Λ::Diagonal{T}) where {T<:Elt} = (collect(Λ.diag), P_DIAG, 1)
prior(Λ::PDMat{T}) where {T<:Elt} = (Matrix(Λ.chol.U), P_UPPER, size(Λ, 1))          # factor carried forward (:93)
prior(Λ::Symmetric{T,<:StridedMatrix{T}}) where {T<:Elt} = (Matrix(Λ), P_DENSE, size(Λ, 1))
prior(Λ::StridedMatrix{T}) where {T<:Elt} = (Λ, P_DENSE, stride(Λ, 2))
prior(Λ) = nothing

build_Λ(::Type{<:AbstractPDMat}, T, A) = PDMat(Cholesky(UpperTriangular(T)))           # reference :93
build_Λ(_, T, A) = Symmetric(A)                                                          # reference :92 (A == T'T)

to_blr(fx::FiniteGP{<:BayesianLinearRegressor}) = fx
to_blr(fx::FiniteGP{<:BasisFunctionRegressor}) = fx.f.blr(fx.f.ϕ(fx.x), fx.Σy)         # basis_function_regression.jl:41

# ---- fused inference: reference :55-58, :60-69, :72-89 ------------------------------------------------------
function fused(fx::FiniteGP, y::AbstractVector{<:Real}, want_posterior::Bool)
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && return nothing
    X, layout, ldx, D, N = xl
    T = eltype(X)
    length(y) == N || throw(error("length(y) != size(fx.x.X, 2)"))                        # reference :74
    yv, mw = convert(Vector{T}, y), convert(Vector{T}, fb.f.mw)
    s, nk = nz
    Lw, pk, ldl = pr
    mw_post = want_posterior ? Vector{T}(undef, D) : Ptr{T}(C_NULL)
    Tp = want_posterior ? Matrix{T}(undef, D, D) : Ptr{T}(C_NULL)
    Ap = (want_posterior && pk != P_UPPER) ? Matrix{T}(undef, D, D) : Ptr{T}(C_NULL)
    lp = Ref{Cdouble}(0.0)
    h = handle()
    sym = T === Float64 ? :blr_posterior_f64 : :blr_posterior_f32
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
    r === nothing ? AbstractGPs.logpdf(fx, y) : r[1]
end

function posterior(fx::FiniteGP, y::AbstractVector{<:Real})
    r = fused(fx, y, true)
    r === nothing && return AbstractGPs.posterior(fx, y)
    _, mw_post, Tp, Ap = r
    blr0 = fx.f isa BasisFunctionRegressor ? fx.f.blr : fx.f
    post = BayesianLinearRegressor(mw_post, build_Λ(typeof(blr0.Λw), Tp, Ap))
    fx.f isa BasisFunctionRegressor ? BasisFunctionRegressor(post, fx.f.ϕ) : post           # :62-65
end

# ---- marginal stream: reference :33, :40-43, :47 --------------------------------------------------------------
function mean_and_var(fx::FiniteGP; want_mean::Bool=true, want_var::Bool=true)
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && return AbstractGPs.mean_and_var(fx)
    X, layout, ldx, D, N = xl
    T = eltype(X)
    mw = convert(Vector{T}, fb.f.mw)
    s, nk = nz
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

# ---- draws: reference :49-53 -- the RNG stream stays Julia's: Z1 = randn(rng, D, S) FIRST, then Z2 -------------
function rand(rng::AbstractRNG, fx::FiniteGP, samples::Int)
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && return AbstractGPs.rand(rng, fx, samples)
    X, layout, ldx, D, N = xl
    T = eltype(X)
    mw = convert(Vector{T}, fb.f.mw)
    s, nk = nz
    Lw, pk, ldl = pr
    Z1 = randn(rng, T, D, samples)       # reference :51
    Z2 = randn(rng, T, N, samples)       # reference :52
    Y = Matrix{T}(undef, N, samples)
    h = handle()
    rc = GC.@preserve X mw s Lw Z1 Z2 Y begin
        if T === Float64
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

# ---- function-space samples: src/sampling_functions.jl:27-38 ------------------------------------------------------
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
blr_and_mapping(b::BayesianLinearRegressor) = (b, identity)
blr_and_mapping(b::BasisFunctionRegressor) = (b.blr, b.ϕ)
function rand(rng::AbstractRNG, b::Union{BayesianLinearRegressor,BasisFunctionRegressor})
    blr, ϕ = blr_and_mapping(b)
    BLRFunctionSample(vec(sample_weights(rng, blr, 1)), ϕ)
end
function rand(rng::AbstractRNG, b::Union{BayesianLinearRegressor,BasisFunctionRegressor}, dims::Dims)
    blr, ϕ = blr_and_mapping(b)
    ws = sample_weights(rng, blr, prod(dims))
    reshape([BLRFunctionSample(collect(w), ϕ) for w in eachcol(ws)], dims)
end

# ---- random-Fourier basis on the device (BASELINE config 5) -----------------------------------------
# A BasisFunctionRegressor whose ϕ is  x -> sqrt(2/D) cos.(Ω'x .+ β)  (reference basis_function_regression.jl:41,62-65:
# bfr(x) = blr(ϕ(x))) can hand the raw inputs to the device: the feature matrix is generated there and consumed by the
# same fused posterior/logpdf path, never crossing PCIe.
struct RandomFourierFeatures{T<:Elt}
    Ω::Matrix{T}      # Din x D
    β::Vector{T}      # D
end
(r::RandomFourierFeatures{T})(x::ColVecs) where {T} = ColVecs(convert(T, sqrt(2 / length(r.β))) .* cos.(r.Ω' * x.X .+ r.β))

function fused_rff(blr::BayesianLinearRegressor, ϕ::RandomFourierFeatures{T}, x::ColVecs, Σy, y::AbstractVector) where {T}
    pr = prior(blr.Λw); nz = noise(Σy)
    (pr === nothing || nz === nothing) && return nothing
    Lw, pk, ldl = pr; s, nk = nz
    Din, D = size(ϕ.Ω); N = length(y)
    Xin = convert(Matrix{T}, x.X); yv = convert(Vector{T}, y); mw = convert(Vector{T}, blr.mw)
    mw′ = Vector{T}(undef, D); Tm = Matrix{T}(undef, D, D); A = Matrix{T}(undef, D, D)
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

# ---- value + gradient of the log marginal likelihood (the rule behind the ccall; SURVEY.md 8f rank 1) ---------------
# Returns (lp, dX, dy, ds, dmw, mw_post, Ainv); see INTEGRATION.md for the ChainRules rrule built on it.
function logpdf_grad(fb, y::AbstractVector{<:Real})
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && error("logpdf_grad: input types outside the device path")
    X, layout, ldx, D, N = xl
    T = eltype(X)
    s, nk = nz
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

# ---- logpdf(fx, Y::AbstractMatrix): shared-X multi-output evidence (AbstractGPs' column-wise fallback) ----------------
function AbstractGPs.logpdf(fx::FiniteGP{<:Union{BayesianLinearRegressor,BasisFunctionRegressor}}, Y::AbstractMatrix{<:Real})
    fb = to_blr(fx)
    xl, nz, pr = xlayout(fb.x), noise(fb.Σy), prior(fb.f.Λw)
    (xl === nothing || nz === nothing || pr === nothing) && return [AbstractGPs.logpdf(fb, y) for y in eachcol(Y)]
    X, layout, ldx, D, N = xl
    T = eltype(X)
    s, nk = nz
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

end # module
