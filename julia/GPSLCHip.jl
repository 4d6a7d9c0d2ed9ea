# GPSLCHip.jl — the reference-side binding of libgpslc_hip.so (include/gpslc_hip.h).
#
# How a CausalGPSLC.jl maintainer uses it: in src/CausalGPSLC.jl, add
#     include("GPSLCHip.jl")
# AFTER the existing includes (types.jl, utils.jl, kernel.jl, likelihood.jl, estimation.jl, driver.jl,
# prediction.jl, model_likelihood.jl must be loaded first: the methods below have the SAME signatures as the
# reference's and therefore replace their bodies).  Nothing in src/inference.jl or src/prediction.jl changes: they
# only use rbfKernelLog / processCov (src/inference.jl:225-227, 286-287, 343-344), sampleITE, getN,
# getNumPosteriorSamples and Gen addresses.
#
# Part 1 (module GPSLCHip) binds EVERY function the header declares, one thin wrapper per symbol — the `ccall`
# type tuples are checked mechanically against include/gpslc_hip.h and causalgpslc.jl_amd/_lib.py by
# tests/test_julia_binding.py (arity, type order, struct layouts), because Julia is not available in the build
# pipeline and this file cannot be executed there.
# Part 2 re-defines the reference's hot-path functions on top of those wrappers.
#
# Conventions (header comment of gpslc_hip.h): column-major Float64 everywhere (Julia's own layout), Bool
# treatments pre-converted to 0.0 / 1.0, status 0 = ok, > 0 = PosDefException(info), < 0 = error.

module GPSLCHip

using LinearAlgebra: PosDefException

const lib = get(ENV, "GPSLC_HIP_LIB", joinpath(@__DIR__, "..", "deps", "libgpslc_hip.so"))

# ---- gpslc_node / gpslc_pack_header: same field order and types as the C structs ------------------------------
struct GPSLCNode
    nF::Int32
    reserved::Int32
    F::Ptr{Float64}
    ls::Ptr{Float64}
    scale::Float64
    noise::Float64
    target::Ptr{Float64}
end

struct PackHeader
    n::Int64
    nX::Int64
    nU::Int64
    S::Int64
    binary_t::Int64
    reserved::Int64
    hyper::NTuple{7,Float64}
end
PackHeader() = PackHeader(0, 0, 0, 0, 0, 0, ntuple(_ -> 0.0, 7))

const FLAG_DEFAULT = UInt32(0)
const FLAG_PROFILE = UInt32(1)
const FLAG_FP32_KERNEL = UInt32(2)

# ---- context --------------------------------------------------------------------------------------------------
mutable struct Ctx
    h::Ptr{Cvoid}
    n::Int
    nX::Int
    nU::Int
    data::Any                      # (X, T, Y) as handed to set_data! (Float64 host copies), or nothing
    cov::Any                       # the dense covariance handed to gpslc_mvn_logpdf last (identity-compared), or nothing
    cov_chol::Any                  # its host factor, for prior draws (HipMvNormal's `random`), or nothing
    function Ctx(n::Integer, nX::Integer, nU::Integer; device::Integer=0, flags::Integer=FLAG_DEFAULT)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:gpslc_create, lib), Cint, (Ref{Ptr{Cvoid}}, Cint, Int64, Int32, Int32, UInt32),
                   r, device, n, nX, nU, flags)
        st == 0 || error("gpslc_create: status $st")
        c = new(r[], n, nX, nU, nothing, nothing, nothing)
        finalizer(destroy!, c)
        c
    end
end

function destroy!(c::Ctx)
    if c.h != C_NULL
        ccall((:gpslc_destroy, lib), Cint, (Ptr{Cvoid},), c.h)
        c.h = C_NULL
    end
    nothing
end

last_error(c::Ctx) = unsafe_string(ccall((:gpslc_last_error, lib), Cstring, (Ptr{Cvoid},), c.h))
version() = unsafe_string(ccall((:gpslc_version, lib), Cstring, ()))

"""0 = ok; > 0 = the 1-based pivot at which a Cholesky broke down -> PosDefException, what PDMats raises inside
Gen.mvnormal; < 0 = error with the library's message."""
function check(c::Ctx, st::Integer)
    st == 0 && return nothing
    st > 0 && throw(PosDefException(st))
    error("gpslc status $st: " * last_error(c))
end

# marshalling: Float64 column-major arrays; Bool / Int inputs promote as they do on subtraction (src/kernel.jl:17)
f64(x::Nothing) = nothing
f64(x::Array{Float64}) = x
f64(x::AbstractArray{<:Real}) = convert(Array{Float64}, collect(x))
f64(x::AbstractArray) = throw(ArgumentError("GPSLCHip.f64: an array of $(eltype(x)) is not a numeric block — nested " *
    "vectors (a `Confounders` value such as [[1.0]]) go through `_umat` first"))
f64(x::Number) = Float64[x]
ptr(::Nothing) = Ptr{Float64}(C_NULL)
ptr(x::Array{Float64}) = pointer(x)

function set_data!(c::Ctx, X, T, Y)
    Xf, Tf, Yf = f64(X), f64(T), f64(Y)
    GC.@preserve Xf Tf Yf check(c, ccall((:gpslc_set_data, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), c.h, ptr(Xf), ptr(Tf), ptr(Yf)))
    c.data = (Xf, Tf, Yf)
    nothing
end

"""Device pointers (AMDGPU.jl ROCArray memory on the ctx's device)."""
set_data_dev!(c::Ctx, X::Ptr{Float64}, T::Ptr{Float64}, Y::Ptr{Float64}) = check(c, ccall((:gpslc_set_data_dev, lib), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), c.h, X, T, Y))

set_tuning!(c::Ctx; max_batch::Integer=0, panel_tiles::Integer=0, n_streams::Integer=0) = check(c,
    ccall((:gpslc_set_tuning, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32), c.h, max_batch, panel_tiles, n_streams))
# persistent factorisation launch for chunks of >= min_matrices matrices of min_tiles .. max_tiles tiles per side
# (max_tiles = 0: one launch per tile column; 0 / -1 / 0 / 0: keep)
set_task_schedule!(c::Ctx; min_tiles::Integer=0, max_tiles::Integer=-1, min_matrices::Integer=0, group::Integer=0) = check(c,
    ccall((:gpslc_set_task_schedule, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32), c.h, min_tiles, max_tiles, min_matrices, group))

"""Placement of the next calls' posterior samples inside a larger ensemble (a rank's block [s0, s1) of S_total): the
library's own normals then do not depend on how the ensemble is sharded.  S_total = 0 restores the default."""
set_ensemble!(c::Ctx, sample_offset::Integer, S_total::Integer) = check(c,
    ccall((:gpslc_set_ensemble, lib), Cint, (Ptr{Cvoid}, Int64, Int64), c.h, sample_offset, S_total))

# ---- src/kernel.jl ----------------------------------------------------------------------------------------------
function rbf_log(c::Ctx, X1::Array{Float64}, X2::Array{Float64}, ls::Vector{Float64})
    n, d = size(X1, 1), size(X1, 2)
    out = Matrix{Float64}(undef, n, n)
    GC.@preserve X1 X2 ls out check(c, ccall((:gpslc_rbf_log, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int32, Ptr{Float64}, Int32, Ptr{Float64}),
        c.h, pointer(X1), pointer(X2), n, d, pointer(ls), length(ls), pointer(out)))
    out
end

rbf_log_dev(c::Ctx, X1::Ptr{Float64}, X2::Ptr{Float64}, n::Integer, d::Integer, ls::Ptr{Float64}, ls_len::Integer,
            out::Ptr{Float64}) = check(c, ccall((:gpslc_rbf_log_dev, lib), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int32, Ptr{Float64}, Int32, Ptr{Float64}),
    c.h, X1, X2, n, d, ls, ls_len, out))

function process_cov(c::Ctx, logcov::Array{Float64}, scale::Float64, noise::Float64)
    n = size(logcov, 1)
    out = similar(logcov)
    GC.@preserve logcov out check(c, ccall((:gpslc_process_cov, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Float64, Ptr{Float64}),
        c.h, pointer(logcov), n, scale, noise, pointer(out)))
    out
end

process_cov_dev(c::Ctx, logcov::Ptr{Float64}, n::Integer, scale::Float64, noise::Float64, out::Ptr{Float64}) =
    check(c, ccall((:gpslc_process_cov_dev, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Float64, Float64, Ptr{Float64}), c.h, logcov, n, scale, noise, out))

# ---- Gen nodes (src/model_likelihood.jl, src/model_prior.jl) ----------------------------------------------------
"""log N(Y; 0, Ycov) of the :Y node for S parameter sets (src/model_likelihood.jl:83-120)."""
function y_logpdf(c::Ctx, S::Integer, U, X_or_nothing, Y_or_nothing, uyLS, xyLS, tyLS::Vector{Float64},
                  yScale::Vector{Float64}, yNoise::Vector{Float64})
    out = Vector{Float64}(undef, S)
    Uf, Xf, Yf, uy, xy = f64(U), f64(X_or_nothing), f64(Y_or_nothing), f64(uyLS), f64(xyLS)
    GC.@preserve Uf Xf Yf uy xy tyLS yScale yNoise out check(c, ccall((:gpslc_y_logpdf, lib), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, S, ptr(Uf), ptr(Xf), ptr(Yf), ptr(uy), ptr(xy), pointer(tyLS), pointer(yScale), pointer(yNoise),
        pointer(out)))
    out
end

"""log N(target; 0, scale exp.(rbfKernelLog(F, F, ls)) + noise I) for S parameter sets: :X => k => :X, :T / :logitT."""
function gp_logpdf(c::Ctx, S::Integer, nF::Integer, F::Array{Float64}, f_shared::Bool, ls::Array{Float64},
                   scale::Vector{Float64}, noise::Vector{Float64}, target::Array{Float64}, t_shared::Bool)
    out = Vector{Float64}(undef, S)
    GC.@preserve F ls scale noise target out check(c, ccall((:gpslc_gp_logpdf, lib), Cint,
        (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Int32, Ptr{Float64}),
        c.h, S, nF, pointer(F), f_shared ? 1 : 0, pointer(ls), pointer(scale), pointer(noise), pointer(target),
        t_shared ? 1 : 0, pointer(out)))
    out
end

"""The fused whole-model score: every node one Gen `update` re-scores (src/model.jl:11-131), one call.  The arrays
the nodes point into must be kept alive by the caller (GC.@preserve) for the duration of the call."""
function nodes_logpdf(c::Ctx, nodes::Vector{GPSLCNode})
    out = Vector{Float64}(undef, length(nodes))
    GC.@preserve nodes out check(c, ccall((:gpslc_nodes_logpdf, lib), Cint,
        (Ptr{Cvoid}, Int32, Ptr{GPSLCNode}, Ptr{Float64}), c.h, length(nodes), pointer(nodes), pointer(out)))
    out
end

"""draws[:, i] = chol(K_i) * target_i (target = the caller's standard normals): Gen's mvnormal(zeros(n), cov) inside
elliptical_slice(trace, :logitT, zeros(n), logitTCov) (src/inference.jl:225-232, 286-289, 343-348)."""
function nodes_draw(c::Ctx, nodes::Vector{GPSLCNode}; want_logpdf::Bool=false)
    draws = Matrix{Float64}(undef, c.n, length(nodes))
    lp = want_logpdf ? Vector{Float64}(undef, length(nodes)) : nothing
    GC.@preserve nodes draws lp check(c, ccall((:gpslc_nodes_draw, lib), Cint,
        (Ptr{Cvoid}, Int32, Ptr{GPSLCNode}, Ptr{Float64}, Ptr{Float64}),
        c.h, length(nodes), pointer(nodes), pointer(draws), ptr(lp)))
    want_logpdf ? (draws, lp) : draws
end

"""log N(x_s; 0, covscale_s * cov): the :U => u => :U nodes (src/model_likelihood.jl:4-10).  Pass `cov` once
(S = 0 just hands it over), then `nothing`."""
function mvn_logpdf(c::Ctx, cov, covscale, x)
    S = x === nothing ? 0 : size(x, 2)
    out = Vector{Float64}(undef, S)
    cf, sf, xf = f64(cov), f64(covscale), f64(x)
    GC.@preserve cf sf xf out check(c, ccall((:gpslc_mvn_logpdf, lib), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, S, ptr(cf), ptr(sf), ptr(xf), pointer(out)))
    out
end

"""chol(covscale_s * cov) z_s per column of `z`: Gen's `mvnormal(zeros(n), uCov)` of `elliptical_slice(trace, :U => k => :U,
zeros(n), uCov)` (src/inference.jl:48-54) and of generateUfromSigmaU's prior draw, with the host's normals; `cov = nothing`
re-uses the covariance `mvn_logpdf` cached."""
function mvn_draw(c::Ctx, cov, covscale, z)
    S = size(z, 2)
    out = Matrix{Float64}(undef, c.n, S)
    cf, sf, zf = f64(cov), f64(covscale), f64(z)
    GC.@preserve cf sf zf out check(c, ccall((:gpslc_mvn_draw, lib), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, S, ptr(cf), ptr(sf), ptr(zf), pointer(out)))
    out
end

# ---- the posterior pack: extractParameters stacked for nBurnIn:stepSize:nOuter (src/utils.jl:92-124) ------------
struct Pack
    U::Union{Array{Float64,3},Nothing}        # n x nU x S
    uyLS::Union{Matrix{Float64},Nothing}      # nU x S
    xyLS::Union{Matrix{Float64},Nothing}      # nX x S
    tyLS::Vector{Float64}                     # S
    yNoise::Vector{Float64}
    yScale::Vector{Float64}
end

# ---- src/estimation.jl, src/driver.jl, src/prediction.jl --------------------------------------------------------
"""The ensemble driver.  Returns (meanSATE S x L, varSATE S x L, meanITE n x S x L | nothing, ite L x n x (S*spp) |
nothing).  `z`: the caller's standard normals n x spp x S x L (Julia's RNG stays in charge) or nothing + `seed`."""
function predict(c::Ctx, p::Pack, doT::Vector{Float64}, pred_noise::Float64; spp::Integer=0, seed::Integer=0,
                 z=nothing, want_mean_ite::Bool=false, want_draws::Bool=false)
    S, L, n = length(p.tyLS), length(doT), c.n
    mS, vS = Matrix{Float64}(undef, S, L), Matrix{Float64}(undef, S, L)
    mI = want_mean_ite ? Array{Float64}(undef, n, S, L) : nothing
    dr = want_draws ? Array{Float64}(undef, L, n, S * spp) : nothing
    zf = f64(z)
    GC.@preserve p doT zf mS vS mI dr check(c, ccall((:gpslc_predict, lib), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Int32, Ptr{Float64}, Float64, Int32, UInt64, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, S, ptr(p.U), ptr(p.uyLS), ptr(p.xyLS), pointer(p.tyLS), pointer(p.yScale), pointer(p.yNoise),
        L, pointer(doT), pred_noise, spp, seed, ptr(zf),
        pointer(mS), pointer(vS), ptr(mI), ptr(dr)))
    mS, vS, mI, dr
end

"""`predict` sharded over several GPUs of one node — `cs` = one context per device (`Ctx(n, nX, nU; device=k)`, each with
the data: `set_data!` on every one), the posterior samples split into contiguous blocks, one host thread per context inside the
library, every device copying its block of the results into these host arrays.  Same results as `predict(cs[1], …)` over all S
samples, bit for bit, seeded draws included.  Returns (meanSATE, varSATE, meanITE | nothing, ite | nothing, info)."""
function predict_multi(cs::Vector{Ctx}, p::Pack, doT::Vector{Float64}, pred_noise::Float64; spp::Integer=0, seed::Integer=0,
                       z=nothing, want_mean_ite::Bool=false, want_draws::Bool=false)
    S, L, n = length(p.tyLS), length(doT), cs[1].n
    mS, vS = Matrix{Float64}(undef, S, L), Matrix{Float64}(undef, S, L)
    mI = want_mean_ite ? Array{Float64}(undef, n, S, L) : nothing
    dr = want_draws ? Array{Float64}(undef, L, n, S * spp) : nothing
    info = Vector{Int32}(undef, S)
    zf = f64(z)
    ctxs = Ptr{Cvoid}[c.h for c in cs]
    GC.@preserve cs ctxs p doT zf mS vS mI dr info check(cs[1], ccall((:gpslc_predict_multi, lib), Cint,
        (Int32, Ptr{Ptr{Cvoid}}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Int32, Ptr{Float64}, Float64, Int32, UInt64, Ptr{Float64},
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
        length(ctxs), pointer(ctxs), S, ptr(p.U), ptr(p.uyLS), ptr(p.xyLS), pointer(p.tyLS), pointer(p.yScale), pointer(p.yNoise),
        L, pointer(doT), pred_noise, spp, seed, ptr(zf),
        pointer(mS), pointer(vS), ptr(mI), ptr(dr), pointer(info)))
    mS, vS, mI, dr, info
end

"""shard_range(S, nblocks, k): the 0-based half-open block [s0, s1) of posterior samples `predict_multi` gives context k
(k = 0 … nblocks − 1) — for callers who leave every block's results on its own device (`predict_dev` per context after
`set_ensemble!(ctx_k, s0, S)`)."""
function shard_range(S::Integer, nblocks::Integer, k::Integer)
    s0, s1 = Ref{Int64}(0), Ref{Int64}(0)
    st = ccall((:gpslc_shard_range, lib), Cint, (Int64, Int32, Int32, Ref{Int64}, Ref{Int64}), S, nblocks, k, s0, s1)
    st == 0 || error("gpslc_shard_range: status $st")
    s0[], s1[]
end

"""As `predict`, every array argument a DEVICE pointer (ROCArray memory); outputs stay in HBM."""
predict_dev(c::Ctx, S::Integer, U::Ptr{Float64}, uyLS::Ptr{Float64}, xyLS::Ptr{Float64}, tyLS::Ptr{Float64},
            yScale::Ptr{Float64}, yNoise::Ptr{Float64}, L::Integer, doT::Ptr{Float64}, pred_noise::Float64,
            spp::Integer, seed::Integer, z::Ptr{Float64}, meanSATE::Ptr{Float64}, varSATE::Ptr{Float64},
            meanITE::Ptr{Float64}, ite_draws::Ptr{Float64}) = check(c, ccall((:gpslc_predict_dev, lib), Cint,
    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
     Int32, Ptr{Float64}, Float64, Int32, UInt64, Ptr{Float64},
     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
    c.h, S, U, uyLS, xyLS, tyLS, yScale, yNoise, L, doT, pred_noise, spp, seed, z, meanSATE, varSATE, meanITE, ite_draws))

"""ITEDistributions with the reference's output layout: MeanITEs S x n, CovITEs S x n x n (jitter included)."""
function ite_distributions(c::Ctx, p::Pack, doT::Float64, pred_noise::Float64; want_cov::Bool=true)
    S, n = length(p.tyLS), c.n
    M = Matrix{Float64}(undef, S, n)
    Cv = want_cov ? Array{Float64}(undef, S, n, n) : nothing
    GC.@preserve p M Cv check(c, ccall((:gpslc_ite_distributions, lib), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Float64, Float64, Ptr{Float64}, Ptr{Float64}),
        c.h, S, ptr(p.U), ptr(p.uyLS), ptr(p.xyLS), pointer(p.tyLS), pointer(p.yScale), pointer(p.yNoise),
        doT, pred_noise, pointer(M), ptr(Cv)))
    M, Cv
end

"""The seven dense blocks of likelihoodDistribution for one parameter set (src/likelihood.jl:8-174)."""
function likelihood_distribution(c::Ctx, U, uyLS, xyLS, tyLS::Float64, yScale::Float64, yNoise::Float64, doT::Float64)
    n = c.n
    blocks = [Matrix{Float64}(undef, n, n) for _ in 1:7]
    Uf, uy, xy = f64(U), f64(uyLS), f64(xyLS)
    GC.@preserve Uf uy xy blocks check(c, ccall((:gpslc_likelihood_distribution, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, ptr(Uf), ptr(uy), ptr(xy), tyLS, yScale, yNoise, doT,
        pointer(blocks[1]), pointer(blocks[2]), pointer(blocks[3]), pointer(blocks[4]), pointer(blocks[5]),
        pointer(blocks[6]), pointer(blocks[7])))
    blocks      # CovWW, CovWWs, CovWWp, CovC11, CovC12, CovC21, CovC22
end

"""SATEsamples: mean[j] + var[j] * z — the variance used as sigma, as the reference does (src/estimation.jl:159)."""
function sate_samples(meanSATE::Vector{Float64}, varSATE::Vector{Float64}, spp::Integer; seed::Integer=0, z=nothing)
    S = length(meanSATE)
    out = Vector{Float64}(undef, S * spp)
    zf = f64(z)
    GC.@preserve meanSATE varSATE zf out begin
        st = ccall((:gpslc_sate_samples, lib), Cint,
            (Ptr{Float64}, Ptr{Float64}, Int64, Int32, UInt64, Ptr{Float64}, Ptr{Float64}),
            pointer(meanSATE), pointer(varSATE), S, spp, seed, ptr(zf), pointer(out))
        st == 0 || error("gpslc_sate_samples: status $st")
    end
    out
end

"""Per-individual mean and the two type-7 quantiles of an n x m sample matrix (src/driver.jl:129-149)."""
function summarize(c::Ctx, samples::Matrix{Float64}, credible_interval::Float64)
    n, m = size(samples)
    mean, lower, upper = Vector{Float64}(undef, n), Vector{Float64}(undef, n), Vector{Float64}(undef, n)
    GC.@preserve samples mean lower upper check(c, ccall((:gpslc_summarize, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, pointer(samples), n, m, credible_interval, pointer(mean), pointer(lower), pointer(upper)))
    mean, lower, upper
end

summarize_dev(c::Ctx, samples::Ptr{Float64}, n::Integer, m::Integer, row_stride::Integer, col_stride::Integer,
              credible_interval::Float64, mean::Ptr{Float64}, lower::Ptr{Float64}, upper::Ptr{Float64}) =
    check(c, ccall((:gpslc_summarize_dev, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Int64, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        c.h, samples, n, m, row_stride, col_stride, credible_interval, mean, lower, upper))

"""1-based failing pivot (0 = ok) of every posterior sample of the last call."""
function last_info(c::Ctx, S::Integer)
    info = Vector{Int32}(undef, S)
    GC.@preserve info check(c, ccall((:gpslc_last_info, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Int64), c.h, pointer(info), S))
    info
end

# ---- posterior pack file (replaces Serialization for the prediction path, src/io.jl:14-34) -----------------------
function pack_save(path::AbstractString, h::PackHeader, X, T, Y, p::Pack)
    Xf, Tf, Yf = f64(X), f64(T), f64(Y)
    GC.@preserve Xf Tf Yf p begin
        st = ccall((:gpslc_pack_save, lib), Cint,
            (Cstring, Ref{PackHeader}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            path, Ref(h), ptr(Xf), ptr(Tf), ptr(Yf), ptr(p.U), ptr(p.uyLS), ptr(p.xyLS), pointer(p.tyLS),
            pointer(p.yNoise), pointer(p.yScale))
        st == 0 || error("gpslc_pack_save: status $st")
    end
    nothing
end

function pack_read_header(path::AbstractString)
    h = Ref(PackHeader())
    st = ccall((:gpslc_pack_read_header, lib), Cint, (Cstring, Ref{PackHeader}), path, h)
    st == 0 || error("gpslc_pack_read_header: status $st")
    h[]
end

"""Data and the posterior samples [s0, s1) (0-based, half-open) of a pack: what one rank of a sharded prediction loads."""
function pack_load(path::AbstractString, s0::Integer, s1::Integer)
    h = pack_read_header(path)
    n, nX, nU, S = h.n, h.nX, h.nU, s1 - s0
    X = nX > 0 ? Matrix{Float64}(undef, n, nX) : nothing
    T, Y = Vector{Float64}(undef, n), Vector{Float64}(undef, n)
    U = nU > 0 ? Array{Float64}(undef, n, nU, S) : nothing
    uy = nU > 0 ? Matrix{Float64}(undef, nU, S) : nothing
    xy = nX > 0 ? Matrix{Float64}(undef, nX, S) : nothing
    ty, yn, ys = Vector{Float64}(undef, S), Vector{Float64}(undef, S), Vector{Float64}(undef, S)
    GC.@preserve X T Y U uy xy ty yn ys begin
        st = ccall((:gpslc_pack_load, lib), Cint,
            (Cstring, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            path, s0, s1, ptr(X), pointer(T), pointer(Y), ptr(U), ptr(uy), ptr(xy), pointer(ty), pointer(yn), pointer(ys))
        st == 0 || error("gpslc_pack_load: status $st")
    end
    h, X, T, Y, Pack(U, uy, xy, ty, yn, ys)
end

# ---- measurement hooks ------------------------------------------------------------------------------------------
profile_reset!(c::Ctx) = check(c, ccall((:gpslc_profile_reset, lib), Cint, (Ptr{Cvoid},), c.h))

function profile_get(c::Ctx)
    l, ms, fl = Ref{Int64}(0), Ref{Float64}(0.0), Ref{Float64}(0.0)
    check(c, ccall((:gpslc_profile_get, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Float64}, Ref{Float64}), c.h, l, ms, fl))
    l[], ms[], fl[]
end

function profile_get_class(c::Ctx, kernel_class::Integer)
    l, ms, fl = Ref{Int64}(0), Ref{Float64}(0.0), Ref{Float64}(0.0)
    check(c, ccall((:gpslc_profile_get_class, lib), Cint, (Ptr{Cvoid}, Int32, Ref{Int64}, Ref{Float64}, Ref{Float64}),
                   c.h, kernel_class, l, ms, fl))
    l[], ms[], fl[]
end

# ---- the helpers the method bodies of part 2 use -----------------------------------------------------------------
"""The context of the kernel-only calls (rbfKernelLog / processCov carry no data set): created on first use."""
const KCTX = Ref{Union{Ctx,Nothing}}(nothing)
function kctx()
    KCTX[] === nothing && (KCTX[] = Ctx(1, 0, 0))
    KCTX[]::Ctx
end

"""dctx(n, nX, nU): a context of that shape, re-used across calls (the data set is handed over again by the caller with
`set_data!`: n (nX + 2) doubles).  The reference's parameter-level functions — conditionalITE(uyLS, …, U, X, T, Y, doT),
likelihoodDistribution — carry their data in the arguments and are called in loops (src/estimation.jl:78-84): a context
per call would put hipMalloc + stream creation inside that loop.  At most 8 shapes are kept; the ninth evicts the ONE shape used
longest ago (its context is left to its finalizer: a caller may still hold it)."""
const DCTX = Dict{NTuple{3,Int},Ctx}()
const DCTX_USED = Dict{NTuple{3,Int},Int}()          # key -> tick of its last use
const DCTX_TICK = Ref(0)
function dctx(n::Integer, nX::Integer, nU::Integer)
    key = (Int(n), Int(nX), Int(nU))
    c = get(DCTX, key, nothing)
    if c === nothing || c.h == C_NULL
        if length(DCTX) >= 8
            lru = argmin(DCTX_USED)
            delete!(DCTX, lru)
            delete!(DCTX_USED, lru)
        end
        c = Ctx(n, nX, nU)
        DCTX[key] = c
    end
    DCTX_USED[key] = (DCTX_TICK[] += 1)
    c
end

"""nctx(n): the context of the Gen-node calls, which carry their feature block and target in the arguments."""
nctx(n::Integer) = dctx(n, 0, 0)

"""One node score: log N(target; 0, scale exp.(rbfKernelLog(F, F, ls)) + noise I), F n x nF, ls nF."""
function gp_score(c::Ctx, F::Matrix{Float64}, ls::Vector{Float64}, scale::Float64, noise::Float64, target::Vector{Float64})
    size(F, 2) == length(ls) || throw(DimensionMismatch("gp_score: $(size(F, 2)) feature columns, $(length(ls)) lengthscales"))
    GC.@preserve F ls target begin
        nodes_logpdf(c, [GPSLCNode(size(F, 2), 0, pointer(F), pointer(ls), scale, noise, pointer(target))])[1]
    end
end

"""chol(scale exp.(rbfKernelLog(F, F, ls)) + noise I) * z for the caller's standard normals z (any n)."""
function draw(c::Ctx, F::Matrix{Float64}, ls::Vector{Float64}, scale::Float64, noise::Float64, z::Vector{Float64})
    size(F, 2) == length(ls) || throw(DimensionMismatch("draw: $(size(F, 2)) feature columns, $(length(ls)) lengthscales"))
    GC.@preserve F ls z begin
        nodes_draw(c, [GPSLCNode(size(F, 2), 0, pointer(F), pointer(ls), scale, noise, pointer(z))])[:, 1]
    end
end

"""log N(x; 0, covscale * cov): `cov` is handed to the library when it is not (by identity) the matrix the context
already holds — SigmaU is one constant matrix per data set (src/utils.jl:17-33) — and re-used otherwise.
Cache contract: IDENTITY (`===`), not contents.  Pass the SAME `SigmaU` array on every score and the scalar `uNoise` as
`covscale` (the replacement line does: `hip_mv_normal(SigmaU, uNoise)`); a temporary such as `SigmaU * uNoise` is a new
array each time and is uploaded and validated again on every score, and a matrix changed in place is NOT noticed —
hand over a copy (or set `c.cov = nothing`) after mutating one."""
function mvn_score(c::Ctx, cov::Matrix{Float64}, covscale::Float64, x::Vector{Float64})
    if c.cov !== cov
        mvn_logpdf(c, cov, nothing, nothing)          # S = 0: hand over + validate (PosDefException if not PD)
        c.cov, c.cov_chol = cov, nothing
    end
    mvn_logpdf(c, nothing, covscale, reshape(x, :, 1))[1]
end

end # module GPSLCHip


# =====================================================================================================================
# Part 2 — the reference's functions, same signatures, bodies on the GPU.  (Evaluated in CausalGPSLC's namespace:
# GPSLCObject, Intervention, Confounders, ... are the reference's own types, src/types.jl.)
# =====================================================================================================================

import LinearAlgebra
import Random

# ---- argument forms -----------------------------------------------------------------------------------------------
# `Confounders` (src/types.jl:85-94) is a union of an n x nU matrix, a length-n vector (nU = 1) and NESTED vectors —
# individual i is U[i], as the Vector{Vector} method of rbfKernelLog reads it (src/kernel.jl:34-42); the reference's own
# estimation tests pass U = [[1.0]] (test/test_data.jl:42).  Every place where a U enters the shim goes through _umat,
# and nU is the column count of ITS result — never size(U, 2), which is 1 for every vector form.
_umat(::Nothing) = nothing
_umat(U::AbstractMatrix{<:Real}) = GPSLCHip.f64(U)                                      # n x nU
_umat(U::AbstractVector{<:Real}) = reshape(GPSLCHip.f64(U), :, 1)                       # n x 1
_umat(U::AbstractVector{<:AbstractVector}) =                                            # n vectors of length nU -> n x nU
    Matrix{Float64}(permutedims(reduce(hcat, [GPSLCHip.f64(collect(u)) for u in U])))
function _umat(U)                                                                       # PersistentVector and friends
    V = [u for u in U]                                # a comprehension narrows the element type (collect keeps Any)
    V isa Union{AbstractVector{<:Real},AbstractVector{<:AbstractVector}} ||
        throw(ArgumentError("_umat: cannot read a $(typeof(U)) as confounders (n x nU matrix, length-n vector or n vectors of length nU)"))
    _umat(V)
end
_ncols(::Nothing) = 0
_ncols(M::AbstractMatrix) = size(M, 2)

_ls(LS::Number) = Float64[LS]
_ls(LS) = GPSLCHip.f64(vec(collect(LS)))
"""A lengthscale argument for a d-column feature block: a scalar applies to every column (src/kernel.jl:17 broadcasts)."""
_lsfor(::Nothing, d::Integer) = Float64[]
_lsfor(LS::Number, d::Integer) = fill(Float64(LS), d)
_lsfor(LS, d::Integer) = (v = _ls(LS); @assert(length(v) == d, "vector lengthscale doesn't match individual"); v)
_mat(X::AbstractMatrix) = GPSLCHip.f64(X)
_mat(X::AbstractVector) = reshape(GPSLCHip.f64(X), :, 1)
"""[U | X | T ...] as one n x nF Float64 block (parts that are `nothing` are skipped; Bool treatments become 0.0 / 1.0)
and the matching lengthscale vector."""
function _features(parts_and_ls::Pair...)
    blocks = Matrix{Float64}[]
    ls = Float64[]
    for (part, LS) in parts_and_ls
        part === nothing && continue
        M = _umat(part)
        push!(blocks, M)
        append!(ls, _lsfor(LS, size(M, 2)))
    end
    Matrix{Float64}(reduce(hcat, blocks)), ls
end
_dot(doT::Union{Bool,Float64}) = Float64(doT)
_dot(doT) = throw(ArgumentError("vector interventions: fill(doT, n) of a vector has no rbfKernelLog method in the " *
                                "reference either (src/likelihood.jl:27-28)"))

# ---- the device side of a GPSLCObject: one Ctx + one posterior pack, for as long as the object lives ----------------
# GPSLCObject is an immutable struct (src/types.jl:249-258): it can carry neither a finalizer nor be a WeakKeyDict key.
# Its `posteriorSamples::Vector{Any}` is the one field with an identity and a lifetime of its own, so the entry is keyed by
# objectid(g.posteriorSamples), holds only a WeakRef to that vector (checked on every lookup: ids can be re-used), and a
# finalizer on the vector frees the device context — hundreds of MB of HBM workspace — as soon as the object is
# collected.  The finalizer touches no Dict (finalizers run at arbitrary allocation points); dead entries (a few words
# each) are swept on the next lookup.  release!(g) does the same eagerly.
# Two GPSLCObjects may share one posteriorSamples vector and differ in data or hyper-parameters (the pack depends on
# nBurnIn:stepSize:nOuter, the context holds X, T, Y): the entry also records the identities of g.hyperparams, g.X, g.T, g.Y and is
# rebuilt when any of them differs.  The cache contract is IDENTITY: an in-place change to g.X / g.T / g.Y or to the posterior
# samples after the first prediction is not seen — call release!(g) after mutating.
mutable struct _DeviceSide
    key::WeakRef
    ids::NTuple{4,UInt}            # objectid of g.hyperparams, g.X, g.T, g.Y at the time the entry was built
    ctx::GPSLCHip.Ctx
    pack::Union{GPSLCHip.Pack,Nothing}
    multi::Dict{Vector{Int},Vector{GPSLCHip.Ctx}}      # devices => one context per entry, each holding the data (ctxs(g, devices))
end
const _GPSLC_DEVICE = Dict{UInt,_DeviceSide}()
_ids(g::GPSLCObject) = (objectid(g.hyperparams), objectid(g.X), objectid(g.T), objectid(g.Y))
function _drop!(d::_DeviceSide)
    GPSLCHip.destroy!(d.ctx)
    foreach(cs -> foreach(GPSLCHip.destroy!, cs), values(d.multi))
    empty!(d.multi)
    d.pack = nothing                                   # n x nU x S doubles on the host
    nothing
end

function _device_side(g::GPSLCObject)
    ps = g.posteriorSamples
    for (k, v) in collect(_GPSLC_DEVICE)              # sweep: entries whose vector is gone release their pack and contexts
        v.key.value === nothing && (_drop!(v); delete!(_GPSLC_DEVICE, k))
    end
    filter!(kv -> kv.second.key.value !== nothing, _GPSLC_DEVICE)
    d = get(_GPSLC_DEVICE, objectid(ps), nothing)
    if d === nothing || d.key.value !== ps || d.ids != _ids(g) || d.ctx.h == C_NULL
        d === nothing || _drop!(d)
        nX = g.X === nothing ? 0 : getNX(g)
        nU = getNU(g) === nothing ? 0 : getNU(g)
        c = GPSLCHip.Ctx(getN(g), nX, nU)
        GPSLCHip.set_data!(c, g.X, g.T, g.Y)           # Bool treatments become 0.0 / 1.0 here
        d = _DeviceSide(WeakRef(ps), _ids(g), c, nothing, Dict{Vector{Int},Vector{GPSLCHip.Ctx}}())
        _GPSLC_DEVICE[objectid(ps)] = d
        finalizer(_ -> GPSLCHip.destroy!(c), ps)
    end
    d
end

"""release!(g): free g's device contexts and cached posterior pack now (otherwise: when g is garbage-collected)."""
function release!(g::GPSLCObject)
    d = pop!(_GPSLC_DEVICE, objectid(g.posteriorSamples), nothing)
    d === nothing || (GPSLCHip.destroy!(d.ctx); _drop!(d))
    nothing
end

"""ctxs(g, devices): one context per entry of `devices` (GPU indices of this node, e.g. 0:7), each holding g.X, g.T, g.Y — what
`GPSLCHip.predict_multi` shards the posterior samples over (SURVEY.md §8e; the loop src/prediction.jl:30-33).  Created on first use
for that device list, freed with g."""
function ctxs(g::GPSLCObject, devices)
    d = _device_side(g)
    key = Int[dev for dev in devices]
    isempty(key) && throw(ArgumentError("devices must name at least one GPU"))
    get!(d.multi, key) do
        map(key) do dev
            c = GPSLCHip.Ctx(d.ctx.n, d.ctx.nX, d.ctx.nU; device=dev)
            GPSLCHip.set_data!(c, g.X, g.T, g.Y)
            c
        end
    end
end

# One GPU (devices === nothing: the object's context on device 0) or the sharded call: same tuple (mS, vS, mI, ite).
function _predict(g::GPSLCObject, devices, doT::Vector{Float64}; kw...)
    pn = g.hyperparams.predictionCovarianceNoise
    devices === nothing && return GPSLCHip.predict(ctx(g), posterior_pack(g), doT, pn; kw...)
    mS, vS, mI, ite, _ = GPSLCHip.predict_multi(ctxs(g, devices), posterior_pack(g), doT, pn; kw...)
    mS, vS, mI, ite
end

"""ctx(g): the device context holding g.X, g.T, g.Y (src/types.jl:249-258), created on first use."""
ctx(g::GPSLCObject) = _device_side(g).ctx

"""posterior_pack(g): extractParameters (src/utils.jl:92-124) for i in nBurnIn:stepSize:nOuter
(src/estimation.jl:72, 78 — the burn-in index itself included), stacked along a trailing sample axis."""
function posterior_pack(g::GPSLCObject)
    d = _device_side(g)
    if d.pack === nothing
        idx = g.hyperparams.nBurnIn:g.hyperparams.stepSize:g.hyperparams.nOuter
        ps = [extractParameters(g, i) for i in idx]            # (uyLS, xyLS, tyLS, yNoise, yScale, U)
        hasU, hasX = ps[1][1] !== nothing, ps[1][2] !== nothing
        d.pack = GPSLCHip.Pack(hasU ? cat((_umat(p[6]) for p in ps)...; dims=3) : nothing,
                               hasU ? reduce(hcat, [Float64.(p[1]) for p in ps]) : nothing,
                               hasX ? reduce(hcat, [Float64.(vec(p[2])) for p in ps]) : nothing,
                               Float64[p[3] for p in ps], Float64[p[4] for p in ps], Float64[p[5] for p in ps])
    end
    d.pack::GPSLCHip.Pack
end

# ---- src/kernel.jl ------------------------------------------------------------------------------------------------
function rbfKernelLogScalar(Xi::SupportedRBFVector, Xiprime::SupportedRBFVector, LS::SupportedRBFLengthscale)   # :13-19
    @assert (size(LS, 1) == size(Xi, 1) || size(LS) == ()) "vector lengthscale doesn't match individual"
    # five flops for one pair of individuals: stays on the host, in the reference's own arithmetic (src/kernel.jl:17) — a context
    # lookup, two uploads, a launch and a download would cost 10^4 times the sum; the matrix methods below are the GPU's
    return -sum((Xi .- Xiprime) .^ 2 ./ LS .^ 2)
end

function rbfKernelLog(X1::SupportedRBFMatrix, X2::SupportedRBFMatrix, LS::SupportedRBFLengthscale)              # :24-32
    @assert size(X1) == size(X2) "X1 and X2 are different sizes!"
    GPSLCHip.rbf_log(GPSLCHip.kctx(), _mat(collect(X1)), _mat(collect(X2)), _ls(LS))
end

function rbfKernelLog(X1::SupportedRBFData, X2::SupportedRBFData, LS::SupportedRBFLengthscale)                  # :34-42
    @assert size(X1) == size(X2) "X1 and X2 are different sizes!"
    GPSLCHip.rbf_log(GPSLCHip.kctx(), _umat(X1), _umat(X2), _ls(LS))     # Vector{Vector}: individual i is X1[i]
end

function processCov(logCov::Union{Float64,Array{Float64}}, scale::Union{Float64,Array{Float64}}, noise::Float64)  # :53-55
    (logCov isa Float64 || !(scale isa Float64)) && return exp.(logCov) * scale + 1LinearAlgebra.I * noise   # scalar / array-scale forms stay on the host
    GPSLCHip.process_cov(GPSLCHip.kctx(), logCov, scale, noise)
end

function processCov(logCov::Union{Float64,Array{Float64}}, scale::Float64)                                        # :57-59
    logCov isa Float64 && return exp(logCov) * scale
    GPSLCHip.process_cov(GPSLCHip.kctx(), logCov, scale, 0.0)
end

# ---- src/likelihood.jl: the four methods differ only in which of U / X are `nothing` ------------------------------
function _likelihood_blocks(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT)
    n = size(Y, 1)
    Um = _umat(U)
    c = GPSLCHip.dctx(n, X === nothing ? 0 : size(X, 2), _ncols(Um))
    GPSLCHip.set_data!(c, X, T, Y)
    b = GPSLCHip.likelihood_distribution(c, Um, uyLS, xyLS, tyLS, yScale, yNoise, _dot(doT))
    # CovWW and CovWWp come back as the reference returns them, wrapped in Symmetric (src/likelihood.jl:31-32): a caller's
    # `CovWWp \ y` then dispatches to Bunch-Kaufman as it does there, not to LU
    Y, LinearAlgebra.Symmetric(b[1]), b[2], LinearAlgebra.Symmetric(b[3]), b[4], b[5], b[6], b[7]   # Y, CovWW, CovWWs, CovWWp, CovC11, CovC12, CovC21, CovC22 (:51)
end

function likelihoodDistribution(uyLS::Vector{Float64}, xyLS::Array{Float64}, tyLS::Float64, yNoise::Float64, yScale::Float64,
                                U::Confounders, X::Covariates, T::Treatment, Y::Outcome, doT::Intervention)      # :8-52
    n = size(Y, 1)
    @assert size(U, 1) == n
    @assert size(X, 1) == n
    @assert size(T, 1) == n
    _likelihood_blocks(uyLS, xyLS, tyLS, yNoise, yScale, U, X, T, Y, doT)
end

function likelihoodDistribution(uyLS::Vector{Float64}, xyLS::Nothing, tyLS::Float64, yNoise::Float64, yScale::Float64,
                                U::Confounders, X::Nothing, T::Treatment, Y::Outcome, doT::Intervention)         # :55-94
    n = size(Y, 1)
    @assert size(U, 1) == n
    @assert size(T, 1) == n
    _likelihood_blocks(uyLS, nothing, tyLS, yNoise, yScale, U, nothing, T, Y, doT)
end

function likelihoodDistribution(uyLS::Nothing, xyLS::Vector{Float64}, tyLS::Float64, yNoise::Float64, yScale::Float64,
                                U::Nothing, X::Covariates, T::Treatment, Y::Outcome, doT::Intervention)          # :97-136
    n = size(Y, 1)
    @assert size(X, 1) == n
    @assert size(T, 1) == n
    _likelihood_blocks(nothing, xyLS, tyLS, yNoise, yScale, nothing, X, T, Y, doT)
end

function likelihoodDistribution(uyLS::Nothing, xyLS::Nothing, tyLS::Float64, yNoise::Float64, yScale::Float64,
                                U::Nothing, X::Nothing, T::Treatment, Y::Outcome, doT::Intervention)             # :139-174
    @assert size(T, 1) == size(Y, 1)
    _likelihood_blocks(nothing, nothing, tyLS, yNoise, yScale, nothing, nothing, T, Y, doT)
end

# ---- src/estimation.jl --------------------------------------------------------------------------------------------
function conditionalITE(uyLS::Union{Vector{Float64},Nothing}, xyLS::Union{Array{Float64},Nothing}, tyLS::Float64,
                        yNoise::Float64, yScale::Float64, U::Union{Confounders,Nothing}, X::Union{Covariates,Nothing},
                        T::Treatment, Y::Outcome, doT::Intervention)                                              # :36-50
    n = size(Y, 1)
    Um = _umat(U)
    c = GPSLCHip.dctx(n, X === nothing ? 0 : size(X, 2), _ncols(Um))
    GPSLCHip.set_data!(c, X, T, Y)
    p = GPSLCHip.Pack(Um === nothing ? nothing : reshape(Um, n, :, 1),
                      uyLS === nothing ? nothing : reshape(copy(uyLS), :, 1),
                      xyLS === nothing ? nothing : reshape(GPSLCHip.f64(vec(xyLS)), :, 1),
                      [tyLS], [yNoise], [yScale])
    M, Cv = GPSLCHip.ite_distributions(c, p, _dot(doT), 0.0)             # CovITE itself: the jitter is ITEDistributions' (:82)
    M[1, :], Cv[1, :, :]
end

function conditionalITE(g::GPSLCObject, psindex::Int64, doT::Intervention)                                        # :57-60
    uyLS, xyLS, tyLS, yNoise, yScale, U = extractParameters(g, psindex)
    conditionalITE(uyLS, xyLS, tyLS, yNoise, yScale, U, g.X, g.T, g.Y, doT)
end

function ITEDistributions(g::GPSLCObject, doT::Intervention)                                                       # :66-86
    GPSLCHip.ite_distributions(ctx(g), posterior_pack(g), _dot(doT), g.hyperparams.predictionCovarianceNoise)
end

function SATEDistributions(g::GPSLCObject, doT::Intervention; devices=nothing)                                    # :127-140
    mS, vS, _, _ = _predict(g, devices, [_dot(doT)])
    mS[:, 1], vS[:, 1]          # O(N^2) per posterior sample: the N x N covariance is never formed
end

# ---- src/driver.jl ------------------------------------------------------------------------------------------------
# Where the standard normals of the draws come from.  `seed = nothing` (default): Julia's global RNG draws them on the
# host, n x spp x S x L, exactly the stream Gen.mvnormal would consume (src/estimation.jl:105) — as long as that tensor
# stays under _HOST_NORMALS_MAX bytes; beyond (BASELINE config 4: 4096 x 10 x 8192 x 64 doubles = 172 GB) ONE UInt64 is
# drawn from the global RNG and seeds the library's Philox4x32-10 stream on the device (DESIGN.md §5), so Random.seed!
# still determines the result.  `seed = k`: the Philox stream with that seed, no host tensor at all.
const _HOST_NORMALS_MAX = 2^31
function _normals(n, spp, S, L, seed)
    seed === nothing || return (UInt64(seed), nothing)
    8 * n * spp * S * L > _HOST_NORMALS_MAX && return (rand(Random.default_rng(), UInt64), nothing)
    (UInt64(0), randn(n, spp, S, L))
end

function sampleITE(g::GPSLCObject, doT::Intervention; samplesPerPosterior::Int64=10,
                   seed::Union{Nothing,Integer}=nothing, devices=nothing)                                          # :86-89
    n, S = getN(g), getNumPosteriorSamples(g)
    sd, z = _normals(n, samplesPerPosterior, S, 1, seed)
    _, _, _, ite = _predict(g, devices, [_dot(doT)]; spp=samplesPerPosterior, seed=sd, z=z, want_draws=true)
    ite[1, :, :]                                                          # n x (S * spp), sample outer / draw inner (:100-107)
end

function sampleSATE(g::GPSLCObject, doT::Intervention; samplesPerPosterior::Int64=10,
                    seed::Union{Nothing,Integer}=nothing, devices=nothing)                                         # :108-111
    MeanSATEs, VarSATEs = SATEDistributions(g, doT; devices=devices)
    seed === nothing || return GPSLCHip.sate_samples(MeanSATEs, VarSATEs, samplesPerPosterior; seed=UInt64(seed))
    z = randn(length(MeanSATEs) * samplesPerPosterior)
    GPSLCHip.sate_samples(MeanSATEs, VarSATEs, samplesPerPosterior; z=z)  # normal(mean, var): variance as sigma (:159)
end

function summarizeEstimates(samples; savetofile::String="", credible_interval::Float64=0.90)                        # :129-149
    Mean, lowerBound, upperBound = GPSLCHip.summarize(GPSLCHip.kctx(), Matrix{Float64}(samples), credible_interval)
    df = DataFrame(Individual=1:size(Mean, 1), Mean=Mean, LowerBound=lowerBound, UpperBound=upperBound)
    if savetofile != ""
        CSV.write(savetofile, df)
        println("Saved mean and 90% credible intervals to " * savetofile)
    end
    return df
end

# ---- src/prediction.jl --------------------------------------------------------------------------------------------
function predictCounterfactualEffects(g::GPSLCObject, nSamplesPerMixture::Int64; fidelity::Int64=100,
                                      minDoT=min(g.T...), maxDoT=max(g.T...),
                                      seed::Union{Nothing,Integer}=nothing, devices=nothing)                       # :23-36
    delta = abs(maxDoT - minDoT)
    step = delta / fidelity
    doTrange = minDoT:step:maxDoT                                          # :24-28
    L, n, S = length(doTrange), getN(g), getNumPosteriorSamples(g)
    sd, z = _normals(n, nSamplesPerMixture, S, L, seed)
    # one factorisation of A per posterior sample, shared by its L levels; `ite` comes back in the reference's
    # layout (L x n x S*spp, level index fastest).  devices = 0:7 shards the posterior samples — the loop of :30-33 — over the
    # eight GPUs of a node: ONE ccall (gpslc_predict_multi), same tensor bit for bit
    _, _, _, ite = _predict(g, devices, Float64.(collect(doTrange)); spp=nSamplesPerMixture, seed=sd, z=z, want_draws=true)
    return ite, doTrange
end

# ---- src/model_likelihood.jl, src/model_prior.jl: the Gaussian nodes of the Gen models ------------------------------
# Every node keeps its ADDRESS and its VALUE TYPE (Vector{Float64}); only the distribution object inside @trace changes,
# so src/inference.jl (Gen.mh on the addresses of src/proposal.jl:8-22, elliptical_slice on :U => k => :U and :logitT)
# runs unchanged and scores these nodes on the GPU.  Three distributions, one per node kind:
#   HipGPNormal   N(0, scale exp.(rbfKernelLog(F, F, ls)) + noise I)   :X => k => :X, :T, :logitT (and :Y)
#   HipYNormal    the :Y node with the data set's X / T taken from a Ctx (no feature block to assemble per call)
#   HipMvNormal   N(0, covscale * cov) for a dense, constant cov         :U => u => :U (cov = SigmaU)
# Replacement text for the @gen bodies, line for line:
#   generateU (src/model_prior.jl:27-30), mapped by generateUfromSigmaU (src/model_likelihood.jl:4-10):
#       @gen function generateU(SigmaU::Matrix{Float64}, uNoise::Float64, n::Int64)::Vector{Float64}
#           @trace(hip_mv_normal(SigmaU, uNoise), :U)
#       end
#       U = @trace(MappedGenerateU(fill(SigmaU, nU), fill(uNoise, nU), fill(n, nU)), :U)      # no SigmaU * uNoise per call
#   generateXfromU (src/model_likelihood.jl:13-22), loop body for k:
#       F, ls = _features(U => uxLS[k, :])
#       X[:, k] = @trace(hip_gp_normal(F, ls, xScale[k], xNoise[k]), :X => k => :X)
#   generateRealTfromUX / U / X (:36-44, :57-62, :77-83)  —  binary: address :logitT (:25-33, :47-54, :65-74):
#       F, ls = _features(U => utLS, X => xtLS)
#       T = @trace(hip_gp_normal(F, ls, tScale, tNoise), :T)
#   generateYfromUXT / UT / XT / T (:83-120):
#       F, ls = _features(U => uyLS, X => xyLS, T => tyLS)
#       Y = @trace(hip_gp_normal(F, ls, yScale, yNoise), :Y)
#     or, with the data set's context:   Y = @trace(hip_y_normal(c, U, X, uyLS, xyLS, tyLS, yScale, yNoise), :Y)
function _register_continuous(D)                  # Gen.is_discrete is part of the Distribution interface where it exists
    if isdefined(Gen, :is_discrete)
        @eval Gen.is_discrete(::$D) = false
    end
    nothing
end

struct HipGPNormal <: Gen.Distribution{Vector{Float64}} end
const hip_gp_normal = HipGPNormal()
(d::HipGPNormal)(args...) = Gen.random(d, args...)

function Gen.logpdf(::HipGPNormal, x::AbstractVector{<:Real}, F::Matrix{Float64}, ls::Vector{Float64}, scale::Float64,
                    noise::Float64)
    GPSLCHip.gp_score(GPSLCHip.nctx(length(x)), F, ls, scale, noise, GPSLCHip.f64(x))
end

function Gen.random(::HipGPNormal, F::Matrix{Float64}, ls::Vector{Float64}, scale::Float64, noise::Float64)
    n = size(F, 1)
    GPSLCHip.draw(GPSLCHip.nctx(n), F, ls, scale, noise, randn(n))      # any n: gpslc_nodes_draw (tiled path beyond n = 640)
end
Gen.has_output_grad(::HipGPNormal) = false
Gen.has_argument_grads(::HipGPNormal) = ntuple(_ -> false, 4)              # F, ls, scale, noise
Gen.logpdf_grad(::HipGPNormal, x, args...) = ntuple(_ -> nothing, 5)       # output + 4 arguments
_register_continuous(HipGPNormal)

struct HipMvNormal <: Gen.Distribution{Vector{Float64}} end
const hip_mv_normal = HipMvNormal()
(d::HipMvNormal)(args...) = Gen.random(d, args...)

function Gen.logpdf(::HipMvNormal, x::AbstractVector{<:Real}, cov::Matrix{Float64}, covscale::Float64)
    GPSLCHip.mvn_score(GPSLCHip.nctx(length(x)), cov, covscale, GPSLCHip.f64(x))
end

function Gen.random(::HipMvNormal, cov::Matrix{Float64}, covscale::Float64)
    c = GPSLCHip.nctx(size(cov, 1))
    if c.cov !== cov || c.cov_chol === nothing
        c.cov !== cov && GPSLCHip.mvn_logpdf(c, cov, nothing, nothing)
        c.cov, c.cov_chol = cov, LinearAlgebra.cholesky(LinearAlgebra.Symmetric(cov)).L
    end
    sqrt(covscale) * (c.cov_chol * randn(size(cov, 1)))
end
Gen.has_output_grad(::HipMvNormal) = false
Gen.has_argument_grads(::HipMvNormal) = ntuple(_ -> false, 2)              # cov, covscale
Gen.logpdf_grad(::HipMvNormal, x, args...) = ntuple(_ -> nothing, 3)       # output + 2 arguments
_register_continuous(HipMvNormal)

struct HipYNormal <: Gen.Distribution{Vector{Float64}} end
const hip_y_normal = HipYNormal()
(d::HipYNormal)(args...) = Gen.random(d, args...)

function Gen.logpdf(::HipYNormal, y::AbstractVector{<:Real}, c::GPSLCHip.Ctx, U, X, uyLS, xyLS, tyLS::Float64,
                    yScale::Float64, yNoise::Float64)
    # y = the value Gen is scoring (Y_or_null); X = the trace's :X => k => :X values (X_or_null) or nothing
    GPSLCHip.y_logpdf(c, 1, _umat(U), X, GPSLCHip.f64(y), uyLS, xyLS, [tyLS], [yScale], [yNoise])[1]
end

function Gen.random(::HipYNormal, c::GPSLCHip.Ctx, U, X, uyLS, xyLS, tyLS::Float64, yScale::Float64, yNoise::Float64)
    # prior sampling only (generate without a constraint on :Y): F = [U | X | T] with the data set's treatment column
    Xh, Th, _ = c.data
    F, ls = _features(_umat(U) => uyLS, (X === nothing ? Xh : X) => xyLS, Th => tyLS)
    Gen.random(hip_gp_normal, F, ls, yScale, yNoise)
end
Gen.has_output_grad(::HipYNormal) = false
Gen.has_argument_grads(::HipYNormal) = ntuple(_ -> false, 8)               # c, U, X, uyLS, xyLS, tyLS, yScale, yNoise
Gen.logpdf_grad(::HipYNormal, y, args...) = ntuple(_ -> nothing, 9)        # output + 8 arguments
_register_continuous(HipYNormal)
