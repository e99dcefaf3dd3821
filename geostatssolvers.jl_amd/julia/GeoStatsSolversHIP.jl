# GeoStatsSolversHIP.jl -- Julia host shim over libgss_hip.so (include/gss.h).
#
# Drop-in for the KrigingSolver / FFTGS / LUGS methods of juliohm/GeoStatsSolvers.jl v0.7.16:
# the solver types keep the reference's parameter surface (src/estimation/krig.jl:64-74,
# src/simulation/fft.jl:51-60, src/simulation/lu.jl:67-74) and extend the same GeoStatsBase
# generics (`solve`, `preprocess`, `solvesingle`, src/GeoStatsSolvers.jl:28); the arithmetic the
# reference delegates to Variography / GeoStatsModels / FFTW / LinearAlgebra is replaced by `ccall`s.
#
# NOTE: there is no Julia toolchain in the build environment, so this file has never been executed;
# the executable twin with identical logic is geostatssolvers.jl_amd/gss/solvers.py, which the test
# suite drives through the same C-ABI.  Keep the two in step.
module GeoStatsSolversHIP

using Meshes
using GeoTables
using Variography
using GeoStatsBase
using Tables
using Random

import GeoStatsBase: solve, preprocess, solvesingle

export KrigingSolverHIP, IDWSolverHIP, LWRSolverHIP, ExpWeight, TricubeWeight, FFTGSHIP, LUGSHIP, SGSHIP

const libgss = get(ENV, "LIBGSS_HIP", "libgss_hip.so")

const GSS_MEM_HOST = Int32(0)
const GSS_KRIG_NO_FACTOR = Int32(1)

# ---- C structs ---------------------------------------------------------------------------
struct GssVgExtra              # one additional nested structure of gss_variogram_t
  kind::Int32
  aniso::Int32
  sill::Float64
  range::Float64
  nu::Float64
  inv_radii::NTuple{3,Float64}
end

struct GssVariogram            # gss_variogram_t
  kind::Int32
  dim::Int32
  sill::Float64
  nugget::Float64
  range::Float64
  nu::Float64
  aniso::Int32
  reserved::Int32
  inv_radii::NTuple{3,Float64}
  nextra::Int32
  reserved2::Int32
  extra::NTuple{3,GssVgExtra}
end

const NOEXTRA = GssVgExtra(Int32(0), Int32(0), 0.0, 1.0, 1.0, (1.0, 1.0, 1.0))

function check(code::Int32)
  code == 0 && return nothing
  buf = Vector{UInt8}(undef, 512)
  ccall((:gss_last_error, libgss), Int32, (Ptr{UInt8}, Int32), buf, 512)
  msg = unsafe_string(pointer(buf))
  code == 1 && throw(ArgumentError(msg))          # GSS_ERR_INVALID (krig.jl:100-102, fft.jl:91-93)
  code == 3 && throw(ErrorException("not positive definite: " * msg))
  error("libgss_hip error $code: $msg")
end

vgkind(::GaussianVariogram) = Int32(0)
vgkind(::ExponentialVariogram) = Int32(1)
vgkind(::SphericalVariogram) = Int32(2)
vgkind(::MaternVariogram) = Int32(3)
vgkind(::CubicVariogram) = Int32(4)
vgkind(::PentasphericalVariogram) = Int32(5)
vgkind(::SineHoleVariogram) = Int32(6)

function structure(γ)            # (kind, aniso, range, nu, inv_radii) of one basic model
  rs = radii(metricball(γ))
  aniso = length(rs) > 1
  ir = ntuple(i -> aniso && i <= length(rs) ? 1.0 / ustrip(rs[i]) : 1.0, 3)
  ν = γ isa MaternVariogram ? Float64(γ.order) : 1.0
  (vgkind(γ), Int32(aniso), aniso ? 1.0 : Float64(ustrip(range(γ))), ν, ir)
end

# `extent` = diameter of the data bounding box: only needed for the non-stationary PowerVariogram, which the
# device handles through the pseudo-covariance A - gamma(h) with A = 2 gamma(extent) (gss.h, GSS_VG_POWER);
# the kriging solver passes it, the simulation solvers do not and therefore keep the reference's assertion.
function cvariogram(γ, dim; extent=nothing)
  if γ isa PowerVariogram && !isnothing(extent)
    A = 2 * Float64(γ.scaling) * Float64(extent)^Float64(γ.exponent) + Float64(nugget(γ)) + floatmin(Float64)
    return GssVariogram(Int32(7), Int32(dim), A, Float64(nugget(γ)), Float64(γ.scaling), Float64(γ.exponent), Int32(0),
                        Int32(0), (1.0, 1.0, 1.0), Int32(0), Int32(0), (NOEXTRA, NOEXTRA, NOEXTRA))
  end
  isstationary(γ) || throw(ArgumentError("variogram model must be stationary"))   # fft.jl:91-93, lu.jl:110
  if γ isa NestedVariogram       # gamma = sum c_i gamma_i: first structure carries the total nugget
    cs, γs = γ.cs, γ.γs
    keep = [i for i in eachindex(γs) if cs[i] * (sill(γs[i]) - nugget(γs[i])) > 0]
    length(keep) <= 4 || throw(ArgumentError("at most 4 nested structures are supported on the device"))
    k0, a0, r0, ν0, ir0 = structure(γs[keep[1]])
    nug = Float64(nugget(γ))
    extras = ntuple(3) do j
      j + 1 > length(keep) && return NOEXTRA
      i = keep[j+1]
      k, a, r, ν, ir = structure(γs[i])
      GssVgExtra(k, a, Float64(cs[i] * (sill(γs[i]) - nugget(γs[i]))), r, ν, ir)
    end
    c0 = Float64(cs[keep[1]] * (sill(γs[keep[1]]) - nugget(γs[keep[1]])))
    return GssVariogram(k0, Int32(dim), c0 + nug, nug, r0, ν0, a0, Int32(0), ir0, Int32(length(keep) - 1), Int32(0), extras)
  end
  k, a, r, ν, ir = structure(γ)
  GssVariogram(k, Int32(dim), Float64(sill(γ)), Float64(nugget(γ)), r, ν, a, Int32(0), ir, Int32(0), Int32(0),
               (NOEXTRA, NOEXTRA, NOEXTRA))
end

# solver parameter `distance` (krig.jl:72, idw.jl:54, lwr.jl:57) -> (GSS_METRIC_*, parameter); a neighbourhood
# overrides it exactly as searcher_ui does (ui.jl:25-31)
metricspec(::Euclidean) = (Int32(0), 0.0)
metricspec(::Cityblock) = (Int32(1), 0.0)
metricspec(::Chebyshev) = (Int32(2), 0.0)
metricspec(d::Haversine) = (Int32(3), Float64(d.radius))
metricspec(d) = throw(ArgumentError("search distance $d is not available on the device"))
searchmetric(p) = isnothing(p.neighborhood) ? metricspec(p.distance) : (Int32(0), 0.0)

# point-major coordinates: a d x n Julia matrix is already in the layout the C-ABI wants
coordmatrix(dom) = reduce(hcat, [collect(Float64, ustrip.(coordinates(centroid(dom, i)))) for i in 1:nelements(dom)])

# ---- KrigingSolver ------------------------------------------------------------------------
@estimsolver KrigingSolverHIP begin
  @param variogram = GaussianVariogram()
  @param mean = nothing
  @param degree = nothing
  @param drifts = nothing
  @param minneighbors = 1
  @param maxneighbors = nothing
  @param neighborhood = nothing
  @param distance = Euclidean()
  @param path = LinearPath()
end

function solve(problem::EstimationProblem, solver::KrigingSolverHIP)
  pdata = data(problem)
  pdomain = domain(problem)
  dtable = values(pdata)
  ddomain = domain(pdata)
  X0 = coordmatrix(pdomain)
  d, m = size(X0)
  μs, σs = [], []
  for covars in covariables(problem, solver), var in covars.names
    p = covars.params[Set([var])]
    zcol = Tables.getcolumn(Tables.columns(dtable), var)
    inds = findall(!ismissing, zcol)                                     # krig.jl:97
    isempty(inds) && throw(AssertionError("all samples of $var are missing, aborting..."))
    z = Float64.(collect(skipmissing(zcol)))
    X = coordmatrix(view(ddomain, inds))
    n = length(z)
    # kriging_ui, ui.jl:40-50
    variant, degree, ndrift, Fd, F0 = Int32(1), Int32(0), Int32(0), C_NULL, C_NULL
    skmean = 0.0
    if !isnothing(p.drifts)
      variant, ndrift = Int32(3), Int32(length(p.drifts))
      Fd = Float64[f(Point(X[:, i]...)) for f in p.drifts, i in 1:n]     # ndrift x n == n x ndrift row-major
      F0 = Float64[f(Point(X0[:, i]...)) for f in p.drifts, i in 1:m]
    elseif !isnothing(p.degree)
      variant, degree = Int32(2), Int32(p.degree)
    elseif !isnothing(p.mean)
      variant, skmean = Int32(0), Float64(p.mean)
    end
    # searcher_ui, ui.jl:11-32
    exact = isnothing(p.maxneighbors)
    k = exact ? n : p.maxneighbors
    if !exact && (k < 1 || k > n)
      @warn "Invalid maximum number of neighbors. Adjusting to $n..."
      k = n
    end
    extent = sqrt(sum(abs2, maximum(X, dims=2) .- minimum(X, dims=2)))
    vg = Ref(cvariogram(p.variogram, d; extent))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    μ = Vector{Float64}(undef, m); σ² = similar(μ); status = Vector{UInt8}(undef, m)
    GC.@preserve X z X0 Fd F0 μ σ² status begin
      check(ccall((:gss_krig_create, libgss), Int32,
                  (Ptr{Ptr{Cvoid}}, Ptr{GssVariogram}, Int32, Float64, Int32, Int32, Ptr{Float64}, Ptr{Float64},
                   Ptr{Float64}, Int64, Int32, Ptr{Cvoid}),
                  h, vg, variant, skmean, degree, ndrift, X, z, Fd, n, exact ? Int32(0) : GSS_KRIG_NO_FACTOR, C_NULL))
      try
        if exact                                                         # krig.jl:166-186
          check(ccall((:gss_krig_predict_global, libgss), Int32,
                      (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Int32,
                       Ptr{Cvoid}), h[], X0, F0, m, μ, σ², status, GSS_MEM_HOST, C_NULL))
        else                                                             # krig.jl:188-234
          radius, ir = -1.0, C_NULL
          if !isnothing(p.neighborhood)
            rs = ustrip.(radii(p.neighborhood))
            length(rs) == 1 ? (radius = Float64(rs[1])) : (radius = 1.0; ir = Float64[1 / r for r in rs])
          end
          met, mpar = searchmetric(p)
          check(ccall((:gss_krig_predict_knn, libgss), Int32,
                      (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int32, Int32, Float64, Ptr{Float64}, Int32,
                       Float64, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Ptr{Int32}, Ptr{Int32}, Int32, Ptr{Cvoid}),
                      h[], X0, F0, m, Int32(k), Int32(p.minneighbors), radius, ir, met, mpar, μ, σ², status, C_NULL,
                      C_NULL, GSS_MEM_HOST, C_NULL))
        end
      finally
        ccall((:gss_krig_destroy, libgss), Int32, (Ptr{Cvoid},), h[])
      end
    end
    miss = status .!= 0                                                  # krig.jl:213-214
    inds = collect(traverse(pdomain, p.path))                            # results in traversal order, krig.jl:179-183
    push!(μs, var => [miss[i] ? missing : μ[i] for i in inds])
    push!(σs, Symbol(var, "_variance") => [miss[i] ? missing : σ²[i] for i in inds])
  end
  georef((; μs..., σs...), pdomain)                                      # krig.jl:163
end

# ---- IDWSolver / LWRSolver (idw.jl:49-153, lwr.jl:53-158) -------------------------------------
struct ExpWeight                 # h -> exp(-a h^p); ExpWeight(3, 2) is the reference default (lwr.jl:58)
  a::Float64
  p::Float64
end
struct TricubeWeight end         # h -> (1 - h^3)^3
(w::ExpWeight)(h) = exp(-w.a * h^w.p)
(::TricubeWeight)(h) = (1 - h^3)^3
weightspec(w::ExpWeight) = (Int32(0), w.a, w.p)
weightspec(::TricubeWeight) = (Int32(1), 0.0, 0.0)
weightspec(f) = throw(ArgumentError("weightfun must be ExpWeight(a, p) or TricubeWeight(): weights are evaluated on the device"))

@estimsolver IDWSolverHIP begin
  @param minneighbors = 1
  @param maxneighbors = nothing
  @param neighborhood = nothing
  @param distance = Euclidean()
  @param exponent = 1
  @param path = LinearPath()
end

@estimsolver LWRSolverHIP begin
  @param minneighbors = 1
  @param maxneighbors = nothing
  @param neighborhood = nothing
  @param distance = Euclidean()
  @param weightfun = ExpWeight(3.0, 2.0)
  @param path = LinearPath()
end

function neighbor_estimate(problem, solver, auxname, call)
  pdata = data(problem)
  pdomain = domain(problem)
  dtable = values(pdata)
  X0 = coordmatrix(pdomain)
  d, m = size(X0)
  μs, σs = [], []
  for covars in covariables(problem, solver), var in covars.names
    p = covars.params[Set([var])]
    zcol = Tables.getcolumn(Tables.columns(dtable), var)
    inds = findall(!ismissing, zcol)                                     # idw.jl:77, lwr.jl:80
    n = length(inds)
    @assert n > 0 "estimation requires data"
    z = Float64.(ustrip.(collect(skipmissing(zcol))))                    # uadjust, idw.jl:109
    X = coordmatrix(view(domain(pdata), inds))
    nmax = isnothing(p.maxneighbors) ? n : min(p.maxneighbors, n)        # idw.jl:93
    @assert p.minneighbors ≤ nmax "invalid min/max number of neighbors"
    k = nmax
    if !isnothing(p.maxneighbors) && (p.maxneighbors < 1 || p.maxneighbors > n)   # searcher_ui, ui.jl:18-20
      @warn "Invalid maximum number of neighbors. Adjusting to $n..."
      k = n
    end
    radius, ir = -1.0, C_NULL
    if !isnothing(p.neighborhood)
      rs = ustrip.(radii(p.neighborhood))
      length(rs) == 1 ? (radius = Float64(rs[1])) : (radius = 1.0; ir = Float64[1 / r for r in rs])
    end
    μ = Vector{Float64}(undef, m); aux = similar(μ); status = Vector{UInt8}(undef, m)
    GC.@preserve X z X0 ir μ aux status check(call(p, X, z, n, d, X0, m, Int32(k), radius, ir, μ, aux, status))
    miss = status .!= 0                                                  # idw.jl:123-124
    inds = collect(traverse(pdomain, p.path))                            # results in traversal order, idw.jl:112-113
    push!(μs, var => [miss[i] ? missing : μ[i] for i in inds])
    push!(σs, Symbol(var, auxname) => [miss[i] ? missing : aux[i] for i in inds])
  end
  georef((; μs..., σs...), pdomain)
end

solve(problem::EstimationProblem, solver::IDWSolverHIP) =
  neighbor_estimate(problem, solver, "_distance", (p, X, z, n, d, X0, m, k, radius, ir, μ, aux, status) -> begin
    @assert p.exponent > 0 "exponent must be positive"                   # idw.jl:96
    met, mpar = searchmetric(p)
    ccall((:gss_idw_predict, libgss), Int32,
          (Ptr{Float64}, Ptr{Float64}, Int64, Int32, Ptr{Float64}, Int64, Int32, Int32, Float64, Ptr{Float64},
           Int32, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Int32, Ptr{Cvoid}),
          X, z, n, Int32(d), X0, m, k, Int32(p.minneighbors), radius, ir, met, mpar, Float64(p.exponent), μ, aux,
          status, GSS_MEM_HOST, C_NULL)
  end)

solve(problem::EstimationProblem, solver::LWRSolverHIP) =
  neighbor_estimate(problem, solver, "_variance", (p, X, z, n, d, X0, m, k, radius, ir, μ, aux, status) -> begin
    wk, wa, wp = weightspec(p.weightfun)
    met, mpar = searchmetric(p)
    ccall((:gss_lwr_predict, libgss), Int32,
          (Ptr{Float64}, Ptr{Float64}, Int64, Int32, Ptr{Float64}, Int64, Int32, Int32, Float64, Ptr{Float64},
           Int32, Float64, Int32, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Int32, Ptr{Cvoid}),
          X, z, n, Int32(d), X0, m, k, Int32(p.minneighbors), radius, ir, met, mpar, wk, wa, wp, μ, aux, status,
          GSS_MEM_HOST, C_NULL)
  end)

# ---- FFTGS ----------------------------------------------------------------------------------
@simsolver FFTGSHIP begin
  @param variogram = GaussianVariogram()
  @param mean = 0.0
  @param minneighbors = 1
  @param maxneighbors = nothing
  @param neighborhood = nothing
  @param distance = Euclidean()
  @global seed = rand(UInt64)
end

function preprocess(problem::SimulationProblem, solver::FFTGSHIP)
  pgrid = parent(domain(problem))
  dims = Int64[size(pgrid)...]
  sp = Float64[ustrip.(spacing(pgrid))...]
  preproc = Dict()
  for covars in covariables(problem, solver), var in covars.names
    p = covars.params[Set([var])]
    vg = Ref(cvariogram(p.variogram, length(dims)))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:gss_fftgs_create, libgss), Int32,
                (Ptr{Ptr{Cvoid}}, Ptr{GssVariogram}, Int32, Ptr{Int64}, Ptr{Float64}, Float64, Int32, Ptr{Cvoid}),
                h, vg, Int32(length(dims)), dims, sp, Float64(p.mean), Int32(0), C_NULL))
    preproc[var] = (handle=h[], μ=p.mean, γ=p.variogram)                # conditional extras: see solvers.py FFTGS
  end
  preproc
end

function solvesingle(problem::SimulationProblem, covars::NamedTuple, solver::FFTGSHIP, preproc; real::Int=0)
  pdomain = domain(problem)
  inds = parentindices(pdomain)
  npts = nelements(pdomain)
  varreal = map(collect(covars.names)) do var
    h = preproc[var].handle
    out = Vector{Float64}(undef, npts)
    ii = Int64.(collect(inds) .- 1)                                      # 0-based on the C side
    GC.@preserve out ii check(ccall((:gss_fftgs_realize, libgss), Int32,
      (Ptr{Cvoid}, UInt64, Int64, Int64, Ptr{Float64}, Ptr{Int64}, Int64, Ptr{Float64}, Int32, Ptr{Cvoid}),
      h, solver.seed, Int64(real), Int64(1), C_NULL, ii, Int64(npts), out, GSS_MEM_HOST, C_NULL))
    var => out
  end
  Dict(varreal)
end

# ---- LUGS -------------------------------------------------------------------------------------
@simsolver LUGSHIP begin
  @param variogram = GaussianVariogram()
  @param mean = nothing
  @param factorization = cholesky          # only cholesky is implemented on the device
  @jparam correlation = 0.0
  @global init = NearestInit()
  @global seed = rand(UInt64)
end

function preprocess(problem::SimulationProblem, solver::LUGSHIP)
  pdomain = domain(problem)
  buff, mask = initbuff(pdomain, variables(problem), solver.init, data=data(problem))   # lu.jl:86
  C = coordmatrix(pdomain)
  d, N = size(C)
  preproc = Dict()
  for covars in covariables(problem, solver)
    conames = covars.names
    @assert length(conames) ∈ (1, 2) "invalid number of covariables"                    # lu.jl:96
    coparams = Dict()
    for var in conames
      p = covars.params[Set([var])]
      dlocs = Int64.(findall(mask[var]) .- 1)                                           # lu.jl:113
      z₁ = Float64.(buff[var][findall(mask[var])])
      !isnothing(p.mean) && !isempty(dlocs) && @warn "mean can only be specified in unconditional simulation"
      μ = isnothing(p.mean) ? 0.0 : Float64(p.mean)
      vg = Ref(cvariogram(p.variogram, d))
      h = Ref{Ptr{Cvoid}}(C_NULL)
      GC.@preserve C dlocs z₁ check(ccall((:gss_lugs_create, libgss), Int32,
        (Ptr{Ptr{Cvoid}}, Ptr{GssVariogram}, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Float64, Int32,
         Ptr{Cvoid}), h, vg, C, N, dlocs, z₁, length(dlocs), μ, Int32(0), C_NULL))
      coparams[Set([var])] = (handle=h[], N=N, ns=N - length(dlocs))
    end
    length(conames) == 2 && (coparams[conames] = covars.params[conames].correlation)    # lu.jl:154-163
    push!(preproc, conames => coparams)
  end
  preproc
end

function lusim_hip(par, seed, real, ρ=nothing, w₁=nothing)                               # lu.jl:198-224
  y = Vector{Float64}(undef, par.N)
  w₂ = Vector{Float64}(undef, par.ns)
  GC.@preserve y w₂ w₁ check(ccall((:gss_lugs_realize, libgss), Int32,
    (Ptr{Cvoid}, UInt64, Int64, Int64, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32,
     Ptr{Cvoid}), par.handle, seed, Int64(real), Int64(1), C_NULL, isnothing(ρ) ? 0.0 : Float64(ρ),
    isnothing(w₁) ? C_NULL : pointer(w₁), y, w₂, GSS_MEM_HOST, C_NULL))
  y, w₂
end

function solvesingle(::SimulationProblem, covars::NamedTuple, solver::LUGSHIP, preproc; real::Int=0)
  conames = covars.names
  coparams = preproc[conames]
  vars = collect(conames)
  v₁ = first(vars)
  Y₁, w₁ = lusim_hip(coparams[Set([v₁])], solver.seed, real)
  result = Dict(v₁ => Y₁)
  if length(conames) == 2
    v₂ = last(vars)
    Y₂, _ = lusim_hip(coparams[Set([v₂])], solver.seed + 1, real, coparams[conames], w₁)
    push!(result, v₂ => Y₂)
  end
  result
end

# ---- SGS (sgs.jl:45-89 over seq.jl:42-141) -----------------------------------------------------
@simsolver SGSHIP begin
  @param variogram = GaussianVariogram()
  @param mean = 0.0
  @param path = LinearPath()
  @param minneighbors = 1
  @param maxneighbors = 10
  @param neighborhood = nothing
  @param distance = Euclidean()
  @global init = NearestInit()
  @global seed = rand(UInt64)
end

function preprocess(problem::SimulationProblem, solver::SGSHIP)
  pdomain = domain(problem)
  buff, mask = initbuff(pdomain, variables(problem), solver.init, data=data(problem))   # seq.jl:85
  C = coordmatrix(pdomain)
  d, N = size(C)
  preproc = Dict()
  for covars in covariables(problem, solver), var in covars.names
    p = covars.params[Set([var])]
    path = Int64.(collect(traverse(pdomain, p.path)) .- 1)               # one visiting order for all realisations
    dlocs = Int64.(findall(mask[var]) .- 1)
    zdata = Float64.(buff[var][mask[var]])
    k = p.maxneighbors
    if k < 1 || k > N                                                     # searcher_ui, ui.jl:18-20
      @warn "Invalid maximum number of neighbors. Adjusting to $N..."
      k = N
    end
    radius, ir = -1.0, C_NULL
    if !isnothing(p.neighborhood)
      rs = ustrip.(radii(p.neighborhood))
      length(rs) == 1 ? (radius = Float64(rs[1])) : (radius = 1.0; ir = Float64[1 / r for r in rs])
    end
    vg = Ref(cvariogram(p.variogram, d))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve C path dlocs zdata ir check(ccall((:gss_sgs_create, libgss), Int32,
      (Ptr{Ptr{Cvoid}}, Ptr{GssVariogram}, Float64, Ptr{Float64}, Int64, Int32, Ptr{Int64}, Ptr{Int64}, Ptr{Float64},
       Int64, Int32, Int32, Float64, Ptr{Float64}, Int32, Ptr{Cvoid}),
      h, vg, Float64(p.mean), C, N, Int32(d), path, dlocs, zdata, length(dlocs), Int32(k), Int32(p.minneighbors),
      radius, ir, Int32(0), C_NULL))
    preproc[var] = (handle=h[], N=N)
  end
  preproc
end

function solvesingle(::SimulationProblem, covars::NamedTuple, solver::SGSHIP, preproc; real::Int=0)
  varreal = map(collect(covars.names)) do var
    h, N = preproc[var]
    out = Vector{Float64}(undef, N)
    GC.@preserve out check(ccall((:gss_sgs_realize, libgss), Int32,
      (Ptr{Cvoid}, UInt64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Cvoid}),
      h, solver.seed, Int64(real), Int64(1), C_NULL, out, GSS_MEM_HOST, C_NULL))
    var => out
  end
  Dict(varreal)
end

end # module
