# The calls of examples/quickstart.py as a user of GeoStatsSolvers.jl writes them; with the shim loaded
# (geostatssolvers.jl_amd/julia/GeoStatsSolversHIP.jl, see INTEGRATION.md) `solve` runs on the MI355X through libgss_hip.so.
# Not executed in this repository's CI (no Julia toolchain in the image): the shim's ccall signatures are frozen against
# include/gss.h by tests/test_shim_signatures.py.
using GeoStatsBase, Meshes, Variography
include(joinpath(@__DIR__, "..", "geostatssolvers.jl_amd", "julia", "GeoStatsSolversHIP.jl"))
using .GeoStatsSolversHIP

data    = georef((z = [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.4, 0.3, 0.2, 0.1, 0.0],), collect(0.0:10.0:100.0)')
problem = EstimationProblem(data, CartesianGrid(100), :z)
γ       = GaussianVariogram(range = 35.0)
globalk = solve(problem, KrigingSolver(:z => (variogram = γ,)))
nearest = solve(problem, KrigingSolver(:z => (variogram = γ, maxneighbors = 3)))
localk  = solve(problem, KrigingSolver(:z => (variogram = γ, maxneighbors = 3, neighborhood = MetricBall(100.0))))

grid = CartesianGrid(100, 100)
ens  = solve(SimulationProblem(grid, :z => Float64, 3), FFTGS(:z => (variogram = GaussianVariogram(range = 10.0),)))
pts  = georef((z = [1.0, -1.0, 1.0],), [25.0 50.0 75.0; 25.0 75.0 50.0])
cens = solve(SimulationProblem(pts, grid, :z => Float64, 100), FFTGS(:z => (variogram = GaussianVariogram(range = 10.0),)))
lens = solve(SimulationProblem(pts, grid, :z => Float64, 2), LUGS(:z => (variogram = SphericalVariogram(range = 10.0),)))
sens = solve(SimulationProblem(pts, grid, :z => Float64, 2),
             SGS(:z => (variogram = SphericalVariogram(range = 35.0), neighborhood = MetricBall(30.0), maxneighbors = 10)))
# several GPUs: one worker process per GPU, `solve(problem, solver; procs = workers())` (INTEGRATION.md, "Multi-GPU")
