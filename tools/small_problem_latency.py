"""Wall time of the solver front-end on problems of the size of the reference's own tests (configs[0] and the
100 x 100 simulation grids of test/simulation/*.jl): first call of the process and the calls after it.
python3 tools/small_problem_latency.py (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
import gss

def timed(label, fn, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); first = time.perf_counter() - t0
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-62s first %8.1f ms   then %7.2f ms (min of %d)" % (label, 1e3 * first, 1e3 * min(ts), reps), flush=True)

rng = np.random.default_rng(1)
xy = rng.uniform(0, 64, (100, 2)); z = rng.normal(size=100)
grid = gss.CartesianGrid(64, 64)
prob = gss.EstimationProblem(gss.georef({"z": z}, xy), grid, "z")
timed("configs[0]: OK, 100 data -> 64 x 64 cells, Gaussian range 20", lambda: gss.solve(prob, gss.KrigingSolver(("z", dict(variogram=gss.GaussianVariogram(range=20.0, nugget=1e-6))))))
timed("  the same with maxneighbors = 16", lambda: gss.solve(prob, gss.KrigingSolver(("z", dict(variogram=gss.GaussianVariogram(range=20.0, nugget=1e-6), maxneighbors=16)))))
g100 = gss.CartesianGrid(100, 100)
sp = gss.SimulationProblem(g100, ("z", float), 3)
timed("FFTGS 100 x 100, Gaussian range 10, 3 realisations", lambda: gss.solve(sp, gss.FFTGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0))))))
timed("LUGS 100 x 100, Gaussian range 10, 3 realisations", lambda: gss.solve(sp, gss.LUGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0))))))
cond = gss.SimulationProblem(gss.georef({"z": z[:25]}, xy[:25] * 100 / 64), g100, "z", 3)
timed("LUGS 100 x 100 conditional on 25 data, spherical range 10", lambda: gss.solve(cond, gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0))))))
timed("SGS 100 x 100 conditional, 16 neighbours, 3 realisations", lambda: gss.solve(cond, gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0), maxneighbors=16, neighborhood=gss.MetricBall(20.0))))))
timed("IDW 100 data -> 64 x 64 cells", lambda: gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=8)))))
timed("FFTGS 100 x 100 conditional on 25 data (fft.jl:25-32), 3 realisations", lambda: gss.solve(cond, gss.FFTGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0), maxneighbors=10)))))
timed("FFTGS 100 x 100 conditional, global kriging of the residuals", lambda: gss.solve(cond, gss.FFTGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0))))))
timed("UK degree 1, 100 data -> 64 x 64 cells, spherical", lambda: gss.solve(prob, gss.KrigingSolver(("z", dict(variogram=gss.SphericalVariogram(range=20.0), degree=1)))))
timed("LWR 100 data -> 64 x 64 cells, 10 neighbours", lambda: gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=10)))))
big = gss.EstimationProblem(gss.georef({"z": rng.normal(size=2000)}, rng.uniform(0, 256, (2000, 2))), gss.CartesianGrid(256, 256), "z")
timed("OK, 2000 data -> 256 x 256 cells, 16 neighbours", lambda: gss.solve(big, gss.KrigingSolver(("z", dict(variogram=gss.SphericalVariogram(range=30.0), maxneighbors=16)))))
timed("OK, 2000 data -> 256 x 256 cells, global", lambda: gss.solve(big, gss.KrigingSolver(("z", dict(variogram=gss.SphericalVariogram(range=30.0))))))
