"""Wall time of the solver API (host arrays in / out) per solver, with the top host-side costs (cProfile)."""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/geostatssolvers.jl_amd")
import numpy as np
import torch
import gss

rng = np.random.default_rng(0)
xd = rng.uniform(0, 1000, (1000, 2))
zd = rng.normal(size=1000)
data = gss.georef({"z": zd}, xd)
grid = gss.CartesianGrid((1000, 1000))
cases = {
    "krig_global": (gss.EstimationProblem(data, grid, "z"), gss.KrigingSolver(("z", dict(variogram=gss.MaternVariogram(range=300.0, order=1.5))))),
    "krig_knn16": (gss.EstimationProblem(data, grid, "z"), gss.KrigingSolver(("z", dict(variogram=gss.MaternVariogram(range=300.0, order=1.5), maxneighbors=16)))),
    "idw16": (gss.EstimationProblem(data, grid, "z"), gss.IDWSolver(("z", dict(maxneighbors=16)))),
    "lwr16": (gss.EstimationProblem(data, grid, "z"), gss.LWRSolver(("z", dict(maxneighbors=16)))),
    "sgs": (gss.SimulationProblem(data, gss.CartesianGrid((256, 256)), "z", 64), gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=35.0), maxneighbors=16)), rng=1)),
    "lugs": (gss.SimulationProblem(gss.georef({"z": zd[:200]}, rng.uniform(0, 64, (200, 2))), gss.CartesianGrid((64, 64)), "z", 100), gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=20.0))), rng=1)),
    "fftgs": (gss.SimulationProblem(gss.CartesianGrid((256, 256, 64)), {"z": float}, 8), gss.FFTGS(("z", dict(variogram=gss.ExponentialVariogram(range=20.0))), rng=1)),
}
only = sys.argv[1:] or list(cases)
for name in only:
    prob, solver = cases[name]
    gss.solve(prob, solver); torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); gss.solve(prob, solver); torch.cuda.synchronize(); pr.disable()
    dt = time.perf_counter() - t0
    print("==== %s: %.1f ms" % (name, dt * 1e3), flush=True)
    st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(6)
