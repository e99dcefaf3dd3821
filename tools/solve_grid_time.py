"""`gss.solve` of an estimation problem on a large Cartesian grid: where the wall time goes (host-side centroids, transfers,
device work).  python3 tools/solve_grid_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch, gss
rng = np.random.default_rng(3)
n = 5000
xyz = rng.uniform(0, 216, (n, 3)); z = rng.normal(size=n)
data = gss.georef({"z": z}, xyz)
grid = gss.CartesianGrid(216, 216, 216)
vg = gss.MaternVariogram(range=30.0, order=1.5)
for name, solver in (("kriging k=16", gss.KrigingSolver(("z", dict(variogram=vg, maxneighbors=16)))),
                     ("idw k=16", gss.IDWSolver(("z", dict(maxneighbors=16))))):
    prob = gss.EstimationProblem(data, grid, "z")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sol = gss.solve(prob, solver)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    t2 = time.perf_counter(); c = grid.centroids(); t3 = time.perf_counter()
    print("%-14s solve %.3f s for %d cells (centroids on the host alone: %.3f s)" % (name, t1 - t0, grid.nelements(), t3 - t2))
