#!/usr/bin/env python3
"""The reference's problem / solver / solve workflow on the MI355X through the Python twin (`gss`), one block per solver.
The inputs are the ones the reference's own test-suite uses (test/estimation/krig.jl:6-19, test/estimation/idw.jl,
test/simulation/fft.jl:3-32, test/simulation/lu.jl, test/simulation/seq.jl); the Julia side of the same calls is in
examples/quickstart.jl.  python examples/quickstart.py   (needs the built library and an MI355X)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np  # noqa: E402

import gss  # noqa: E402

out = {}

# ---- estimation: KrigingSolver (global neighbourhood, k nearest, k nearest inside a ball) --------------------------
x = np.arange(0.0, 101.0, 10.0)[:, None]
z = np.array([0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.4, 0.3, 0.2, 0.1, 0.0])
data = gss.georef({"z": z}, x)
problem = gss.EstimationProblem(data, gss.CartesianGrid(100), "z")
vg = gss.GaussianVariogram(range=35.0)
for name, params in (("global", dict(variogram=vg)),
                     ("nearest", dict(variogram=vg, maxneighbors=3)),
                     ("local", dict(variogram=vg, maxneighbors=3, neighborhood=gss.MetricBall(100.0)))):
    sol = gss.solve(problem, gss.KrigingSolver(("z", params)))
    out["kriging_" + name] = (sol["z"], sol["z_variance"])
    print("kriging %-8s mean[0:3] = %s   variance[0:3] = %s" % (name, np.round(sol["z"][:3], 4), np.round(sol["z_variance"][:3], 4)))

# universal kriging with a drift of degree 1 and a simple-kriging mean are solver parameters as in the reference
sol = gss.solve(problem, gss.KrigingSolver(("z", dict(variogram=vg, degree=1))))
out["kriging_uk"] = (sol["z"], sol["z_variance"])

# ---- estimation: IDW and LWR ---------------------------------------------------------------------------------------
pts = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
d2 = gss.georef({"z": [1.0, 0.0, 1.0]}, pts)
p2 = gss.EstimationProblem(d2, gss.CartesianGrid(100, 100), "z")
out["idw"] = gss.solve(p2, gss.IDWSolver())["z"]
out["lwr"] = gss.solve(gss.EstimationProblem(gss.georef({"z": z}, x), gss.CartesianGrid(100), "z"),
                       gss.LWRSolver(("z", dict(maxneighbors=8))))["z"]
print("idw at the three data cells:", np.round(out["idw"].reshape(100, 100)[[25, 75, 50], [25, 50, 75]], 3))

# ---- simulation: FFTGS, unconditional and conditional --------------------------------------------------------------
grid = gss.CartesianGrid(100, 100)
sim = gss.SimulationProblem(grid, ("z", float), 3)
ens = gss.solve(sim, gss.FFTGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0))), rng=2019))
out["fftgs"] = np.stack(ens["z"])
print("FFTGS: %d realisations of %d cells, sample variance %.3f" % (len(ens["z"]), ens[0].z.size, out["fftgs"].var()))
cond = gss.SimulationProblem(gss.georef({"z": [1.0, -1.0, 1.0]}, pts), grid, ("z", float), 100)
cens = gss.solve(cond, gss.FFTGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0))), rng=2022))
out["fftgs_cond"] = np.stack(cens["z"])

# ---- simulation: LUGS (conditional, two correlated variables) and SGS ---------------------------------------------
g1 = gss.CartesianGrid(100)
ldat = gss.georef({"z": [0.0, 1.0, 0.0, 1.0, 0.0]}, np.array([0.0, 25.0, 50.0, 75.0, 100.0])[:, None])
lens = gss.solve(gss.SimulationProblem(ldat, g1, ("z", float), 2),
                 gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0))), rng=123))
out["lugs"] = np.stack(lens["z"])
co = gss.solve(gss.SimulationProblem(gss.CartesianGrid(500), {"z": float, "y": float}, 1),
               gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0))),
                        ("y", dict(variogram=gss.GaussianVariogram(range=10.0))),
                        (("z", "y"), dict(correlation=0.95)), rng=123))
out["lugs_corr"] = float(np.corrcoef(co["z"][0], co["y"][0])[0, 1])
print("LUGS co-simulation (noise correlation 0.95, two different models): correlation of the two fields %.3f" % out["lugs_corr"])
sens = gss.solve(gss.SimulationProblem(d2, grid, ("z", float), 2),
                 gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=35.0), neighborhood=gss.MetricBall(30.0),
                                    maxneighbors=10)), rng=2017))
out["sgs"] = np.stack(sens["z"])
cell = int(np.argmin(((grid.centroids() - pts[0]) ** 2).sum(axis=1)))      # the cell that received the first datum
print("SGS: %d realisations, value at the cell of the datum z = 1: %s" % (len(sens["z"]), out["sgs"][:, cell]))
