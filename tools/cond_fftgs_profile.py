import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/geostatssolvers.jl_amd")
import numpy as np, torch, gss
e, R = 128, 16
rng = np.random.default_rng(9)
grid = gss.CartesianGrid((e, e, e))
xd = rng.uniform(0.0, float(e), (1000, 3)); zd = rng.normal(size=1000)
prob = gss.SimulationProblem(gss.georef({"z": zd}, xd), grid, "z", R)
solver = gss.FFTGS(("z", dict(variogram=gss.ExponentialVariogram(range=20.0))), rng=3)
solver.solve(prob); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
solver.solve(prob); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
