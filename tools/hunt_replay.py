"""Replays one case of tools/hunt_large_k.py and prints per-point differences.  python3 tools/hunt_replay.py seed case"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np
import gss
from gss.engine import KrigHandle
from oracle import kriging as K
from oracle.variogram import Variogram
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(target + 1):
    dim = int(rng.integers(1, 4))
    k = int(rng.choice([rng.integers(65, 130), rng.integers(130, 257), rng.integers(257, 769), rng.integers(769, 900)]))
    n = int(k + rng.integers(1, 400))
    m = int(rng.integers(1, 12))
    variant, okw = [(K.SK, dict(mean=0.3)), (K.OK, {}), (K.UK, dict(degree=1)), (K.UK, dict(degree=2))][int(rng.integers(0, 4))]
    kind = ["exponential", "spherical", "matern", "gaussian"][int(rng.integers(0, 4))]
    nug = float(rng.choice([0.0, 0.02, 0.3])) if kind != "gaussian" else 0.05
    aniso = dim > 1 and rng.random() < 0.3
    radii = tuple(float(v) for v in rng.uniform(15, 60, dim)) if aniso else None
    vkw = dict(nugget=nug, sill=float(rng.uniform(0.5, 3.0)))
    if kind == "matern":
        vkw["nu"] = float(rng.choice([0.5, 1.5, 2.5, 1.0, 0.8]))
    if radii is None:
        vkw["range"] = float(rng.uniform(10, 60))
    x = rng.uniform(0, 100, (n, dim))
    if dim == 1:
        x = np.sort(x, axis=0) + np.arange(n)[:, None] * 1e-3
    dup = False
    if rng.random() < 0.3 and nug > 0:
        x[1] = x[0]; dup = True
    z = rng.normal(size=n) + 0.01 * x[:, 0]
    x0 = rng.uniform(0, 100, (m, dim))
    x0[0] = x[int(rng.integers(0, n))]
    ball = float(rng.uniform(30, 120)) if rng.random() < 0.4 else None
print("dim", dim, "k", k, "n", n, "m", m, "variant", variant, okw, kind, vkw, "radii", radii, "ball", ball, "dup", dup)
ctor = dict(exponential=gss.ExponentialVariogram, spherical=gss.SphericalVariogram, matern=gss.MaternVariogram, gaussian=gss.GaussianVariogram)[kind]
gkw = dict(vkw)
if "nu" in gkw: gkw["order"] = gkw.pop("nu")
gvg = ctor(gss.MetricBall(radii), **gkw) if radii else ctor(**gkw)
ovg = Variogram(kind, radii=radii, **vkw) if radii else Variogram(kind, **vkw)
h = KrigHandle(gvg, variant, x, z, mean=okw.get("mean"), degree=okw.get("degree"), factor=False)
mu, var, st, idx, cnt = h.predict_knn(x0, k, minneighbors=3, radius=ball, return_idx=True)
h.close()
rmu, rvar, rst, ridx, rcnt = K.approxsolve(variant, ovg, x, z, x0, k, 3, mean=okw.get("mean") or 0.0, degree=okw.get("degree"), radius=ball, return_idx=True)
print("cnt", cnt, "st", st, rst)
print("mu diff", np.abs(mu - rmu)); print("var diff", np.abs(var - rvar)); print("rmu", rmu); print("rvar", rvar)
# condition number of the first point's system
i = 0
nb = ridx[i][:rcnt[i]]
from oracle.variogram import cov_pairwise
C = cov_pairwise(ovg, x[nb])
print("cond(C) first point %.3e" % np.linalg.cond(C), "neighbours", len(nb))
