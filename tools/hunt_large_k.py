#!/usr/bin/env python3
"""Random hunt over the moving-neighbourhood kernels beyond 64 neighbours (register tiles 65..256, slab 257..768, scalar
beyond): random k, dimension, variant, model, nugget, anisotropy, ball, duplicates -- device against the oracle.
python3 tools/hunt_large_k.py [seed] [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np  # noqa: E402

import gss  # noqa: E402
from gss.engine import KrigHandle  # noqa: E402
from oracle import kriging as K  # noqa: E402
from oracle.variogram import Variogram, cov_pairwise  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
worst = 0.0
for it in range(cases):
    dim = int(rng.integers(1, 4))
    k = int(rng.choice([rng.integers(65, 130), rng.integers(130, 257), rng.integers(257, 769), rng.integers(769, 900)]))
    n = int(k + rng.integers(1, 400))
    m = int(rng.integers(1, 12))
    variant, okw = [(K.SK, dict(mean=0.3)), (K.OK, {}), (K.UK, dict(degree=1)), (K.UK, dict(degree=2))][int(rng.integers(0, 4))]
    kind = ["exponential", "spherical", "matern"][int(rng.integers(0, 3))]
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
    # (no coincident samples: two samples at distance zero have covariance C(0) = sill with each other, the system is
    #  exactly singular -- the variance is still unique, the mean is not -- and the Gaussian model is left to the k <= 64
    #  tests: a hundred neighbours under it are singular to working precision)
    z = rng.normal(size=n) + 0.01 * x[:, 0]
    x0 = rng.uniform(0, 100, (m, dim))
    x0[0] = x[int(rng.integers(0, n))]
    ball = float(rng.uniform(30, 120)) if rng.random() < 0.4 else None
    ctor = dict(exponential=gss.ExponentialVariogram, spherical=gss.SphericalVariogram, matern=gss.MaternVariogram,
                gaussian=gss.GaussianVariogram)[kind]
    gkw = dict(vkw)
    if "nu" in gkw:
        gkw["order"] = gkw.pop("nu")
    gvg = ctor(gss.MetricBall(radii), **gkw) if radii else ctor(**gkw)
    ovg = Variogram(kind, radii=radii, **vkw) if radii else Variogram(kind, **vkw)
    h = KrigHandle(gvg, variant, x, z, mean=okw.get("mean"), degree=okw.get("degree"), factor=False)
    mu, var, st, idx, cnt = h.predict_knn(x0, k, minneighbors=3, radius=ball, return_idx=True)
    h.close()
    rmu, rvar, rst, ridx, rcnt = K.approxsolve(variant, ovg, x, z, x0, k, 3, mean=okw.get("mean") or 0.0,
                                               degree=okw.get("degree"), radius=ball, return_idx=True)
    tag = "case %d: dim %d k %d n %d %s %s nug %g aniso %s ball %s" % (it, dim, k, n, ["SK", "OK", "UK", "EDK"][variant] if variant < 4 else variant, kind, nug, aniso, ball)
    if not (np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt)):
        print("INDEX MISMATCH", tag); sys.exit(1)
    ok = (st == 0) & (rst == 0)
    if not np.array_equal(st == 1, rst == 1):
        print("MISSING PATTERN MISMATCH", tag); sys.exit(1)
    if ok.any():
        # the bar follows the conditioning of each point's covariance matrix (a smooth model over hundreds of dense samples
        # on a line is singular to working precision): 1e-8, or 1e3 eps cond(C) where that is larger
        conds = np.array([np.linalg.cond(cov_pairwise(ovg, x[ridx[i][:rcnt[i]]])) if ok[i] else 1.0 for i in range(m)])
        tol = np.maximum(1e-8, 1e3 * 2.2e-16 * conds[ok])
        scale = np.maximum(1.0, np.abs(rvar[ok]))
        e = max(float(np.max(np.abs(mu[ok] - rmu[ok]) / (np.maximum(1.0, np.abs(rmu[ok])) * scale * tol))),
                float(np.max(np.abs(var[ok] - rvar[ok]) / (scale * tol))))
        worst = max(worst, e)
        tol = 1.0
        if e > tol:
            print("VALUE MISMATCH %.3e" % e, tag, "singular flags", int((st == 2).sum()), int((rst == 2).sum())); sys.exit(1)
print("%d cases, worst error / tolerance %.3g" % (cases, worst))
