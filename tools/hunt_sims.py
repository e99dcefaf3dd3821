#!/usr/bin/env python3
"""Random hunt over LUGS (both factorisations, conditional or not, co-simulation) and SGS (both mask readings, paths,
balls, k up to 40) on small grids against the oracle.  python3 tools/hunt_sims.py [seed] [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np  # noqa: E402

import gss  # noqa: E402
from gss.engine import LUGSHandle, SGSHandle  # noqa: E402
from oracle import fftgs as OF, lugs as OL, sgs as OS  # noqa: E402
from oracle.variogram import Variogram  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
CT = dict(exponential=gss.ExponentialVariogram, spherical=gss.SphericalVariogram, matern=gss.MaternVariogram)


def model():
    kind = ["exponential", "spherical", "matern"][int(rng.integers(0, 3))]
    kw = dict(range=float(rng.uniform(3, 15)), sill=float(rng.uniform(0.5, 2.5)), nugget=float(rng.choice([0.0, 0.05, 0.3])))
    okw = dict(kw)
    if kind == "matern":
        okw["nu"] = float(rng.choice([0.5, 1.5, 2.5]))
        kw["order"] = okw["nu"]
    return CT[kind](**kw), Variogram(kind, **okw), "%s %s" % (kind, okw)


worst = 0.0
for it in range(cases):
    d = int(rng.integers(1, 4))
    dims = tuple(int(v) for v in rng.integers(3, {1: 900, 2: 40, 3: 12}[d], d))
    cent = OF.grid_centroids(dims)
    N = cent.shape[0]
    nd = int(rng.integers(0, max(1, min(N // 2, 200))))
    dl = np.sort(rng.choice(N, nd, replace=False)) if nd else np.empty(0, dtype=np.int64)
    zd = rng.normal(size=nd)
    mean = float(rng.choice([0.0, 0.7, -1.2]))
    gvg, ovg, desc = model()
    if rng.random() < 0.5:                                   # ---- LUGS
        fact = "lu" if rng.random() < 0.3 else "cholesky"
        tag = "LUGS %s dims %s nd %d %s" % (fact, dims, nd, desc)
        h = LUGSHandle(gvg, cent, dl, zd, mean=mean, factorization=fact)
        p = OL.preprocess(ovg, cent, cent[dl] if nd else None, zd if nd else None, mean=mean, factorization=fact)
        R = int(rng.integers(1, 6))
        y, w = h.realize(5 + it, 2, R)
        ry, rw = OL.realize(p, 5 + it, 2, R)
        e = float(np.max(np.abs(y - ry)))
        if nd and not np.array_equal(y[:, dl], np.tile(zd, (R, 1))):
            print("DATA NOT HONOURED", tag); sys.exit(1)
        if rng.random() < 0.4:                               # a second variable correlated with the first
            g2, o2, _ = model()
            h2 = LUGSHandle(g2, cent, dl, zd, mean=0.0, factorization=fact)
            p2 = OL.preprocess(o2, cent, cent[dl] if nd else None, zd if nd else None, mean=0.0, factorization=fact)
            rho = float(rng.uniform(-0.95, 0.95))
            y2, _ = h2.realize(9, 0, R, rho=rho, w1=w)
            ry2, _ = OL.realize(p2, 9, 0, R, rho=rho, w1=rw)
            e = max(e, float(np.max(np.abs(y2 - ry2))))
            h2.close()
        h.close()
    else:                                                    # ---- SGS
        k = int(rng.integers(1, min(N - 1, 40) + 1))
        nmin = int(rng.integers(1, min(k, 3) + 1))
        ball = {}
        if rng.random() < 0.5:
            ball = dict(radius=float(rng.uniform(2, 12))) if d == 1 or rng.random() < 0.6 else dict(radii=tuple(float(v) for v in rng.uniform(2, 12, d)))
        mode = int(rng.integers(0, 2))
        path = None if mode == 0 else rng.permutation(N)
        ma = bool(rng.integers(0, 2))
        tag = "SGS dims %s nd %d k %d nmin %d ball %s path %s mask_after %s %s" % (dims, nd, k, nmin, ball, mode, ma, desc)
        h = SGSHandle(gvg, cent, path, dl, zd, mean, k, nmin, ball.get("radius"), ball.get("radii"), mask_after_search=ma)
        R = int(rng.integers(1, 4))
        z = h.realize(42, 1, R)
        h.close()
        ref = OS.realize(ovg, mean, cent, path, dl, zd, 42, 1, R, maxneighbors=k, minneighbors=nmin, mask_after_search=ma, **ball)
        e = float(np.max(np.abs(z - ref)))
        if nd and not np.array_equal(z[:, dl], np.tile(zd, (R, 1))):
            print("DATA NOT HONOURED", tag); sys.exit(1)
    worst = max(worst, e)
    if not e < 1e-8:
        print("MISMATCH %.3e case %d" % (e, it), tag); sys.exit(1)
print("%d cases, worst error %.3g" % (cases, worst))
