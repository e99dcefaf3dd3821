#!/usr/bin/env python3
"""Random hunt over the covariance evaluation on the device (gss_cov_pairwise) against the oracle: every stationary model,
Matern of any order in (0, 12], nested models of up to four structures, anisotropy, dimensions 1..3, lags from 0 and 1e-9
to thousands of ranges.  python3 tools/hunt_covariance.py [seed] [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np  # noqa: E402

import gss  # noqa: E402
from gss.engine import HipEngine  # noqa: E402
from oracle.variogram import Nested, Variogram, cov_pairwise  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
CT = dict(gaussian=gss.GaussianVariogram, exponential=gss.ExponentialVariogram, spherical=gss.SphericalVariogram,
          matern=gss.MaternVariogram, cubic=gss.CubicVariogram, pentaspherical=gss.PentasphericalVariogram,
          sinehole=gss.SineHoleVariogram)


def one(dim):
    kind = list(CT)[int(rng.integers(0, len(CT)))]
    sill = float(rng.uniform(0.2, 3.0))
    # (pure-nugget structures, nugget = sill, except under the Gaussian model, whose regularised nugget would exceed the sill)
    kw = dict(sill=sill, nugget=sill * float(rng.choice([0.0, 0.01, 0.4, 1.0] if kind != "gaussian" else [0.0, 0.01, 0.4])))
    okw = dict(kw)
    if kind == "matern":
        nu = float(rng.choice([0.5, 1.5, 2.5, 1.0, 2.0, 3.0, rng.uniform(0.05, 12.0), rng.uniform(0.9, 1.1), rng.uniform(0.45, 0.55)]))
        kw["order"] = nu; okw["nu"] = nu
    radii = None
    if dim > 1 and rng.random() < 0.4:
        radii = tuple(float(v) for v in 10.0 ** rng.uniform(-1, 2, dim))
    else:
        r = float(10.0 ** rng.uniform(-1, 2)); kw["range"] = r; okw["range"] = r
    g = CT[kind](gss.MetricBall(radii), **kw) if radii else CT[kind](**kw)
    o = Variogram(kind, radii=radii, **okw) if radii else Variogram(kind, **okw)
    return g, o, "%s %s radii %s" % (kind, okw, radii)


worst = 0.0
for it in range(cases):
    dim = int(rng.integers(1, 4))
    ns = int(rng.choice([1, 1, 2, 3, 4]))
    parts = [one(dim) for _ in range(ns)]
    if all(p[1].sill - p[1].nugget <= 0.0 for p in parts):
        continue                                                # nothing but nuggets: not a model the solvers take
    if ns == 1:
        g, o, desc = parts[0]
    else:
        ws = [float(rng.uniform(0.2, 2.0)) for _ in range(ns)]
        g = ws[0] * parts[0][0]
        for w, p in zip(ws[1:], parts[1:]):
            g = g + w * p[0]
        o = Nested([(w, p[1]) for w, p in zip(ws, parts)])
        desc = " + ".join("%.2f * (%s)" % (w, p[2]) for w, p in zip(ws, parts))
    a = rng.uniform(0, 50, (40, dim))
    scale = 10.0 ** rng.uniform(-9, 3.5, (60, 1))
    b = np.vstack([a[:10], a[:10] + 1e-9, a[int(rng.integers(0, 40))] + rng.normal(size=(60, dim)) * scale])
    C = np.asarray(HipEngine.cov_pairwise(g, a, b))
    R = cov_pairwise(o, a, b)
    sill = float(np.max(np.abs(R))) + 1e-300
    e = float(np.max(np.abs(C - R))) / sill
    worst = max(worst, e)
    if not e < 5e-13:
        i, j = np.unravel_index(np.argmax(np.abs(C - R)), C.shape)
        print("MISMATCH %.3e case %d dim %d at lag %.6e: device %.17g oracle %.17g | %s" %
              (e, it, dim, float(np.linalg.norm(a[i] - b[j])), C[i, j], R[i, j], desc)); sys.exit(1)
print("%d cases, worst relative difference %.3g" % (cases, worst))
