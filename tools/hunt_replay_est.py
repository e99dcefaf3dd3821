"""Replays one LWR case of tools/hunt_estimators.py: the worst point, its neighbour count and the condition number of its
weighted normal equations.  python3 tools/hunt_replay_est.py seed case"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np
from gss.engine import HipEngine
from oracle import idw_lwr as E, kriging as K
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(target + 1):
    dim = int(rng.integers(1, 4)); n = int(rng.integers(2, 700))
    k = int(rng.choice([rng.integers(1, min(n, 64) + 1), rng.integers(1, n + 1), n])); m = int(rng.integers(1, 300))
    x = rng.uniform(0, 100, (n, dim))
    if rng.random() < 0.3 and n > 4: x[1] = x[0]
    nz = int(rng.choice([1, 1, 2, 5])); z = rng.normal(size=(nz, n)) if nz > 1 else rng.normal(size=n)
    x0 = rng.uniform(-10, 110, (m, dim)); x0[0] = x[int(rng.integers(0, n))]
    kw = {}
    if rng.random() < 0.4: kw["radius"] = float(rng.uniform(10, 80))
    elif dim > 1 and rng.random() < 0.2: kw["radii"] = tuple(float(v) for v in rng.uniform(10, 80, dim))
    if not kw and rng.random() < 0.3: kw["distance"] = ["cityblock", "chebyshev"][int(rng.integers(0, 2))]
    nmin = int(rng.integers(1, min(k, 4) + 1))
    if rng.random() < 0.5:
        ex = float(rng.choice([1.0, 2.0, 0.5, 3.0])); kind = "idw"
    else:
        wk = int(rng.integers(0, 2)); kind = "lwr"
print(kind, "dim", dim, "n", n, "k", k, "m", m, "nz", nz, kw, "nmin", nmin, "weight", wk)
zs = z if nz > 1 else z[None, :]
mu, ax, st = HipEngine.lwr(x, z, x0, k, nmin, (wk, 3.0, 2.0), **kw)
wf = E.tricube if wk == 1 else E.exp_weight(3.0, 2.0)
r = E.lwr(x, zs[0], x0, k, nmin, wf, kw.get("radius"), kw.get("radii"), kw.get("distance"))
mu0 = np.asarray(mu) if nz == 1 else np.asarray(mu)[0]
ok = (np.asarray(st) == 0) & (r[2] == 0)
err = np.where(ok, np.abs(mu0 - r[0]) / np.maximum(1, np.abs(r[0])), 0)
i = int(np.argmax(err))
idx, cnt = K.knn_search(x, x0[i:i + 1], k, kw.get("radius"), kw.get("radii"), kw.get("distance"))
nb = idx[0][:cnt[0]]
d = np.sqrt(((x[nb] - x0[i]) ** 2).sum(1)); w = wf(d / d.max())
X = np.hstack([np.ones((len(nb), 1)), x[nb]])
A = X.T @ (w[:, None] * X)
print("worst point", i, "err %.3e" % err[i], "neighbours", len(nb), "weights>0", int((w > 0).sum()), "cond(X'WX) %.3e" % np.linalg.cond(A), "mu", mu0[i], r[0][i])
