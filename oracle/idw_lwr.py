"""IDW and LWR estimation (oracle side; TEST INFRASTRUCTURE ONLY — never imported by the product).

CPU restatement of
    /root/reference/src/estimation/idw.jl:58-153   (IDWSolver.solve)
    /root/reference/src/estimation/lwr.jl:61-158   (LWRSolver.solve)
on top of the neighbour search of `oracle.kriging.knn_search` ([DEP] Meshes KNearestSearch / KBallSearch,
`ui.jl:25-31`).  Parity pinning: the reference holds no numeric golden vectors for these two solvers (its tests
only run them and check units, `test/estimation/idw.jl`, `test/estimation/lwr.jl`); the oracle is pinned by the
closed-form properties those tests imply (exact interpolation at data locations, `idw.jl:127-130`; a linear field
is reproduced exactly by LWR) — see tests/test_oracle_idw_lwr.py.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import numpy as np

from .kriging import metric_dist, metric_key


def _neighbours(x, center, k, radius, radii, distance=None):
    """searchdists!(neighbors, distances, center, searcher) — idw.jl:120, lwr.jl:123.
    Ascending (d2, index); with a ball only d2 <= r2 (Mahalanobis for an anisotropic ball)."""
    inv = None
    r2 = None
    if radii is not None:
        inv = 1.0 / np.asarray(radii, dtype=np.float64)
        r2 = 1.0
    elif radius is not None:
        r2 = float(radius) ** 2
    d2 = metric_key(x, center, distance, inv)
    order = np.argsort(d2, kind="stable")[:k]
    if r2 is not None:
        order = order[d2[order] <= r2]
    return order, metric_dist(d2[order], distance)


def idw(x: np.ndarray, z: np.ndarray, xdom: np.ndarray, maxneighbors: Optional[int] = None, minneighbors: int = 1,
        exponent: float = 1.0, radius: Optional[float] = None, radii: Optional[Sequence[float]] = None,
        distance=None):
    """idw.jl:111-142.  Returns (mu, dist, status): status 1 = fewer than `minneighbors` neighbours (`missing`)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    z = np.asarray(z, dtype=np.float64)
    xdom = np.atleast_2d(np.asarray(xdom, dtype=np.float64))
    n = x.shape[0]
    nmax = n if maxneighbors is None else min(int(maxneighbors), n)      # idw.jl:93
    assert n > 0, "estimation requires data"                              # idw.jl:95
    assert exponent > 0, "exponent must be positive"                      # idw.jl:96
    assert minneighbors <= nmax, "invalid min/max number of neighbors"    # idw.jl:97
    m = xdom.shape[0]
    mu = np.full(m, np.nan)
    sd = np.full(m, np.nan)
    st = np.zeros(m, dtype=np.uint8)
    for p in range(m):
        is_, ds = _neighbours(x, xdom[p], nmax, radius, radii, distance)
        if is_.size < minneighbors:                                       # idw.jl:123-124
            st[p] = 1
            continue
        with np.errstate(divide="ignore"):
            ws = 1.0 / ds ** exponent                                     # idw.jl:128
        sw = ws.sum()
        if np.isinf(sw):                                                  # idw.jl:131-134
            j = int(np.flatnonzero(ds == 0.0)[0])
            mu[p] = z[is_[j]]
            sd[p] = 0.0
        else:                                                             # idw.jl:135-139
            ws = ws / sw
            mu[p] = float(np.sum(ws * z[is_]))
            sd[p] = float(ds.min()) if ds.size else np.nan
    return mu, sd, st


def idw_compositional(x: np.ndarray, parts: np.ndarray, xdom: np.ndarray, maxneighbors: Optional[int] = None,
                      minneighbors: int = 1, exponent: float = 1.0, radius: Optional[float] = None,
                      radii: Optional[Sequence[float]] = None, distance=None):
    """idw.jl:111-142 on a column of compositions (`test/estimation/idw.jl:47-65`), written with the compositions' own
    operations instead of logarithms: `ws[i] * vs[i]` is powering (parts .^ w), `sum` perturbation (parts .* parts), no
    closure ([RECALL] CoDa.jl; the package is not in the tree).  `parts`: (n, D) positive.  Returns (mu (m, D), dist,
    status)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    parts = np.asarray(parts, dtype=np.float64)
    xdom = np.atleast_2d(np.asarray(xdom, dtype=np.float64))
    n = x.shape[0]
    nmax = n if maxneighbors is None else min(int(maxneighbors), n)
    m = xdom.shape[0]
    mu = np.full((m, parts.shape[1]), np.nan)
    sd = np.full(m, np.nan)
    st = np.zeros(m, dtype=np.uint8)
    for p in range(m):
        is_, ds = _neighbours(x, xdom[p], nmax, radius, radii, distance)
        if is_.size < minneighbors:                                       # idw.jl:123-124
            st[p] = 1
            continue
        with np.errstate(divide="ignore"):
            ws = 1.0 / ds ** exponent                                     # idw.jl:128
        sw = ws.sum()
        if np.isinf(sw):                                                  # idw.jl:131-134
            mu[p] = parts[is_[int(np.flatnonzero(ds == 0.0)[0])]]
            sd[p] = 0.0
        else:
            ws = ws / sw                                                  # idw.jl:136
            acc = np.ones(parts.shape[1])
            for w, v in zip(ws, parts[is_]):                              # idw.jl:138: perturbation of the powers
                acc = acc * v ** w
            mu[p] = acc
            sd[p] = float(ds.min())
    return mu, sd, st


def default_weightfun(h):
    """lwr.jl:58: h -> exp(-3 h^2)."""
    return np.exp(-3.0 * h * h)


def tricube(h):
    return (1.0 - h ** 3) ** 3


def exp_weight(a: float, p: float) -> Callable:
    return lambda h: np.exp(-a * h ** p)


def lwr(x: np.ndarray, z: np.ndarray, xdom: np.ndarray, maxneighbors: Optional[int] = None, minneighbors: int = 1,
        weightfun: Callable = default_weightfun, radius: Optional[float] = None,
        radii: Optional[Sequence[float]] = None, distance=None):
    """lwr.jl:114-147.  Returns (mu, var, status); `var` is norm(r) exactly as the reference stores it
    under `<var>_variance` (lwr.jl:141-142,154).  status 1 = too few neighbours, 2 = singular normal equations
    (the reference's `\\` would throw a SingularException there)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    z = np.asarray(z, dtype=np.float64)
    xdom = np.atleast_2d(np.asarray(xdom, dtype=np.float64))
    n = x.shape[0]
    nmax = n if maxneighbors is None else min(int(maxneighbors), n)      # lwr.jl:96
    assert n > 0, "estimation requires data"
    assert minneighbors <= nmax, "invalid min/max number of neighbors"
    m = xdom.shape[0]
    mu = np.full(m, np.nan)
    var = np.full(m, np.nan)
    st = np.zeros(m, dtype=np.uint8)
    for p in range(m):
        is_, ds = _neighbours(x, xdom[p], nmax, radius, radii, distance)
        if is_.size < minneighbors or is_.size == 0:                      # lwr.jl:126-127
            st[p] = 1
            continue
        with np.errstate(invalid="ignore", divide="ignore"):
            deltas = ds / ds.max()                                        # lwr.jl:132
        W = weightfun(deltas)                                             # lwr.jl:136
        X = np.hstack([np.ones((is_.size, 1)), x[is_]])                   # lwr.jl:137
        A = X.T @ (W[:, None] * X)
        b = X.T @ (W * z[is_])
        x0 = np.concatenate([[1.0], xdom[p]])                             # lwr.jl:142
        try:
            if not np.all(np.isfinite(A)) or np.linalg.cond(A) > 1e15:
                raise np.linalg.LinAlgError
            theta = np.linalg.solve(A, b)                                 # lwr.jl:139
            a = np.linalg.solve(A, x0)
        except np.linalg.LinAlgError:
            st[p] = 2
            continue
        mu[p] = float(theta @ x0)                                         # lwr.jl:143
        r = W * (X @ a)                                                   # lwr.jl:144
        var[p] = float(np.linalg.norm(r))                                 # lwr.jl:145
    return mu, var, st
