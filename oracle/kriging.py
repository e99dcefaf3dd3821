"""Kriging estimation oracle (test infrastructure only).

Restates `/root/reference/src/estimation/krig.jl:76-234` (preprocess / solve /
exactsolve / approxsolve), `/root/reference/src/ui.jl:11-50` (searcher_ui /
kriging_ui) and the [DEP] GeoStatsModels 0.2 `fit` / `predictprob` pair those call
(SURVEY.md Appendix A.2):

    LHS = [C F; F' 0]   (covariance form for stationary variograms)
    RHS = [c0; f0]
    [lambda; nu] = LHS \\ RHS           (Cholesky for SK, Bunch-Kaufman otherwise)
    mu      = lambda . z                (SK: mean + lambda . (z - mean))
    sigma^2 = max(0, sill - RHS . [lambda; nu])

Support is a point at the element centroid by default.  `support=(cell, nsub)`
restates what the reference's dependencies do when `predictprob` receives a grid cell (krig.jl:180,226 pass
`pdomain[ind]`): [RECALL] Variography regularises gamma(point, geometry) by averaging over sample points inside the
geometry.  The scheme here -- documented, not the dependency's exact one, which is version dependent -- is the midpoint
rule: `nsub` points per axis at the centres of the sub-cells of a cell of size `cell` around the centroid;
c0_i = mean_s C(x_i, s), f0 = mean_s f(s), sigma^2 = mean_{s,s'} C(s, s') - RHS . [lambda; nu].
"""
from __future__ import annotations

import itertools
import warnings
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import scipy.linalg as sla

from .variogram import Variogram, cov_pairwise, isstationary, pairwise

SK, OK, UK, EDK = 0, 1, 2, 3


# ----------------------------------------------------------------------------
# ui.jl
# ----------------------------------------------------------------------------
def kriging_ui(variogram, mean=None, degree=None, drifts=None) -> int:
    """ui.jl:40-50 -- precedence drifts > degree > mean > ordinary."""
    if drifts is not None:
        return EDK
    if degree is not None:
        return UK
    if mean is not None:
        return SK
    return OK


def searcher_ui(nelem: int, maxneighbors, neighborhood):
    """ui.jl:11-32 -- returns (kind, nmax); warns exactly like ui.jl:19."""
    if maxneighbors is None:
        nmax = nelem
    elif maxneighbors < 1 or maxneighbors > nelem:
        warnings.warn(f"Invalid maximum number of neighbors. Adjusting to {nelem}...")
        nmax = nelem
    else:
        nmax = maxneighbors
    return ("knearest" if neighborhood is None else "kball"), nmax


# ----------------------------------------------------------------------------
# drift terms
# ----------------------------------------------------------------------------
def uk_exponents(dim: int, degree: int) -> np.ndarray:
    """[DEP] monomial exponents with total degree <= `degree`, graded order.

    Column order does not change the kriging weights."""
    exps = []
    for total in range(degree + 1):
        for e in itertools.product(range(total + 1), repeat=dim):
            if sum(e) == total:
                exps.append(e)
    return np.asarray(exps, dtype=np.int64)           # nc x dim


def drift_matrix(variant: int, coords: np.ndarray, degree: Optional[int] = None,
                 drift_vals: Optional[np.ndarray] = None) -> np.ndarray:
    """F (npts x nc): SK none, OK ones, UK monomials, EDK user drift values."""
    n = coords.shape[0]
    if variant == SK:
        return np.zeros((n, 0))
    if variant == OK:
        return np.ones((n, 1))
    if variant == UK:
        ex = uk_exponents(coords.shape[1], int(degree))
        return np.prod(coords[:, None, :] ** ex[None, :, :], axis=-1)
    if variant == EDK:
        return np.asarray(drift_vals, dtype=np.float64).reshape(n, -1)
    raise ValueError(variant)


# ----------------------------------------------------------------------------
# fit / predictprob  [DEP GeoStatsModels]   (call sites krig.jl:176,180,223,226)
# ----------------------------------------------------------------------------
@dataclass
class FittedKriging:
    variant: int
    vg: Variogram
    x: np.ndarray
    z: np.ndarray
    mean: float
    degree: Optional[int]
    n: int
    nc: int
    lhs: np.ndarray
    fact: tuple
    ok: bool


def fit(variant: int, vg: Variogram, x: np.ndarray, z: np.ndarray, mean: float = 0.0,
        degree: Optional[int] = None, drift_data: Optional[np.ndarray] = None) -> FittedKriging:
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    z = np.asarray(z, dtype=np.float64)
    n = x.shape[0]
    F = drift_matrix(variant, x, degree, drift_data)
    nc = F.shape[1]
    lhs = np.zeros((n + nc, n + nc))
    # stationary: covariance form; otherwise (power model) the variogram form [G F; F' 0]
    lhs[:n, :n] = cov_pairwise(vg, x) if isstationary(vg) else pairwise(vg, x)
    lhs[:n, n:] = F
    lhs[n:, :n] = F.T
    ok = True
    if variant == SK:
        c, info = sla.lapack.dpotrf(lhs, lower=1)
        fact = ("chol", c)
        ok = info == 0
    else:
        ldu, piv, info = sla.lapack.dsytrf(lhs, lower=1)
        fact = ("bk", ldu, piv)
        ok = info == 0
    return FittedKriging(variant, vg, x, z, float(mean), degree, n, nc, lhs, fact, ok)


def _solve(fk: FittedKriging, rhs: np.ndarray) -> np.ndarray:
    if fk.fact[0] == "chol":
        sol, _ = sla.lapack.dpotrs(fk.fact[1], rhs, lower=1)
    else:
        sol, _ = sla.lapack.dsytrs(fk.fact[1], fk.fact[2], rhs, lower=1)
    return sol


def block_samples(dim: int, cell, nsub: int) -> np.ndarray:
    """Offsets (nsub^dim x dim) of the sample points of a cell about its centroid: centres of the sub-cells."""
    cell = np.asarray(cell, dtype=np.float64).reshape(dim)
    t = (np.arange(nsub) + 0.5) / nsub - 0.5
    grids = np.meshgrid(*[t * cell[a] for a in range(dim)], indexing="ij")
    return np.stack([g.ravel() for g in grids], axis=1)


def predict(fk: FittedKriging, x0: np.ndarray, drift_dom: Optional[np.ndarray] = None, support=None):
    """predictprob -> (mean, variance) at points x0 (m x d); vectorised over RHS columns.  `support=(cell, nsub)`:
    x0 are cell centroids and the right-hand sides are regularised over the cell (module docstring)."""
    x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
    m = x0.shape[0]
    rhs = np.empty((fk.n + fk.nc, m))
    stat = isstationary(fk.vg)
    cvv = fk.vg.sill
    if support is not None:
        assert stat and fk.variant != EDK, "block support: stationary models, no external drifts"
        off = block_samples(x0.shape[1], support[0], int(support[1]))
        rhs[:] = 0.0
        for o in off:
            rhs[:fk.n] += cov_pairwise(fk.vg, fk.x, x0 + o)
            if fk.nc:
                rhs[fk.n:] += drift_matrix(fk.variant, x0 + o, fk.degree, None).T
        rhs /= len(off)
        cvv = float(np.mean(cov_pairwise(fk.vg, off, off)))       # the same for every cell of a regular grid
    else:
        rhs[:fk.n] = cov_pairwise(fk.vg, fk.x, x0) if stat else pairwise(fk.vg, fk.x, x0)
        if fk.nc:
            rhs[fk.n:] = drift_matrix(fk.variant, x0, fk.degree, drift_dom).T
    w = _solve(fk, rhs)
    lam = w[:fk.n]
    if fk.variant == SK:
        mu = fk.mean + lam.T @ (fk.z - fk.mean)
    else:
        mu = lam.T @ fk.z
    var = cvv - np.sum(rhs * w, axis=0) if stat else np.sum(rhs * w, axis=0)
    return mu, np.maximum(var, 0.0)


# ----------------------------------------------------------------------------
# neighbour search  [DEP Meshes KNearestSearch / KBallSearch]  (krig.jl:210)
# ----------------------------------------------------------------------------
def sqdist(x: np.ndarray, c: np.ndarray, inv_radii: Optional[np.ndarray] = None) -> np.ndarray:
    """Squared distance accumulated in dimension order, one rounding per op, no FMA.

    This is the build's tie/ordering contract: neighbours are ranked by
    ascending (this FP64 value, data index)."""
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    acc = np.zeros(x.shape[0])
    for k in range(x.shape[1]):
        t = x[:, k] - c[k]
        if inv_radii is not None:
            t = t * inv_radii[k]
        acc = acc + t * t
    return acc


def metric_key(x: np.ndarray, c: np.ndarray, distance=None, inv_radii: Optional[np.ndarray] = None) -> np.ndarray:
    """Ranking key of the search for the solver parameter `distance` (krig.jl:72; [DEP] Distances.jl):
    Euclidean -> squared distance (sqdist), "cityblock" -> sum |t|, "chebyshev" -> max |t|,
    ("haversine", r) -> sin^2(dlat/2) + cos(lat1) cos(lat2) sin^2(dlon/2) for (lon, lat) in degrees."""
    name = "euclidean" if distance is None else (distance if isinstance(distance, str) else distance[0])
    if name == "euclidean":
        return sqdist(x, c, inv_radii)
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    if name == "cityblock":
        acc = np.zeros(x.shape[0])
        for k in range(x.shape[1]):
            acc = acc + np.abs(x[:, k] - c[k])
        return acc
    if name == "chebyshev":
        acc = np.zeros(x.shape[0])
        for k in range(x.shape[1]):
            acc = np.maximum(acc, np.abs(x[:, k] - c[k]))
        return acc
    if name == "haversine":
        D = 0.017453292519943295
        s1 = np.sin(((c[1] - x[:, 1]) * 0.5) * D)
        s2 = np.sin(((c[0] - x[:, 0]) * 0.5) * D)
        cc = np.cos(x[:, 1] * D) * np.cos(c[1] * D)
        return s1 * s1 + cc * (s2 * s2)
    raise ValueError(distance)


def metric_dist(key: np.ndarray, distance=None) -> np.ndarray:
    """Distance from the ranking key."""
    name = "euclidean" if distance is None else (distance if isinstance(distance, str) else distance[0])
    if name == "euclidean":
        return np.sqrt(key)
    if name == "haversine":
        return 2.0 * float(distance[1]) * np.arcsin(np.minimum(np.sqrt(key), 1.0))
    return key


def knn_search(x: np.ndarray, centers: np.ndarray, k: int, radius: Optional[float] = None,
               radii: Optional[Sequence[float]] = None, distance=None):
    """Exact k nearest neighbours of each centre among rows of x.

    Returns (idx [m x k] int32, 0-based, -1 padded; count [m]).  With a ball only
    neighbours with d^2 <= r^2 are kept (anisotropic ball: Mahalanobis, r = 1)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    centers = np.atleast_2d(np.asarray(centers, dtype=np.float64))
    m = centers.shape[0]
    n = x.shape[0]
    k = min(k, n)
    inv = None
    r2 = None
    if radii is not None:
        inv = 1.0 / np.asarray(radii, dtype=np.float64)
        r2 = 1.0
    elif radius is not None:
        r2 = float(radius) * float(radius)
    idx = np.full((m, k), -1, dtype=np.int32)
    cnt = np.zeros(m, dtype=np.int32)
    for p in range(m):
        d2 = metric_key(x, centers[p], distance, inv)
        order = np.argsort(d2, kind="stable")[:k]          # (key, index) ascending
        if r2 is not None:
            order = order[d2[order] <= r2]
        idx[p, :order.size] = order
        cnt[p] = order.size
    return idx, cnt


# ----------------------------------------------------------------------------
# exactsolve / approxsolve   (krig.jl:166-234)
# ----------------------------------------------------------------------------
def exactsolve(variant, vg, x, z, xdom, mean=0.0, degree=None, drift_data=None, drift_dom=None, support=None):
    """krig.jl:166-186: fit once on all samples, predict every domain point (`support`: see `predict`)."""
    fk = fit(variant, vg, x, z, mean, degree, drift_data)
    return predict(fk, xdom, drift_dom, support)


def approxsolve(variant, vg, x, z, xdom, maxneighbors, minneighbors=1, mean=0.0, degree=None,
                drift_data=None, drift_dom=None, radius=None, radii=None, return_idx=False, distance=None, support=None):
    """krig.jl:188-234: per point k-NN, fit on the neighbours, predict; too few -> missing (NaN)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    xdom = np.atleast_2d(np.asarray(xdom, dtype=np.float64))
    z = np.asarray(z, dtype=np.float64)
    m = xdom.shape[0]
    idx, cnt = knn_search(x, xdom, maxneighbors, radius, radii, distance)
    mu = np.full(m, np.nan)
    var = np.full(m, np.nan)
    status = np.zeros(m, dtype=np.uint8)
    for p in range(m):
        nn = int(cnt[p])
        if nn < minneighbors or nn == 0:
            status[p] = 1                      # `missing, missing` krig.jl:213-214
            continue
        ii = idx[p, :nn]
        dd = None if drift_data is None else np.asarray(drift_data)[ii]
        d0 = None if drift_dom is None else np.asarray(drift_dom)[p:p + 1]
        fk = fit(variant, vg, x[ii], z[ii], mean, degree, dd)
        a, b = predict(fk, xdom[p:p + 1], d0, support)     # krig.jl:226 hands the cell itself: `support`, see predict
        mu[p], var[p] = a[0], b[0]
    if return_idx:
        return mu, var, status, idx, cnt
    return mu, var, status
