"""LUGS oracle (test infrastructure only).

Restates `/root/reference/src/simulation/lu.jl:76-169` (preprocess) and `:171-224`
(solvesingle / lusim) with LAPACK via scipy ([DEP] LinearAlgebra `cholesky`, `\\`, `*`).
`initbuff(domain, vars, NearestInit(); data)` ([DEP] GeoStatsBase, lu.jl:86) is restated as:
each datum is copied to its nearest domain element (later data overwrite earlier ones).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import scipy.linalg as sla

from . import kriging, philox
from .variogram import Variogram, cov_pairwise, isstationary


def initbuff_nearest(centroids: np.ndarray, data_coords, data_vals):
    N = centroids.shape[0]
    buff = np.zeros(N)
    mask = np.zeros(N, dtype=bool)
    if data_coords is not None:
        xd = np.atleast_2d(np.asarray(data_coords, dtype=np.float64))
        zd = np.asarray(data_vals, dtype=np.float64)
        for i in range(xd.shape[0]):
            if np.isnan(zd[i]):
                continue
            j = int(kriging.knn_search(centroids, xd[i:i + 1], 1)[0][0, 0])
            buff[j] = zd[i]
            mask[j] = True
    return buff, mask


@dataclass
class LUGSParams:
    z1: np.ndarray
    d2: np.ndarray
    L22: np.ndarray
    mean: float
    dlocs: np.ndarray
    slocs: np.ndarray


def _factor(mat: np.ndarray, factorization: str) -> np.ndarray:
    if factorization == "cholesky":
        return np.linalg.cholesky(mat)                           # cholesky(Symmetric(.)).L
    if factorization == "lu":
        _, L, _ = sla.lu(mat)                                    # lu(Symmetric(.)).L  (lu.jl:70,107)
        return L
    raise ValueError(factorization)


def preprocess(vg: Variogram, centroids: np.ndarray, data_coords=None, data_vals=None,
               mean: Optional[float] = None, factorization: str = "cholesky") -> LUGSParams:
    assert isstationary(vg), "variogram model must be stationary"        # lu.jl:110
    buff, mask = initbuff_nearest(centroids, data_coords, data_vals)     # lu.jl:86
    dlocs = np.flatnonzero(mask)                                         # lu.jl:113
    z1 = buff[dlocs]                                                     # lu.jl:114
    slocs = np.flatnonzero(~mask)                                        # lu.jl:117
    Dd, Ds = centroids[dlocs], centroids[slocs]
    C22 = cov_pairwise(vg, Ds)                                           # lu.jl:124
    if dlocs.size == 0:
        d2 = np.zeros(slocs.size)
        L22 = _factor(C22, factorization)                                # lu.jl:128
    else:
        C11 = cov_pairwise(vg, Dd)                                       # lu.jl:131
        C12 = cov_pairwise(vg, Dd, Ds)                                   # lu.jl:132
        L11 = _factor(C11, factorization)                                # lu.jl:134
        B12 = sla.solve_triangular(L11, C12, lower=True)                 # lu.jl:135
        d2 = B12.T @ sla.solve_triangular(L11, z1, lower=True)           # lu.jl:136-138
        L22 = _factor(C22 - B12.T @ B12, factorization)                  # lu.jl:139
    mu = 0.0 if mean is None else float(mean)                            # lu.jl:147
    return LUGSParams(z1, d2, L22, mu, dlocs, slocs)


def lusim(p: LUGSParams, w2: np.ndarray, rho: Optional[float] = None, w1: Optional[np.ndarray] = None):
    """lu.jl:198-224 with the normal draw `w2` supplied by the caller."""
    npts = p.dlocs.size + p.slocs.size
    y = np.empty(npts)
    if rho is None:
        y2 = p.d2 + p.L22 @ w2                                           # lu.jl:211
    else:
        y2 = p.d2 + p.L22 @ (rho * w1 + np.sqrt(1 - rho ** 2) * w2)      # lu.jl:213
    y[p.dlocs] = p.z1                                                    # lu.jl:217
    y[p.slocs] = y2                                                      # lu.jl:218
    if p.dlocs.size == 0:
        y = y + p.mean                                                   # lu.jl:221
    return y, w2


def realize(p: LUGSParams, seed: int, first_real: int, nreals: int, var_index: int = 0,
            rho: Optional[float] = None, w1: Optional[np.ndarray] = None):
    """Philox-noise realisations, shape (nreals, npts); also returns the normals used."""
    ns = p.slocs.size
    ys, ws = [], []
    for r in range(nreals):
        w2 = philox.normal(seed + var_index, first_real + r, ns)
        y, _ = lusim(p, w2, rho, None if w1 is None else w1[r])
        ys.append(y)
        ws.append(w2)
    return np.stack(ys), np.stack(ws)
