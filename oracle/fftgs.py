"""FFTGS oracle (test infrastructure only).

Restates `/root/reference/src/simulation/fft.jl:62-143` (preprocess) and `:145-198`
(solvesingle) with numpy's pocketfft in place of FFTW ([DEP] conventions, SURVEY.md A.7:
`fft` unnormalised, `ifft` scaled 1/N, `fftshift` rolls each axis by n//2).

Array convention: a Julia array of size dims=(n1,n2,n3) (column-major, n1 fastest) is held
as a numpy C-order array of shape dims[::-1]; flattening either gives the same linear order,
which is the order of grid elements and of every realisation vector.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import kriging, philox
from .variogram import Variogram, cov_pairwise, isstationary


def grid_centroids(dims: Sequence[int], origin=None, spacing=None) -> np.ndarray:
    """Centroids of a CartesianGrid in element (Julia linear) order, shape (N, d)."""
    d = len(dims)
    origin = np.zeros(d) if origin is None else np.asarray(origin, dtype=np.float64)
    spacing = np.ones(d) if spacing is None else np.asarray(spacing, dtype=np.float64)
    axes = [origin[a] + (np.arange(dims[a]) + 0.5) * spacing[a] for a in range(d)]
    mesh = np.meshgrid(*axes[::-1], indexing="ij")          # slowest axis first
    return np.stack([m.ravel() for m in mesh[::-1]], axis=1)


@dataclass
class FFTGSPreproc:
    vg: Variogram
    mean: float
    dims: tuple
    F: np.ndarray                      # spectral amplitude, shape dims[::-1]
    zbar: Optional[np.ndarray] = None  # conditional mean on the problem domain
    krig: Optional[dict] = None
    dinds: Optional[np.ndarray] = None


def preprocess(vg: Variogram, dims: Sequence[int], origin=None, spacing=None, mean: float = 0.0,
               data_coords: Optional[np.ndarray] = None, data_vals: Optional[np.ndarray] = None,
               inds: Optional[np.ndarray] = None, maxneighbors=None, minneighbors=1,
               radius=None) -> FFTGSPreproc:
    dims = tuple(int(x) for x in dims)
    if not isstationary(vg):                                   # fft.jl:91-93
        raise ValueError("variogram model must be stationary")
    cent = grid_centroids(dims, origin, spacing)               # fft.jl:97
    center = tuple(x // 2 for x in dims)                        # fft.jl:69 (1-based CartesianIndex)
    cindex = np.ravel_multi_index(tuple(c - 1 for c in center)[::-1], dims[::-1])  # fft.jl:70
    cs = cov_pairwise(vg, cent[cindex:cindex + 1], cent)[0]          # fft.jl:98
    C = cs.reshape(dims[::-1])                                  # fft.jl:99
    F = np.sqrt(np.abs(np.fft.fftn(np.fft.fftshift(C))))        # fft.jl:102
    F.flat[0] = 0.0                                             # fft.jl:103
    pre = FFTGSPreproc(vg, float(mean), dims, F)
    if data_coords is not None:                                 # fft.jl:105-135
        pdom = cent if inds is None else cent[np.asarray(inds)]
        kw = dict(maxneighbors=maxneighbors, minneighbors=minneighbors, radius=radius)
        pre.krig = kw
        pre.zbar = _krige(vg, mean, np.asarray(data_coords, float), np.asarray(data_vals, float), pdom, kw)
        found = [int(kriging.knn_search(pdom, c[None, :], 1)[0][0, 0]) for c in np.atleast_2d(data_coords)]
        _, first = np.unique(found, return_index=True)          # unique(first.(found)) keeps first-seen order
        pre.dinds = np.asarray(found)[np.sort(first)]
    return pre


def _krige(vg, mean, xd, zd, pdom, kw):
    """`solve(prob, KrigingSolver(var => (variogram, mean, ...)))` -> SK since mean is set (fft.jl:115-126)."""
    if kw.get("maxneighbors") is None:
        mu, _ = kriging.exactsolve(kriging.SK, vg, xd, zd, pdom, mean=mean)
    else:
        _, k = kriging.searcher_ui(xd.shape[0], kw["maxneighbors"], kw.get("radius"))
        mu, _, _ = kriging.approxsolve(kriging.SK, vg, xd, zd, pdom, k, kw.get("minneighbors", 1),
                                       mean=mean, radius=kw.get("radius"))
    return mu


def solvesingle(pre: FFTGSPreproc, noise: np.ndarray, inds: Optional[np.ndarray] = None,
                origin=None, spacing=None) -> np.ndarray:
    """One realisation from uniform `noise` (flat, element order)  -- fft.jl:145-198."""
    shape = pre.dims[::-1]
    N = int(np.prod(shape))
    U = np.asarray(noise, dtype=np.float64).reshape(shape)
    P = pre.F * np.exp(1j * np.angle(np.fft.fftn(U)))           # fft.jl:163
    Z = np.real(np.fft.ifftn(P))                                # fft.jl:166
    s2 = np.sum(Z * Z) / (N - 1)                                # fft.jl:169  var(Z, mean=0)
    Z = np.sqrt(pre.vg.sill / s2) * Z + pre.mean                # fft.jl:170
    zu = Z.ravel() if inds is None else Z.ravel()[np.asarray(inds)]   # fft.jl:173
    if pre.krig is None:
        return zu
    cent = grid_centroids(pre.dims, origin, spacing)
    pdom = cent if inds is None else cent[np.asarray(inds)]
    zbar_u = _krige(pre.vg, pre.mean, pdom[pre.dinds], zu[pre.dinds], pdom, pre.krig)   # fft.jl:178-188
    return pre.zbar + (zu - zbar_u)                             # fft.jl:191


def realize(pre: FFTGSPreproc, seed: int, first_real: int, nreals: int, inds=None) -> np.ndarray:
    """Realisations with the build's Philox noise contract (oracle.philox), shape (nreals, npts)."""
    N = int(np.prod(pre.dims))
    return np.stack([solvesingle(pre, philox.uniform(seed, first_real + r, N), inds) for r in range(nreals)])
