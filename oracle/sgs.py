"""Sequential Gaussian simulation (oracle side; TEST INFRASTRUCTURE ONLY — never imported by the product).

CPU restatement of
    /root/reference/src/simulation/sgs.jl:56-89    (SGS -> SeqSim with SimpleKriging + Normal marginal)
    /root/reference/src/simulation/seq.jl:76-141   (SeqSim.solvesingle: the path loop)
It walks the path once PER REALISATION exactly as the reference does (search among simulated cells, fit, draw);
it does not use the device's weight-sharing shortcut, so agreement with the device checks that shortcut too.
Draws: rand(rng, Normal(mu, sigma)) = mu + sigma * eps with eps = philox.normal(seed, realisation)[cell]
(the build's counter-based contract, DESIGN.md section 3 — Julia's MersenneTwister stream is not reproduced).
Parity pinning: the reference's only numeric assertion for this path is that hard data are honoured exactly
(test/simulation/sgs.jl:18-20); tests/test_oracle_sgs.py checks it together with the analytic two-cell case.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import philox
from .kriging import metric_key
from .variogram import cov_pairwise


def solvesingle(vg, mean: float, cent: np.ndarray, path: np.ndarray, dlocs: np.ndarray, zdata: np.ndarray,
                eps: np.ndarray, maxneighbors: int = 10, minneighbors: int = 1, radius: Optional[float] = None,
                radii: Optional[Sequence[float]] = None, mask_after_search: bool = False,
                distance=None) -> np.ndarray:
    """seq.jl:76-141 for one realisation; `eps[cell]` is the standard normal consumed at that cell.
    `mask_after_search`: how `search!(neighbors, p, searcher, mask=simulated)` (seq.jl:105) treats the mask -- False: the k
    nearest among the simulated cells; True: the k nearest cells of the whole domain (p itself included), of which the
    simulated ones are kept ([DEP] Meshes KNearestSearch / KBallSearch as recalled: the mask filters the tree query)."""
    cent = np.atleast_2d(np.asarray(cent, dtype=np.float64))
    N = cent.shape[0]
    real = np.zeros(N)
    simulated = np.zeros(N, dtype=bool)
    if len(dlocs):                                                          # initbuff, seq.jl:85
        real[dlocs] = zdata
        simulated[dlocs] = True
    inv = None
    r2 = None
    if radii is not None:
        inv = 1.0 / np.asarray(radii, dtype=np.float64)
        r2 = 1.0
    elif radius is not None:
        r2 = float(radius) ** 2
    k = min(int(maxneighbors), N)                                           # searcher_ui, ui.jl:16-23
    sill = float(vg.sill)
    smarg = np.sqrt(sill)                                                   # sgs.jl:66
    for ind in path:                                                        # seq.jl:102
        if simulated[ind]:
            continue
        if mask_after_search:
            d2 = metric_key(cent, cent[ind], distance, inv)                 # `distance` parameter, seq.jl:91-98
            order = np.argsort(d2, kind="stable")[:k]
            if r2 is not None:
                order = order[d2[order] <= r2]
            nb = order[simulated[order]]
        else:
            cand = np.flatnonzero(simulated)                                # search!(..., mask=simulated) seq.jl:105
            d2 = metric_key(cent[cand], cent[ind], distance, inv)
            order = np.argsort(d2, kind="stable")[:k]                       # (key, index) ascending
            if r2 is not None:
                order = order[d2[order] <= r2]
            nb = cand[order]
        if nb.size < minneighbors or nb.size == 0:                          # seq.jl:107-109
            real[ind] = mean + smarg * eps[ind]
        else:
            C = cov_pairwise(vg, cent[nb])                                  # fit(SimpleKriging), seq.jl:121
            c0 = cov_pairwise(vg, cent[nb], cent[ind][None, :])[:, 0]
            try:
                L = np.linalg.cholesky(C)
                lam = np.linalg.solve(L.T, np.linalg.solve(L, c0))
                mu = mean + lam @ (real[nb] - mean)                         # predictprob, seq.jl:126
                var = max(0.0, sill - lam @ c0)
                real[ind] = mu + np.sqrt(var) * eps[ind]                    # seq.jl:129
            except np.linalg.LinAlgError:                                   # status(fitted) false, seq.jl:127
                real[ind] = mean + smarg * eps[ind]
        simulated[ind] = True                                               # seq.jl:133
    return real


def realize(vg, mean, cent, path, dlocs, zdata, seed: int, first_real: int, nreals: int, **kw) -> np.ndarray:
    N = np.atleast_2d(cent).shape[0]
    if path is None:
        path = np.arange(N)
    return np.stack([solvesingle(vg, mean, cent, path, dlocs, zdata, philox.normal(seed, first_real + r, N), **kw)
                     for r in range(nreals)]) if nreals else np.empty((0, N))
