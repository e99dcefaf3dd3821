"""SGS oracle known answers and the SGS host front-end (no GPU).  The reference asserts only that hard data are
honoured (test/simulation/sgs.jl:18-20); the two-cell closed form pins the kriging/draw arithmetic."""
import numpy as np
import pytest

import gss
from oracle import fftgs as offt, philox, sgs as S
from oracle.variogram import Variogram, cov_h
from oracle_engine import OracleEngine


def test_two_cell_closed_form_and_marginal_fallback():
    vg = Variogram("exponential", range=3.0, sill=2.0)
    cent = np.array([[0.0], [1.0]])
    eps = np.array([0.3, -1.2])
    z = S.solvesingle(vg, 0.5, cent, np.arange(2), np.empty(0, dtype=int), np.empty(0), eps, maxneighbors=1)
    rho = cov_h(vg, np.array(1.0)) / 2.0
    z0 = 0.5 + np.sqrt(2.0) * 0.3                                              # nothing simulated yet: marginal
    assert np.isclose(z[0], z0)
    assert np.isclose(z[1], 0.5 + rho * (z0 - 0.5) + np.sqrt(2.0 * (1 - rho ** 2)) * -1.2)
    # ball that excludes the first cell -> second cell is marginal too (seq.jl:107-109)
    z = S.solvesingle(vg, 0.5, cent, np.arange(2), np.empty(0, dtype=int), np.empty(0), eps, maxneighbors=1, radius=0.5)
    assert np.isclose(z[1], 0.5 + np.sqrt(2.0) * -1.2)
    # minneighbors = 2 can never be met with one predecessor
    z = S.solvesingle(vg, 0.5, cent, np.arange(2), np.empty(0, dtype=int), np.empty(0), eps, maxneighbors=2,
                      minneighbors=2)
    assert np.isclose(z[1], 0.5 + np.sqrt(2.0) * -1.2)


def test_reference_case_honours_hard_data():                 # test/simulation/sgs.jl:2-20
    cent = offt.grid_centroids((100, 100), (0.5, 0.5), (1.0, 1.0))
    x = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
    dl = np.array([int(np.argmin(((cent - p) ** 2).sum(1))) for p in x])
    order = np.argsort(dl)
    vg = Variogram("spherical", range=35.0)
    z = S.realize(vg, 0.0, cent, None, dl[order], np.array([1.0, 0.0, 1.0])[order], 2017, 0, 1, radius=30.0)[0]
    Z = z.reshape(100, 100).T                                 # Z[i, j] = cell (i+1, j+1) of LinearIndices
    assert Z[24, 24] == 1.0 and Z[49, 74] == 0.0 and Z[74, 49] == 1.0
    assert np.all(np.isfinite(z)) and 0.3 < z.std() < 2.0


def test_unconditional_moments_and_random_path():
    cent = offt.grid_centroids((24, 24))
    vg = Variogram("spherical", range=6.0, sill=1.5)
    path = np.random.default_rng(0).permutation(cent.shape[0])
    zs = S.realize(vg, 2.0, cent, path, np.empty(0, dtype=int), np.empty(0), 7, 0, 12, maxneighbors=8)
    assert abs(zs.mean() - 2.0) < 0.25 and abs(zs.var() - 1.5) < 0.4
    g = zs.reshape(12, 24, 24)
    lag1 = np.mean((g[:, :, 1:] - 2.0) * (g[:, :, :-1] - 2.0))
    assert abs(lag1 - float(cov_h(vg, np.array(1.0)))) < 0.35
    # realisation r depends on (seed, r) only
    again = S.realize(vg, 2.0, cent, path, np.empty(0, dtype=int), np.empty(0), 7, 5, 2, maxneighbors=8)
    assert np.array_equal(again, zs[5:7])


def test_sgs_solver_through_solve_with_stand_in():
    grid = gss.CartesianGrid(20, 15)
    data = gss.georef(dict(z=[1.0, 0.0, 1.0]), [(5.2, 4.9), (10.0, 12.0), (15.7, 7.5)])
    solver = gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=8.0), neighborhood=gss.MetricBall(7.0))),
                     rng=11, engine=OracleEngine)
    sol = gss.solve(gss.SimulationProblem(data, grid, "z", 3), solver)
    reals = sol["z"]
    cent = grid.centroids()
    assert len(reals) == 3 and reals[0].shape == (300,)
    for p, v in zip([(5.2, 4.9), (10.0, 12.0), (15.7, 7.5)], [1.0, 0.0, 1.0]):
        j = int(np.argmin(((cent - np.array(p)) ** 2).sum(1)))
        assert all(r[j] == v for r in reals)                                 # test/simulation/sgs.jl:18-20
    usol = gss.solve(gss.SimulationProblem(grid, {"z": float}, 2),
                     gss.SGS(("z", dict(path=("random", 3), maxneighbors=5)), rng=11, engine=OracleEngine))
    assert len(usol["z"]) == 2 and np.all(np.isfinite(usol["z"][1]))
    with pytest.raises(ValueError):
        gss.SGS(("z", dict(nope=1)))
    with pytest.raises(NotImplementedError):
        gss.solve(gss.SimulationProblem(grid, {"z": float}, 1), gss.SGS(("z", dict(path="source")),
                                                                        engine=OracleEngine))
    # RandomPath: every realisation walks its own permutation (seq.jl:99-102 calls traverse inside solvesingle)
    sol_r = gss.solve(gss.SimulationProblem(data, grid, "z", 3),
                      gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=8.0), path=("random", 3), maxneighbors=5)),
                              rng=11, engine=OracleEngine))
    from oracle import sgs as osgs
    from oracle.variogram import Variogram
    N = cent.shape[0]
    dl = np.array(sorted(int(np.argmin(((cent - np.array(p)) ** 2).sum(1))) for p in [(5.2, 4.9), (10.0, 12.0), (15.7, 7.5)]))
    zd = np.array([{int(np.argmin(((cent - np.array(p)) ** 2).sum(1))): v
                    for p, v in zip([(5.2, 4.9), (10.0, 12.0), (15.7, 7.5)], [1.0, 0.0, 1.0])}[j] for j in dl])
    for r in range(3):
        path = np.random.default_rng([3, r]).permutation(N)
        ref = osgs.realize(Variogram("spherical", range=8.0), 0.0, cent, path, dl, zd, 11, r, 1, maxneighbors=5,
                           mask_after_search=True)[0]          # the front-end default: mask = "after"
        assert np.array_equal(sol_r["z"][r], ref)
    sol_d = gss.solve(gss.SimulationProblem(data, grid, "z", 1),
                      gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=8.0), path=("random", 3), maxneighbors=5)),
                              rng=11, engine=OracleEngine, mask="during"))
    ref_d = osgs.realize(Variogram("spherical", range=8.0), 0.0, cent, np.random.default_rng([3, 0]).permutation(N), dl, zd,
                         11, 0, 1, maxneighbors=5)[0]
    assert np.array_equal(sol_d["z"][0], ref_d) and not np.array_equal(sol_d["z"][0], sol_r["z"][0])
