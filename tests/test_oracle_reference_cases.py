"""The reference's own numerical assertions for this path, run against the oracle.

These are the only places where the reference pins results: kriging exactness
at three data cells with atol=1e-3 (test/estimation/krig.jl:35-37,50-52,70-72), FFTGS domain/length
on a grid view (test/simulation/fft.jl:21-22) and the ui dispatch rules (test/ui.jl:6-37)."""
import warnings

import numpy as np
import pytest

from oracle import fftgs, kriging as K, lugs, philox
from oracle.variogram import Variogram

VG = Variogram("gaussian", range=35.0, nugget=0.0)
X2 = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
Z2 = np.array([1.0, 0.0, 1.0])
GRID2 = fftgs.grid_centroids((100, 100), (0.5, 0.5), (1.0, 1.0))    # CartesianGrid((100,100),(0.5,0.5),(1.,1.))


def _Z(mu):
    return mu.reshape(100, 100).T            # Z[i, j] as Julia's asarray


def test_krig_2d_global():                    # test/estimation/krig.jl:22-37
    mu, _ = K.exactsolve(K.OK, VG, X2, Z2, GRID2)
    Z = _Z(mu)
    assert abs(Z[24, 24] - 1.0) < 1e-3 and abs(Z[49, 74] - 0.0) < 1e-3 and abs(Z[74, 49] - 1.0) < 1e-3


def test_krig_2d_nearest():                   # test/estimation/krig.jl:39-52
    mu, _, st = K.approxsolve(K.OK, VG, X2, Z2, GRID2, 3)
    Z = _Z(mu)
    assert abs(Z[24, 24] - 1.0) < 1e-3 and abs(Z[49, 74] - 0.0) < 1e-3 and abs(Z[74, 49] - 1.0) < 1e-3
    assert not st.any()


def test_krig_2d_local_ball():                # test/estimation/krig.jl:54-72
    mu, _, st = K.approxsolve(K.OK, VG, X2, Z2, GRID2, 3, radius=100.0)
    S = mu                                    # linear index = (j-1)*100 + i
    assert abs(S[24 * 100 + 24] - 1.0) < 1e-3 and abs(S[74 * 100 + 49] - 0.0) < 1e-3 \
        and abs(S[49 * 100 + 74] - 1.0) < 1e-3


def test_krig_1d_runs():                      # test/estimation/krig.jl:6-19 (no assertions there)
    x = np.arange(0.0, 101.0, 10.0)[:, None]
    z = np.array([0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.4, 0.3, 0.2, 0.1, 0.0])
    g = fftgs.grid_centroids((100,))
    mu, var = K.exactsolve(K.OK, VG, x, z, g)
    assert np.all(np.isfinite(mu)) and np.all(var >= 0)
    mu3, _, _ = K.approxsolve(K.OK, VG, x, z, g, 3)
    mub, _, _ = K.approxsolve(K.OK, VG, x, z, g, 3, radius=100.0)
    assert np.allclose(mu3, mub)              # the ball of radius 100 never binds here


def test_fftgs_view_length_and_domain():      # test/simulation/fft.jl:14-22
    pre = fftgs.preprocess(Variogram("gaussian", range=10.0), (100, 100))
    inds = np.arange(5000)
    z = fftgs.realize(pre, 2022, 0, 3, inds=inds)
    assert z.shape == (3, 5000)


def test_fftgs_anisotropic_and_conditional_run():   # test/simulation/fft.jl:8-12,24-32
    pre = fftgs.preprocess(Variogram("gaussian", radii=(20.0, 5.0)), (100, 100))
    z = fftgs.realize(pre, 2019, 0, 1)
    assert np.all(np.isfinite(z))
    pre = fftgs.preprocess(Variogram("gaussian", range=10.0), (100, 100), data_coords=X2, data_vals=[1.0, -1.0, 1.0])
    z = fftgs.solvesingle(pre, philox.uniform(2022, 0, 10000))
    assert z.shape == (10000,) and np.all(np.isfinite(z))


def test_lugs_cases_run():                    # test/simulation/lu.jl:8-45 (no assertions there)
    cent = fftgs.grid_centroids((100,))
    xd = np.array([[0.0], [25.0], [50.0], [75.0], [100.0]])
    zd = np.array([0.0, 1.0, 0.0, 1.0, 0.0])
    vg = Variogram("spherical", range=10.0)
    p = lugs.preprocess(vg, cent, xd, zd)
    y, _ = lugs.realize(p, 123, 0, 2)
    assert y.shape == (2, 100) and np.array_equal(y[0, p.dlocs], p.z1)
    pu = lugs.preprocess(vg, cent)
    yu, _ = lugs.realize(pu, 123, 0, 2)
    assert yu.shape == (2, 100)
    p1 = lugs.preprocess(vg, cent, xd, zd, factorization="lu")             # lu.jl:72 custom factorization
    assert p1.L22.shape == p.L22.shape


def test_lugs_2d_and_anisotropic_inputs():    # test/simulation/lu.jl:29-64 (no assertions there)
    """Co-simulation with `GaussianVariogram(range=10.0)` on 500 cells and the two 100 x 100 cases with
    `GaussianVariogram(range=10.0)` / `GaussianVariogram(MetricBall((20., 5.)))`, all with the nugget left at its
    default 0.  In exact arithmetic these covariances are positive definite; in float64 the Cholesky of lu.jl:128 hits a
    non-positive pivot (condition number far beyond 1e16) -- shown here with `regularize=False`.  That the reference's
    suite runs them is the evidence for SURVEY.md A.4: its Variography evaluates the Gaussian model with
    `nugget + 1e-6` ([RECALL] for the value), which `oracle.variogram` applies by default.  With it the inputs run as
    written and honour the statistics."""
    from oracle.variogram import cov_pairwise
    cent = fftgs.grid_centroids((100, 100))
    with pytest.raises(np.linalg.LinAlgError):
        np.linalg.cholesky(cov_pairwise(Variogram("gaussian", range=10.0, regularize=False), cent[:2500]))  # lu.jl:124,128
    with pytest.raises(np.linalg.LinAlgError):                                    # a quarter of the anisotropic case
        np.linalg.cholesky(cov_pairwise(Variogram("gaussian", radii=(20.0, 5.0), regularize=False), cent[:2500]))
    with pytest.raises(np.linalg.LinAlgError):                                    # the co-simulation's second variable
        np.linalg.cholesky(cov_pairwise(Variogram("gaussian", range=10.0, regularize=False), fftgs.grid_centroids((500,))))
    # as written (lu.jl:41-52): 100 x 100, GaussianVariogram(range=10.0), three realisations
    p = lugs.preprocess(Variogram("gaussian", range=10.0), cent)
    y, _ = lugs.realize(p, 123, 0, 3)
    assert y.shape == (3, 10000) and np.all(np.isfinite(y)) and 0.5 < y.var() < 2.0
    # as written (lu.jl:54-64): 100 x 100, GaussianVariogram(MetricBall((20., 5.)))
    p = lugs.preprocess(Variogram("gaussian", radii=(20.0, 5.0)), cent)
    y, _ = lugs.realize(p, 123, 0, 3)
    f = y.reshape(3, 100, 100)                # [r, j, i]: x is the fast axis, the long radius lies along x
    assert np.all(np.isfinite(y)) and np.mean(np.abs(np.diff(f, axis=2))) < np.mean(np.abs(np.diff(f, axis=1)))
    # as written (lu.jl:29-39): co-simulation on 500 cells, spherical z, Gaussian y, correlation 0.95
    c5 = fftgs.grid_centroids((500,))
    pz = lugs.preprocess(Variogram("spherical", range=10.0), c5)
    py = lugs.preprocess(Variogram("gaussian", range=10.0), c5)
    z, w1 = lugs.realize(pz, 123, 0, 1)
    yy, _ = lugs.realize(py, 124, 0, 1, rho=0.95, w1=w1)
    assert z.shape == (1, 500) and yy.shape == (1, 500) and np.all(np.isfinite(yy))


def test_fftgs_and_kriging_gaussian_inputs_as_written():
    """test/simulation/fft.jl:3-12,25-32 and test/estimation/krig.jl:6-19 use `GaussianVariogram` with nugget 0 as
    well; they run either way (no factorisation of a dense lattice), the epsilon only moves the numbers by ~1e-6."""
    pre = fftgs.preprocess(Variogram("gaussian", range=10.0), (100, 100))
    pre0 = fftgs.preprocess(Variogram("gaussian", range=10.0, regularize=False), (100, 100))
    assert 0 < np.max(np.abs(pre.F - pre0.F)) < 1e-3 * np.max(pre0.F)
    mu, _ = K.exactsolve(K.OK, VG, X2, Z2, GRID2)
    mu0, _ = K.exactsolve(K.OK, Variogram("gaussian", range=35.0, nugget=0.0, regularize=False), X2, Z2, GRID2)
    assert 0 < np.max(np.abs(mu - mu0)) < 1e-5


def test_krig_2d_custom_path():               # test/estimation/krig.jl:78-90 (runs, no assertion there)
    """`path=MultiGridPath()`: an estimation solver visits the cells in that order and stores its results in it
    (krig.jl:179-183); the estimates are those of the linear path, permuted."""
    import gss
    from oracle_engine import OracleEngine
    data = gss.georef({"z": Z2}, X2)
    grid = gss.CartesianGrid((100, 100), (0.5, 0.5), (1.0, 1.0))
    prob = gss.EstimationProblem(data, grid, "z")
    kw = dict(variogram=gss.GaussianVariogram(range=35.0, nugget=0.0), maxneighbors=3, neighborhood=gss.MetricBall(100.0))
    lin = gss.solve(prob, gss.KrigingSolver(("z", kw), engine=OracleEngine))
    mg = gss.solve(prob, gss.KrigingSolver(("z", dict(kw, path="multigrid")), engine=OracleEngine))
    order = gss.solvers.multigrid_order((100, 100))
    assert sorted(order.tolist()) == list(range(10000)) and order[0] == 0 and order[1] == 64   # coarsest level first
    assert np.array_equal(mg["z"], lin["z"][order]) and np.array_equal(mg["z_variance"], lin["z_variance"][order])


def test_ui_dispatch():                       # test/ui.jl:6-37
    assert K.searcher_ui(3, 2, None) == ("knearest", 2)
    assert K.searcher_ui(3, 2, "ball") == ("kball", 2)
    assert K.searcher_ui(3, None, None) == ("knearest", 3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert K.searcher_ui(3, 4, None) == ("knearest", 3)
        assert str(w[0].message) == "Invalid maximum number of neighbors. Adjusting to 3..."
    assert K.kriging_ui("g") == K.OK and K.kriging_ui("g", 0.0) == K.SK
    assert K.kriging_ui("g", None, 2) == K.UK and K.kriging_ui("g", None, None, [lambda x: 1]) == K.EDK
    assert K.kriging_ui("g", 1.0, 2, [lambda x: 1]) == K.EDK               # latter options override former
