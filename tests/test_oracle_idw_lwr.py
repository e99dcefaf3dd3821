"""IDW / LWR oracle known answers and the host front-ends IDWSolver / LWRSolver (no GPU).

The reference's tests for these solvers (test/estimation/idw.jl, test/estimation/lwr.jl) run them on tiny
inputs and assert units only; the numeric pins below are closed forms of idw.jl:128-139 and lwr.jl:132-145."""
import numpy as np
import pytest

import gss
from oracle import fftgs as offt, idw_lwr as E
from oracle_engine import OracleEngine


def test_idw_closed_forms():
    x = np.array([[0.0], [3.0]])
    z = np.array([2.0, 8.0])
    mu, sd, st = E.idw(x, z, np.array([[1.0], [0.0], [4.5]]))
    # weights 1/1 and 1/2 -> (2 + 4) / 1.5 ; zero distance copies the sample (idw.jl:131-134)
    assert np.allclose(mu, [4.0, 2.0, (2 / 4.5 + 8 / 1.5) / (1 / 4.5 + 1 / 1.5)], atol=1e-15)
    assert np.allclose(sd, [1.0, 0.0, 1.5]) and not st.any()
    mu2, _, _ = E.idw(x, z, np.array([[1.0]]), exponent=2)
    assert np.isclose(mu2[0], (2 / 1 + 8 / 4) / (1 + 1 / 4))
    # maxneighbors = 1 -> nearest sample value; ball that holds nothing -> missing (idw.jl:123-124)
    mu1, sd1, st1 = E.idw(x, z, np.array([[1.0], [2.9]]), maxneighbors=1)
    assert np.allclose(mu1, [2.0, 8.0])
    _, _, stb = E.idw(x, z, np.array([[10.0]]), maxneighbors=2, radius=1.0)
    assert stb[0] == 1
    with pytest.raises(AssertionError):
        E.idw(x, z, x, exponent=0)
    with pytest.raises(AssertionError):
        E.idw(x, z, x, maxneighbors=1, minneighbors=2)


def test_idw_reference_test_inputs():                    # test/estimation/idw.jl:3-9, 57-71 (scalar analogue)
    x = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
    z = np.array([1.0, 0.0, 1.0])
    grid = offt.grid_centroids((100, 100))
    mu, sd, st = E.idw(x, z, grid, maxneighbors=3)
    Z = mu.reshape(100, 100).T
    assert abs(Z[24, 24] - 1) < 5e-2 and abs(Z[49, 74]) < 5e-2 and abs(Z[74, 49] - 1) < 5e-2
    assert mu.min() >= 0.0 and mu.max() <= 1.0 and not st.any()       # convex combination of the data
    assert np.isclose(sd.reshape(100, 100).T[24, 24], np.hypot(0.5, 0.5))


def test_lwr_reproduces_linear_fields_and_reference_1d_case():
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 10, (60, 2))
    z = 1.5 - 0.7 * x[:, 0] + 0.2 * x[:, 1]
    dom = rng.uniform(0, 10, (25, 2))
    for k in (5, 12, None):
        mu, var, st = E.lwr(x, z, dom, maxneighbors=k)
        assert np.allclose(mu, 1.5 - 0.7 * dom[:, 0] + 0.2 * dom[:, 1], atol=1e-9) and not st.any()
        assert np.all(var > 0)
    # test/estimation/lwr.jl:3-16: y = x^2 + small noise on 100 points, maxneighbors = 10
    N = 100
    xs = np.linspace(0, 1, N)
    y = xs ** 2 + np.array([i / 1000 for i in range(1, N + 1)]) * np.random.default_rng(2017).normal(size=N)
    cent = offt.grid_centroids((N,), (0.0,), (1.0 / N,))
    yhat, yvar, st = E.lwr(xs[:, None], y, cent, maxneighbors=10)
    assert np.max(np.abs(yhat - cent[:, 0] ** 2)) < 0.15 and np.all(np.isfinite(yvar))
    # k = 2 collinear-free 1-D fit is the chord; k = 1 cannot fit a line -> singular
    mu, _, st = E.lwr(np.array([[0.0], [2.0]]), np.array([1.0, 3.0]), np.array([[0.5]]), maxneighbors=2)
    assert np.isclose(mu[0], 1.5)
    _, _, st = E.lwr(np.array([[0.0], [2.0]]), np.array([1.0, 3.0]), np.array([[0.5]]), maxneighbors=1)
    assert st[0] == 2


def test_lwr_variance_is_norm_of_the_linear_smoother_row():
    # mean = r' z / (weights) : the predictor is linear in z with coefficients r_l / w_l ... check lwr.jl:144-145
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (30, 1))
    z = rng.normal(size=30)
    p = np.array([[0.4]])
    mu, var, _ = E.lwr(x, z, p, maxneighbors=8)
    order = np.argsort(np.abs(x[:, 0] - 0.4), kind="stable")[:8]
    d = np.abs(x[order, 0] - 0.4)
    W = np.exp(-3 * (d / d.max()) ** 2)
    X = np.c_[np.ones(8), x[order]]
    A = X.T @ (W[:, None] * X)
    r = W * (X @ np.linalg.solve(A, np.array([1.0, 0.4])))
    assert np.isclose(var[0], np.linalg.norm(r)) and np.isclose(mu[0], r @ z[order])


def _problem(n=40, dims=(12, 10), seed=0, missing=True):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 10, (n, 2))
    z = np.sin(x[:, 0]) + 0.1 * x[:, 1]
    if missing:
        z[::7] = np.nan
    data = gss.georef(dict(z=z), x)
    return gss.EstimationProblem(data, gss.CartesianGrid(*dims), "z"), x, z


def test_idw_and_lwr_solvers_through_solve_with_stand_in():
    prob, x, z = _problem()
    keep = ~np.isnan(z)
    grid = prob.domain.centroids()
    sol = gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=5, exponent=2)), engine=OracleEngine))
    mu, sd, _ = E.idw(x[keep], z[keep], grid, 5, exponent=2)
    assert sol.names()[:2] == ["z", "z_distance"]                            # idw.jl:148-149
    assert np.allclose(sol["z"], mu) and np.allclose(sol["z_distance"], sd)
    sol = gss.solve(prob, gss.IDWSolver(engine=OracleEngine))                        # all samples (idw.jl:93)
    mu, sd, _ = E.idw(x[keep], z[keep], grid)
    assert np.allclose(sol["z"], mu)
    sol = gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=9)), engine=OracleEngine))
    mu, var, _ = E.lwr(x[keep], z[keep], grid, 9)
    assert sol.names()[:2] == ["z", "z_variance"]                            # lwr.jl:153-154
    assert np.allclose(sol["z"], mu) and np.allclose(sol["z_variance"], var)
    sol = gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=9, weightfun=gss.TricubeWeight(),
                                                    neighborhood=gss.MetricBall(4.0), minneighbors=4)),
                                        engine=OracleEngine))
    mu, var, st = E.lwr(x[keep], z[keep], grid, 9, 4, E.tricube, radius=4.0)
    assert np.array_equal(np.isnan(sol["z"]), st != 0)
    assert np.allclose(sol["z"][st == 0], mu[st == 0])


def test_idw_lwr_parameter_errors():
    prob, _, _ = _problem(missing=False)
    with pytest.raises(AssertionError, match="exponent must be positive"):
        gss.solve(prob, gss.IDWSolver(("z", dict(exponent=0)), engine=OracleEngine))
    with pytest.raises(AssertionError, match="invalid min/max number of neighbors"):
        gss.solve(prob, gss.LWRSolver(("z", dict(minneighbors=5, maxneighbors=3)), engine=OracleEngine))
    with pytest.raises(NotImplementedError):
        gss.solve(prob, gss.LWRSolver(("z", dict(weightfun=lambda h: 1 - h)), engine=OracleEngine))
    with pytest.raises(NotImplementedError):
        gss.solve(prob, gss.IDWSolver(("z", dict(distance="minkowski")), engine=OracleEngine))
    with pytest.raises(ValueError, match="positive radius"):
        gss.solve(prob, gss.IDWSolver(("z", dict(distance="haversine")), engine=OracleEngine))
    with pytest.raises(ValueError):
        gss.IDWSolver(("z", dict(variogram=None)))
    with pytest.warns(UserWarning, match="Invalid maximum number of neighbors"):
        gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=10 ** 6)), engine=OracleEngine))


def test_search_distances_of_the_reference_tests():
    """`distance` (idw.jl:54, lwr.jl:57; test/estimation/idw.jl:23-29 uses Haversine(1.0)): closed forms of the
    metric keys and the solvers running on a longitude / latitude grid through the stand-in engine."""
    from oracle.kriging import knn_search, metric_dist, metric_key
    pts = np.array([[0.0, 0.0], [90.0, 0.0], [0.0, 90.0], [180.0, 0.0], [10.0, -20.0]])
    key = metric_key(pts, pts[0], ("haversine", 2.0))
    d = metric_dist(key, ("haversine", 2.0))
    assert np.allclose(d[:4], [0.0, np.pi, np.pi, 2 * np.pi], atol=1e-12)         # quarter / half great circles, r = 2
    assert np.allclose(metric_key(pts, pts[4], "cityblock"), [30, 100, 120, 190, 0])
    assert np.allclose(metric_key(pts, pts[4], "chebyshev"), [20, 80, 110, 170, 0])
    # the three metrics rank differently
    q = np.array([[50.0, 40.0]])
    ranks = {m: tuple(knn_search(pts, q, 5, distance=m)[0][0]) for m in (None, "cityblock", "chebyshev", ("haversine", 1.0))}
    assert len(set(ranks.values())) >= 2
    data = gss.georef(dict(z=[4.0, -1.0, 3.0]), [(50.0, -30.0), (100.0, 30.0), (200.0, 10.0)])
    dom = gss.CartesianGrid((60, 30), (1.0, -89.0), (358.0 / 60, 178.0 / 30))
    prob = gss.EstimationProblem(data, dom, "z")
    sol = gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=3, distance=("haversine", 1.0))), engine=OracleEngine))
    mu, sd, _ = E.idw(np.array([(50.0, -30.0), (100.0, 30.0), (200.0, 10.0)]), np.array([4.0, -1.0, 3.0]),
                      dom.centroids(), 3, distance=("haversine", 1.0))
    assert np.allclose(sol["z"], mu) and np.allclose(sol["z_distance"], sd)
    assert sol["z"].min() >= -1.0 and sol["z"].max() <= 4.0 and sd.max() <= np.pi
    # a neighbourhood overrides the metric (searcher_ui, ui.jl:25-31)
    sol2 = gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=3, distance="chebyshev",
                                                     neighborhood=gss.MetricBall(500.0))), engine=OracleEngine))
    ref = E.lwr(np.array([(50.0, -30.0), (100.0, 30.0), (200.0, 10.0)]), np.array([4.0, -1.0, 3.0]), dom.centroids(), 3,
                radius=500.0)
    assert np.allclose(sol2["z"], ref[0], equal_nan=True)


def test_compositional_idw_replays_the_reference_assertions_and_known_answers():
    """test/estimation/idw.jl:47-65 -- the one numerical assertion the reference holds on IDW: three `Composition` data,
    `IDWSolver()` on a 100 x 100 grid, `aitchison(S[cell], datum) < 1e-2` at the three data cells.  The oracle follows
    idw.jl:128-141 with the compositions' own operations (powering, perturbation: `idw_compositional`); the twin maps
    the parts to logarithms, estimates them as value columns that share one search and one weight vector, and maps
    back -- the two agree to rounding because both operations are linear in the log-parts."""
    data = [gss.Composition(0.1, 0.2), gss.Composition(0.3, 0.4), gss.Composition(0.5, 0.6)]
    coord = [(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)]
    grid = gss.CartesianGrid(100, 100)
    problem = gss.EstimationProblem(gss.georef({"z": data}, coord), grid, "z")
    sol = gss.solve(problem, gss.IDWSolver(engine=OracleEngine))
    S = sol["z"]
    lin = lambda i, j: (i - 1) + 100 * (j - 1)                      # LinearIndices(size(grid))[i, j], 1-based
    assert gss.aitchison(S[lin(25, 25)], data[0]) < 1e-2
    assert gss.aitchison(S[lin(50, 75)], data[1]) < 1e-2
    assert gss.aitchison(S[lin(75, 50)], data[2]) < 1e-2
    # the restatement with the compositions' own arithmetic
    parts = np.array([c.parts for c in data])
    mu, sd, st = E.idw_compositional(np.array(coord), parts, grid.centroids())
    got = np.array([c.parts for c in S])
    assert np.max(np.abs(got / mu - 1.0)) < 1e-13 and not st.any()
    assert np.allclose(sol["z_distance"], sd)
    # known answers: (i) the clr coordinates of the estimate are the weighted means of the data's clr coordinates with
    # weights that sum to one (the estimate stays in the simplex's linear structure); (ii) midway between two data the
    # estimate is their geometric mean; (iii) a datum is reproduced at its own location (idw.jl:131-134)
    x2 = np.array([[0.0], [2.0]])
    p2 = np.array([[0.2, 0.3, 0.5], [0.6, 0.3, 0.1]])
    m2, _, _ = E.idw_compositional(x2, p2, np.array([[1.0], [0.0], [0.5]]))
    assert np.allclose(m2[0], np.sqrt(p2[0] * p2[1]), rtol=1e-15)
    assert np.array_equal(m2[1], p2[0])
    w = np.array([1 / 0.5, 1 / 1.5]) / (1 / 0.5 + 1 / 1.5)
    clr = lambda v: np.log(v) - np.log(v).mean()
    assert np.allclose(clr(m2[2]), w[0] * clr(p2[0]) + w[1] * clr(p2[1]), atol=1e-15)
    # Composition arithmetic itself (CoDa's definitions, [RECALL]): perturbation, powering, scale invariance
    a, b = gss.Composition(1.0, 2.0, 4.0), gss.Composition(2.0, 2.0, 1.0)
    assert np.allclose((a + b).parts, [2.0, 4.0, 4.0]) and np.allclose((0.5 * a).parts, [1.0, np.sqrt(2.0), 2.0])
    assert gss.aitchison(a, gss.Composition(3.0, 6.0, 12.0)) < 1e-15
    assert np.isclose(gss.aitchison(gss.Composition(0.1, 0.2), gss.Composition(0.2, 0.1)), np.sqrt(2.0) * np.log(2.0))


def test_compositions_with_missing_values_and_a_neighbourhood():
    """Missing compositions are dropped like missing numbers (idw.jl:77); estimation points without enough neighbours
    come back as `missing` (None) -- idw.jl:123-124."""
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 50, (30, 2))
    parts = rng.uniform(0.1, 1.0, (30, 3))
    data = [gss.Composition(p) for p in parts]
    data[4] = None
    dom = gss.PointSet(rng.uniform(0, 50, (40, 2)))
    prob = gss.EstimationProblem(gss.georef({"c": data}, x), dom, "c")
    sol = gss.solve(prob, gss.IDWSolver(("c", dict(maxneighbors=6, neighborhood=gss.MetricBall(12.0), exponent=2)),
                                        engine=OracleEngine))
    keep = np.arange(30) != 4
    mu, sd, st = E.idw_compositional(x[keep], parts[keep], dom.centroids(), maxneighbors=6, exponent=2.0, radius=12.0)
    assert st.any() and not st.all()
    for j, c in enumerate(sol["c"]):
        assert (c is None) == bool(st[j])
        if c is not None:
            assert np.allclose(c.parts, mu[j], rtol=1e-13)
