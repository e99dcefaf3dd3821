"""Host-side logic of the solver front-ends (no GPU): parameter surface, model / searcher choice,
missing values, error and warning behaviour of the reference (krig.jl:76-164, ui.jl, fft.jl:62-143,
lu.jl:76-169).  Arithmetic is supplied by the oracle stand-in engine in tests/oracle_engine.py."""
import warnings

import numpy as np
import pytest

import gss
from gss.engine import EDK, OK, SK, UK
from oracle_engine import OracleEngine


def test_exports_mirror_the_reference_module():        # GeoStatsSolvers.jl:46-69 (hot-path subset)
    for name in ("KrigingSolver", "FFTGS", "LUGS", "solve", "EstimationProblem", "SimulationProblem"):
        assert hasattr(gss, name)


def test_ui_dispatch_rules():                           # test/ui.jl:6-37
    dom = gss.PointSet(np.random.default_rng(0).uniform(size=(3, 2)))
    assert gss.searcher_ui(dom, 2, "euclidean", None) == ("KNearestSearch", 2)
    assert gss.searcher_ui(dom, 2, None, gss.MetricBall(1.0)) == ("KBallSearch", 2)
    assert gss.searcher_ui(dom, None, "euclidean", None) == ("KNearestSearch", 3)
    with pytest.warns(UserWarning, match=r"Invalid maximum number of neighbors. Adjusting to 3\.\.\."):
        assert gss.searcher_ui(dom, 4, "euclidean", None) == ("KNearestSearch", 3)
    g = gss.CartesianGrid(10, 10)
    v = gss.GaussianVariogram()
    assert gss.kriging_ui(g, v, None, None, None) == OK
    assert gss.kriging_ui(g, v, 0.0, None, None) == SK
    assert gss.kriging_ui(g, v, None, 2, None) == UK
    assert gss.kriging_ui(g, v, None, None, [lambda x: 1]) == EDK


def test_solver_parameter_surface():
    s = gss.KrigingSolver(("a", dict(mean=1.0)), b=dict(degree=1, variogram=gss.SphericalVariogram(range=20.0)))
    assert s.params("a")["mean"] == 1.0 and s.params("a")["variogram"].kind == "gaussian"     # krig.jl:65 default
    assert s.params("b")["degree"] == 1 and s.params("c")["minneighbors"] == 1
    with pytest.raises(ValueError):
        gss.KrigingSolver(("a", dict(nope=1)))
    l = gss.LUGS(("z", {}), ("y", {}), (("z", "y"), dict(correlation=0.7)))
    p = gss.SimulationProblem(gss.CartesianGrid(10), (("z", float), ("y", float), ("w", float)), 1)
    assert l.covariables(p) == [("z", "y"), ("w",)]
    assert gss.FFTGS().params("z")["mean"] == 0.0                                              # fft.jl:53


def test_variogram_constructors():
    v = gss.GaussianVariogram(gss.MetricBall((20.0, 5.0)))
    assert v.radii == (20.0, 5.0) and v.range == 1.0
    assert gss.SphericalVariogram(gss.MetricBall(7.0)).range == 7.0
    assert gss.MaternVariogram(range=3.0, order=1.5).nu == 1.5


def test_kriging_missing_values_and_errors():
    data = gss.georef({"z": [1.0, None, 0.0, np.nan, 1.0]}, [(25.0, 25.0), (1.0, 1.0), (50.0, 75.0), (2.0, 2.0), (75.0, 50.0)])
    grid = gss.CartesianGrid((20, 20), (0.0, 0.0), (5.0, 5.0))
    solver = gss.KrigingSolver(("z", dict(variogram=gss.GaussianVariogram(range=35.0))), engine=OracleEngine)
    pre = solver.preprocess(gss.EstimationProblem(data, grid, "z"))
    assert pre["z"]["x"].shape == (3, 2) and pre["z"]["z"].tolist() == [1.0, 0.0, 1.0]          # krig.jl:97-107
    sol = gss.solve(gss.EstimationProblem(data, grid, "z"), solver)
    assert set(sol.names()) == {"z", "z_variance"} and sol["z"].shape == (400,)                # krig.jl:160-163
    allmiss = gss.georef({"z": [np.nan, np.nan]}, [(0.0, 0.0), (1.0, 1.0)])
    with pytest.raises(AssertionError, match="all samples of z are missing, aborting..."):    # krig.jl:100-102
        gss.solve(gss.EstimationProblem(allmiss, grid, "z"), solver)


def test_kriging_variants_through_solve_match_direct_oracle():
    from oracle import kriging as K
    from oracle.variogram import Variogram
    rng = np.random.default_rng(1)
    xy = rng.uniform(0, 50, (30, 2))
    z = rng.normal(size=30)
    data = gss.georef({"z": z}, xy)
    dom = gss.PointSet(rng.uniform(0, 50, (40, 2)))
    vg = gss.ExponentialVariogram(range=20.0)
    ovg = Variogram("exponential", range=20.0)
    for params, ref in [(dict(), K.exactsolve(K.OK, ovg, xy, z, dom.coords)),
                        (dict(mean=0.5), K.exactsolve(K.SK, ovg, xy, z, dom.coords, mean=0.5)),
                        (dict(degree=1), K.exactsolve(K.UK, ovg, xy, z, dom.coords, degree=1)),
                        (dict(mean=0.5, degree=1), K.exactsolve(K.UK, ovg, xy, z, dom.coords, degree=1))]:
        sol = gss.solve(gss.EstimationProblem(data, dom, "z"),
                        gss.KrigingSolver(("z", dict(variogram=vg, **params)), engine=OracleEngine))
        assert np.allclose(sol["z"], ref[0]) and np.allclose(sol["z_variance"], ref[1])
    # moving neighbourhood with too few neighbours -> missing (NaN)                              krig.jl:213-214
    sol = gss.solve(gss.EstimationProblem(data, gss.PointSet([[500.0, 500.0]]), "z"),
                    gss.KrigingSolver(("z", dict(variogram=vg, maxneighbors=5, minneighbors=2,
                                                 neighborhood=gss.MetricBall(3.0))), engine=OracleEngine))
    assert np.isnan(sol["z"][0]) and np.isnan(sol["z_variance"][0])
    # external drift functions are evaluated on the host and shipped as values
    drifts = [lambda p: 1.0, lambda p: p[0]]
    sol = gss.solve(gss.EstimationProblem(data, dom, "z"),
                    gss.KrigingSolver(("z", dict(variogram=vg, drifts=drifts)), engine=OracleEngine))
    ref = K.exactsolve(K.UK, ovg, xy, z, dom.coords, degree=None, drift_data=None) if False else None
    fd = np.stack([[f(c) for f in drifts] for c in xy])
    f0 = np.stack([[f(c) for f in drifts] for c in dom.coords])
    mu, var = K.exactsolve(K.EDK, ovg, xy, z, dom.coords, drift_data=fd, drift_dom=f0)
    assert np.allclose(sol["z"], mu) and np.allclose(sol["z_variance"], var)


def test_unsupported_options_fail_loudly():
    data = gss.georef({"z": [1.0, 0.0]}, [(0.0, 0.0), (1.0, 1.0)])
    grid = gss.CartesianGrid(4, 4)
    for bad in (dict(distance="minkowski"), dict(path="source")):     # unknown named paths need an explicit order
        with pytest.raises(NotImplementedError):
            gss.solve(gss.EstimationProblem(data, grid, "z"), gss.KrigingSolver(("z", bad), engine=OracleEngine))
    with pytest.raises(ValueError, match="Cartesian grids"):                                    # fft.jl:40-42
        gss.solve(gss.SimulationProblem(gss.PointSet(np.zeros((4, 2))), ("z", float), 1), gss.FFTGS(engine=OracleEngine))


def test_fftgs_and_lugs_through_solve_with_stand_in():
    grid = gss.CartesianGrid(20, 16)
    vgrid = gss.view(grid, range(0, 100))
    sol = gss.solve(gss.SimulationProblem(vgrid, ("z", float), 3),
                    gss.FFTGS(("z", dict(variogram=gss.ExponentialVariogram(range=5.0))), rng=5, engine=OracleEngine))
    assert gss.domain(sol) == vgrid and len(sol[0].z) == 100 and len(sol["z"]) == 3           # fft.jl tests :21-22
    S = gss.georef({"z": [0.0, 1.0, 0.0]}, np.array([[0.0], [10.0], [20.0]]))
    sol = gss.solve(gss.SimulationProblem(S, gss.CartesianGrid(30), "z", 2),
                    gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=6.0))), rng=3, engine=OracleEngine))
    assert len(sol) == 2 and sol[0].z.shape == (30,)
    # custom factorization (test/simulation/lu.jl:66-76): `lu` and `cholesky` both run; anything else is refused
    s_lu = gss.solve(gss.SimulationProblem(S, gss.CartesianGrid(30), "z", 1),
                     gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=6.0), factorization="lu")), rng=3,
                              engine=OracleEngine))
    assert np.all(np.isfinite(s_lu[0].z)) and not np.array_equal(s_lu[0].z, sol[0].z)
    with pytest.raises(ValueError, match="factorization"):
        gss.solve(gss.SimulationProblem(gss.CartesianGrid(10), ("z", float), 1),
                  gss.LUGS(("z", dict(factorization="qr")), engine=OracleEngine))
    with pytest.raises(AssertionError):                                                        # lu.jl:96
        l3 = gss.LUGS((("a", "b", "c"), {}), engine=OracleEngine)
        gss.solve(gss.SimulationProblem(gss.CartesianGrid(5), (("a", float), ("b", float), ("c", float)), 1), l3)


def test_product_engine_refuses_to_run_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present")
    from gss import _lib
    data = gss.georef({"z": [1.0, 0.0]}, [(0.0, 0.0), (1.0, 1.0)])
    with pytest.raises(_lib.GSSError, match="no CPU fallback"):
        gss.solve(gss.EstimationProblem(data, gss.CartesianGrid(4, 4), "z"), gss.KrigingSolver())


def test_custom_paths_return_results_in_traversal_order():
    """`path` (krig.jl:73, idw.jl:55): estimation is per-point independent, but the reference stores the results in
    the order the path visits the domain (krig.jl:179-183) -- so does the twin."""
    rng = np.random.default_rng(4)
    data = gss.georef({"z": rng.normal(size=12)}, rng.uniform(0, 8, (12, 2)))
    grid = gss.CartesianGrid(8, 6)
    prob = gss.EstimationProblem(data, grid, "z")
    vg = gss.ExponentialVariogram(range=5.0)
    base = gss.solve(prob, gss.KrigingSolver(("z", dict(variogram=vg, maxneighbors=5)), engine=OracleEngine))
    order = np.random.default_rng(9).permutation(48)
    for path, perm in ((order, order), (("random", 9), order)):
        sol = gss.solve(prob, gss.KrigingSolver(("z", dict(variogram=vg, maxneighbors=5, path=path)), engine=OracleEngine))
        assert np.array_equal(sol["z"], base["z"][perm]) and np.array_equal(sol["z_variance"], base["z_variance"][perm])
    ib = gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=4)), engine=OracleEngine))
    ip = gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=4, path=order)), engine=OracleEngine))
    assert np.array_equal(ip["z"], ib["z"][order]) and np.array_equal(ip["z_distance"], ib["z_distance"][order])
    with pytest.raises(ValueError, match="permutation"):
        gss.solve(prob, gss.LWRSolver(("z", dict(path=np.zeros(48, dtype=int))), engine=OracleEngine))


def test_explicit_init_places_the_data_where_told():
    """`init = ExplicitInit(orig, dest)` ([DEP] GeoStatsBase, initbuff at lu.jl:86 / seq.jl:85): row orig[i] of the
    data goes to cell dest[i], wherever its coordinates are; NearestInit stays the default; missing values are skipped."""
    data = gss.georef({"z": [0.0, np.nan, 2.0, 3.0]}, np.array([[0.1], [5.2], [9.2], [50.3]]))
    grid = gss.CartesianGrid(60)
    init = ("explicit", [3, 1, 0], [5, 6, 40])      # datum 3 -> cell 5, datum 1 (missing) -> nowhere, datum 0 -> cell 40
    vg = gss.SphericalVariogram(range=8.0)
    for solver in (gss.LUGS(("z", dict(variogram=vg)), engine=OracleEngine, init=init, rng=1),
                   gss.SGS(("z", dict(variogram=vg, maxneighbors=6)), engine=OracleEngine, init=init, rng=1)):
        sol = gss.solve(gss.SimulationProblem(data, grid, "z", 2), solver)
        for r in sol["z"]:
            assert r[5] == 3.0 and r[40] == 0.0 and r[6] != 2.0 and np.all(np.isfinite(r))
    near = gss.solve(gss.SimulationProblem(data, grid, "z", 1), gss.SGS(("z", dict(variogram=vg)), engine=OracleEngine, rng=1))
    assert near["z"][0][0] == 0.0 and near["z"][0][9] == 2.0 and near["z"][0][50] == 3.0      # NearestInit
    with pytest.raises(NotImplementedError):
        gss.solve(gss.SimulationProblem(data, grid, "z", 1), gss.LUGS(("z", dict(variogram=vg)), engine=OracleEngine,
                                                                      init="random"))
    with pytest.raises(ValueError, match="outside"):
        gss.solve(gss.SimulationProblem(data, grid, "z", 1), gss.SGS(("z", dict(variogram=vg)), engine=OracleEngine,
                                                                     init=("explicit", [0], [60])))


def test_nearest_cell_of_a_cartesian_grid_is_arithmetic_and_follows_the_search_rule():
    """NearestInit / fft.jl:129-132 on a full CartesianGrid: the containing cell or a neighbour along an axis, chosen by
    the search's rule (squared distance in dimension order, ties to the lower index) -- equal to the exhaustive search
    over all centroids, also for data on cell boundaries (integer coordinates on a unit grid: ties), outside the grid,
    with an origin and anisotropic spacing; views go through the search."""
    import gss
    from gss import solvers as S
    from gss.geo import DomainView
    from oracle_engine import OracleEngine
    rng = np.random.default_rng(3)
    for dims, origin, spacing in (((7, 5), (0.0, 0.0), (1.0, 1.0)), ((6, 4, 5), (-2.0, 3.0, 0.5), (0.5, 2.0, 1.25)),
                                  ((9,), (1.0,), (0.25,))):
        g = gss.CartesianGrid(dims, origin, spacing)
        d = len(dims)
        lo = np.asarray(origin)
        hi = lo + np.asarray(dims) * np.asarray(spacing)
        x = np.vstack([rng.uniform(lo - 1.0, hi + 1.0, (200, d)),
                       lo + np.asarray(spacing) * rng.integers(0, np.asarray(dims) + 1, (100, d)),   # cell corners: ties
                       lo + np.asarray(spacing) * (rng.integers(0, np.asarray(dims), (50, d)) + 0.5)])   # centroids
        cent = g.centroids()
        acc = np.zeros((x.shape[0], cent.shape[0]))
        for a in range(d):
            t = cent[None, :, a] - x[:, None, a]
            acc = acc + t * t
        ref = np.argmin(acc, axis=1)                     # first minimum = lowest index among ties
        got = S._nearest_cells(None, g, None, x)         # no engine needed
        assert np.array_equal(got, ref)
        assert np.array_equal(S._nearest_cells(OracleEngine, g, cent, x[:0]), np.empty(0, dtype=np.int64))
    g = gss.CartesianGrid((8, 6))
    view = DomainView(g, np.array([3, 10, 11, 40]))
    got = S._nearest_cells(OracleEngine, view, view.centroids(), np.array([[3.4, 0.2], [2.6, 1.4]]))
    assert np.array_equal(got, [0, 1])                   # positions inside the view


def test_variables_that_share_samples_and_model_share_the_kriging_system():
    """Several variables on the same samples with the same variogram object and a global neighbourhood: the first is fitted
    and predicted, the others reuse its weights (one batched product) and its variances -- same numbers as a solve per
    variable (krig.jl:141-161 repeats everything per variable).  A variable with a missing value, or with a model of its
    own, does not share."""
    rng = np.random.default_rng(5)
    xy = rng.uniform(0, 50, (40, 2))
    tab = {"a": rng.normal(size=40), "b": rng.normal(size=40) + 2.0, "c": rng.normal(size=40)}
    tab["c"][7] = np.nan                                                # its samples differ: fitted on its own
    data = gss.georef(tab, xy)
    dom = gss.PointSet(rng.uniform(0, 50, (60, 2)))
    vg = gss.SphericalVariogram(range=20.0, nugget=0.1)
    other = gss.SphericalVariogram(range=20.0, nugget=0.1)
    for extra in (dict(), dict(mean=0.5), dict(degree=1)):
        counted = []

        class Counting(OracleEngine):
            class Krig(OracleEngine.Krig):
                def __init__(self, *a, **k):
                    counted.append(1)
                    super().__init__(*a, **k)

        together = gss.solve(gss.EstimationProblem(data, dom, ("a", "b", "c")),
                             gss.KrigingSolver(("a", dict(variogram=vg, **extra)), ("b", dict(variogram=vg, **extra)),
                                               ("c", dict(variogram=vg, **extra)), engine=Counting))
        assert len(counted) == 2                                        # a (shared with b) and c
        for v in ("a", "b", "c"):
            alone = gss.solve(gss.EstimationProblem(data, dom, v),
                              gss.KrigingSolver((v, dict(variogram=vg, **extra)), engine=OracleEngine))
            assert np.max(np.abs(together[v] - alone[v])) < 1e-10
            assert np.max(np.abs(together[f"{v}_variance"] - alone[f"{v}_variance"])) < 1e-10
        counted.clear()
        gss.solve(gss.EstimationProblem(data, dom, ("a", "b")),
                  gss.KrigingSolver(("a", dict(variogram=vg, **extra)), ("b", dict(variogram=other, **extra)), engine=Counting))
        assert len(counted) == 2                                        # two model objects: two systems


def test_models_with_a_nugget_beyond_the_sill_are_refused():
    """A nugget above the sill makes the structured part of gamma negative; inside a nested model the device would have
    dropped such a structure and kept its nugget (found by tools/hunt_covariance.py).  Refused at construction; a Gaussian
    structure with nugget = sill is refused when the nested model is bound (its regularised nugget exceeds the sill)."""
    with pytest.raises(ValueError, match="nugget"):
        gss.ExponentialVariogram(sill=0.3, nugget=0.4)
    with pytest.raises(ValueError, match="sill"):
        gss.SphericalVariogram(sill=0.0)
    ok = gss.SphericalVariogram(sill=1.0, nugget=1.0)                     # a pure nugget is a valid structure
    assert ok.sill == ok.nugget
    from gss.engine import _vg_struct
    nested = 0.5 * gss.GaussianVariogram(sill=1.0, nugget=1.0) + gss.ExponentialVariogram(range=3.0)
    with pytest.raises(ValueError, match="exceeds its sill"):
        _vg_struct(nested, 2)
