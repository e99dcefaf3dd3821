"""IDW / LWR on the device (gss_idw_predict / gss_lwr_predict) vs the oracle restatement of idw.jl:111-142 and
lwr.jl:114-147.  Tolerance: 1e-10 relative-or-absolute on means and the auxiliary column (sums are reduced
in a different order on the device; LWR solves about the estimation point), statuses bit-exact."""
import numpy as np
import pytest

from oracle import idw_lwr as E

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-10):
    a, b = np.asarray(a), np.asarray(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    return np.all(np.abs(a[ok] - b[ok]) <= tol * (1.0 + np.abs(b[ok])))


def _data(n, m, dim, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 100, (n, dim))
    z = np.sin(x[:, 0] / 17.0) + 0.01 * x.sum(axis=1) + 0.1 * rng.normal(size=n)
    c = rng.uniform(-5, 105, (m, dim))
    c[:4] = x[:4]                                   # zero distances (idw.jl:131-134)
    return x, z, c


@pytest.mark.parametrize("n,m,dim,k", [(500, 700, 2, 8), (3000, 900, 3, 64), (40, 300, 1, 5), (64, 200, 3, 64),
                                       (2000, 500, 3, None), (70, 129, 2, None), (1500, 300, 1, None)])
@pytest.mark.parametrize("exponent", [1, 2, 2.5])
def test_idw_matches_oracle(n, m, dim, k, exponent):
    from gss.engine import HipEngine
    x, z, c = _data(n, m, dim, n + dim)
    mu, sd, st = HipEngine.idw(x, z, c, n if k is None else k, 1, exponent)
    rmu, rsd, rst = E.idw(x, z, c, k, 1, exponent)
    assert np.array_equal(st, rst) and _close(mu, rmu) and _close(sd, rsd)
    assert np.array_equal(mu[:4], z[:4]) and np.all(sd[:4] == 0.0)


@pytest.mark.parametrize("n,m,dim,k", [(500, 700, 2, 8), (3000, 900, 3, 64), (40, 300, 1, 5), (64, 200, 3, 64),
                                       (2000, 500, 3, None), (70, 129, 2, None), (1500, 300, 1, None)])
@pytest.mark.parametrize("weight", ["default", "tricube", "exp1"])
def test_lwr_matches_oracle(n, m, dim, k, weight):
    from gss.engine import HipEngine
    x, z, c = _data(n, m, dim, 3 * n + dim)
    spec, wf = dict(default=((0, 3.0, 2.0), E.default_weightfun), tricube=((1, 0.0, 0.0), E.tricube),
                    exp1=((0, 2.0, 1.0), E.exp_weight(2.0, 1.0)))[weight]
    if weight == "tricube" and (k is not None and k <= dim + 1):
        pytest.skip("tricube zeroes the farthest neighbour: fewer than d+1 effective points")
    mu, var, st = HipEngine.lwr(x, z, c, n if k is None else k, 1, spec)
    rmu, rvar, rst = E.lwr(x, z, c, k, 1, wf)
    assert np.array_equal(st, rst) and not st.any()
    assert _close(mu, rmu, 1e-9) and _close(var, rvar, 1e-9)


def test_balls_minneighbors_and_singular_points():
    from gss.engine import HipEngine
    x, z, c = _data(800, 600, 2, 11)
    for kw in (dict(radius=6.0), dict(radii=(9.0, 4.0))):
        for k in (12, None):
            kk = 800 if k is None else k
            mu, sd, st = HipEngine.idw(x, z, c, kk, 3, 1.0, **kw)
            rmu, rsd, rst = E.idw(x, z, c, k, 3, 1.0, **kw)
            assert np.array_equal(st, rst) and (st == 1).any() and (st == 0).any()
            assert _close(mu, rmu) and _close(sd, rsd)
            mu, var, st = HipEngine.lwr(x, z, c, kk, 4, (0, 3.0, 2.0), **kw)
            rmu, rvar, rst = E.lwr(x, z, c, k, 4, E.default_weightfun, **kw)
            assert np.array_equal(st == 1, rst == 1)
            ok = (st == 0) & (rst == 0)
            assert ok.sum() > 100 and _close(mu[ok], rmu[ok], 1e-8) and _close(var[ok], rvar[ok], 1e-8)
    # one neighbour cannot carry a line: GSS_PT_SINGULAR, NaN outputs (the reference's `\` throws)
    mu, var, st = HipEngine.lwr(x, z, c[10:20], 1, 1, (0, 3.0, 2.0))
    assert np.all(st == 2) and np.all(np.isnan(mu))


def test_linear_field_and_convexity_properties_at_scale():
    """Size-independent properties on 2e5 points x 5e4 samples: LWR reproduces a linear field; IDW stays inside the
    data range and interpolates the samples."""
    from gss.engine import HipEngine
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 1000, (50_000, 3))
    z = 2.0 + 0.01 * x[:, 0] - 0.02 * x[:, 1] + 0.005 * x[:, 2]
    c = rng.uniform(0, 1000, (200_000, 3))
    mu, var, st = HipEngine.lwr(x, z, c, 32)
    assert not st.any()
    assert np.max(np.abs(mu - (2.0 + 0.01 * c[:, 0] - 0.02 * c[:, 1] + 0.005 * c[:, 2]))) < 1e-8
    zi = rng.normal(size=50_000)
    mu, sd, st = HipEngine.idw(x, zi, np.concatenate([c[:1000], x[:1000]]), 16, 1, 2.0)
    assert mu.min() >= zi.min() and mu.max() <= zi.max()
    assert np.array_equal(mu[1000:], zi[:1000]) and np.all(sd[1000:] == 0) and np.all(sd[:1000] > 0)


def test_solvers_through_solve_and_argument_errors():
    import gss
    from gss import _lib
    rng = np.random.default_rng(2)
    xs = rng.uniform(0, 20, (50, 2))
    zs = np.cos(xs[:, 0] / 3.0) + 0.05 * xs[:, 1]
    zs[5] = np.nan
    prob = gss.EstimationProblem(gss.georef(dict(z=zs), xs), gss.CartesianGrid(20, 20), "z")
    keep = ~np.isnan(zs)
    grid = prob.domain.centroids()
    sol = gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=6))))
    rmu, rsd, _ = E.idw(xs[keep], zs[keep], grid, 6)
    assert sol.names() == ["z", "z_distance"] and _close(sol["z"], rmu) and _close(sol["z_distance"], rsd)
    sol = gss.solve(prob, gss.LWRSolver())
    rmu, rvar, _ = E.lwr(xs[keep], zs[keep], grid)
    assert sol.names() == ["z", "z_variance"] and _close(sol["z"], rmu, 1e-9) and _close(sol["z_variance"], rvar, 1e-9)
    from gss.engine import HipEngine
    with pytest.raises(_lib.GSSError, match="exponent must be positive"):
        HipEngine.idw(xs[keep], zs[keep], grid, 5, 1, 0.0)


METRICS = ["cityblock", "chebyshev", ("haversine", 6371.0)]


@pytest.mark.parametrize("distance", METRICS)
def test_search_metrics_indices_and_estimators(distance):
    """`distance` parameter (krig.jl:72, idw.jl:54, lwr.jl:57): neighbour indices bit-exact under the metric's key,
    IDW / LWR / moving-neighbourhood kriging on those neighbours within the usual tolerances."""
    from gss.engine import HipEngine, KrigHandle, OK
    import gss
    from oracle import kriging as K
    from oracle.variogram import Variogram
    rng = np.random.default_rng(21)
    if distance[0] == "haversine":
        x = np.c_[rng.uniform(-180, 180, 700), rng.uniform(-80, 80, 700)]
        c = np.c_[rng.uniform(-180, 180, 400), rng.uniform(-85, 85, 400)]
    else:
        x = rng.uniform(0, 100, (700, 3))
        c = rng.uniform(0, 100, (400, 3))
    c[:3] = x[:3]
    z = np.sin(x[:, 0] / 20.0) + 0.01 * x[:, 1]
    for k in (1, 7, 64, 100, 200):   # beyond 64: passes of 64 (indexed search, or the exhaustive kernel for haversine)
        idx, cnt = HipEngine.knn_search(x, c, k, distance=distance)
        ridx, rcnt = K.knn_search(x, c, k, distance=distance)
        assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt)
    for k in (9, 150, None):
        kk = 700 if k is None else k
        mu, sd, st = HipEngine.idw(x, z, c, kk, 1, 2.0, distance=distance)
        rmu, rsd, rst = E.idw(x, z, c, k, 1, 2.0, distance=distance)
        assert np.array_equal(st, rst) and _close(mu, rmu) and _close(sd, rsd, 1e-9)
        mu, var, st = HipEngine.lwr(x, z, c, kk, 1, (0, 3.0, 2.0), distance=distance)
        rmu, rvar, rst = E.lwr(x, z, c, k, 1, E.default_weightfun, distance=distance)
        assert np.array_equal(st, rst) and _close(mu, rmu, 1e-8) and _close(var, rvar, 1e-8)
    if distance[0] != "haversine":
        h = KrigHandle(gss.ExponentialVariogram(range=30.0), OK, x, z, factor=False)
        mu, var, st, idx, cnt = h.predict_knn(c, 12, return_idx=True, distance=distance)
        r = K.approxsolve(K.OK, Variogram("exponential", range=30.0), x, z, c, 12, return_idx=True, distance=distance)
        assert np.array_equal(idx, r[3]) and _close(mu, r[0], 1e-9) and _close(var, r[1], 1e-9)


def test_reference_haversine_cases_and_metric_errors():
    """test/estimation/idw.jl:23-29 (Haversine(1.0), 3 data on a lon/lat grid) through solve; a ball cannot be
    combined with a metric at the C-ABI (searcher_ui never does, ui.jl:25-31)."""
    import gss
    from gss import _lib
    from gss.engine import HipEngine
    xs = np.array([(50.0, -30.0), (100.0, 30.0), (200.0, 10.0)])
    zs = np.array([4.0, -1.0, 3.0])
    dom = gss.CartesianGrid((200, 100), (1.0, -89.0), (358.0 / 200, 178.0 / 100))
    prob = gss.EstimationProblem(gss.georef(dict(z=zs), xs), dom, "z")
    sol = gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=3, distance=("haversine", 1.0)))))
    mu, sd, _ = E.idw(xs, zs, dom.centroids()[::37], 3, distance=("haversine", 1.0))
    assert _close(sol["z"][::37], mu) and _close(sol["z_distance"][::37], sd, 1e-9)
    sol = gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=3, distance=("haversine", 6371.0)))))
    assert np.isfinite(sol["z"]).sum() > 0.9 * sol["z"].size
    with pytest.raises(_lib.GSSError, match="cannot be combined"):
        l = _lib.lib()
        from gss.engine import ptr, check, MEM_HOST, current_stream
        idx = np.empty((3, 2), dtype=np.int32)
        cnt = np.empty(3, dtype=np.int32)
        check(l.gss_knn_search(ptr(xs), 3, 2, ptr(xs), 3, 2, 5.0, None, 1, 0.0, ptr(idx), ptr(cnt), MEM_HOST,
                               current_stream()))
    with pytest.raises(_lib.GSSError, match="longitude"):
        HipEngine.knn_search(np.zeros((4, 3)), np.zeros((2, 3)), 2, distance=("haversine", 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("distance", ["cityblock", "chebyshev"])
@pytest.mark.parametrize("n,dim,k", [(30000, 3, 20), (5000, 2, 64), (20000, 1, 7)])
def test_metric_search_with_box_bounds_equals_exhaustive_search(monkeypatch, distance, n, dim, k):
    """Cityblock and Chebyshev searches use the k-d ordered index with per-metric box bounds; the lists equal the
    exhaustive kernel's (GSS_KNN_BRUTE=1) and the oracle's on a subset, ties on a coarse lattice included."""
    from gss.engine import HipEngine
    from oracle import kriging as K
    rng = np.random.default_rng(n + k)
    x = np.round(rng.uniform(0, 60, (n, dim)), 1 if dim > 1 else 3)
    c = np.vstack([x[rng.integers(0, n, 2500)] + rng.normal(0, 0.3, (2500, dim)), rng.uniform(-10, 70, (2500, dim))])
    idx, cnt = HipEngine.knn_search(x, c, k, distance=distance)
    monkeypatch.setenv("GSS_KNN_BRUTE", "1")
    bidx, bcnt = HipEngine.knn_search(x, c, k, distance=distance)
    monkeypatch.delenv("GSS_KNN_BRUTE")
    assert np.array_equal(idx, bidx) and np.array_equal(cnt, bcnt)
    ridx, rcnt = K.knn_search(x, c[::50], k, distance=distance)
    assert np.array_equal(idx[::50], ridx) and np.array_equal(cnt[::50], rcnt)


def test_host_arrays_in_pieces_equal_device_arrays_idw_lwr():
    """Host domain arrays beyond 131 072 points travel piece by piece beside the computation (HostPipe): same
    results as device arrays, bit for bit."""
    import torch
    from gss.engine import HipEngine
    rng = np.random.default_rng(22)
    x = rng.uniform(0, 100, (3000, 2))
    z = rng.normal(size=3000)
    m = 131072 * 3 + 5
    x0 = rng.uniform(0, 100, (m, 2))
    for fn, kw in ((HipEngine.idw, dict(exponent=2.0)), (HipEngine.lwr, {})):
        dev = fn(x, z, torch.as_tensor(x0, device="cuda"), 12, **kw)
        host = fn(x, z, x0, 12, **kw)
        for a, b in zip(dev, host):
            assert isinstance(b, np.ndarray) and np.array_equal(a.cpu().numpy(), b, equal_nan=True)


@pytest.mark.parametrize("dim,k,ball", [(2, 12, None), (3, 40, None), (2, 30, 14.0), (3, 90, None)])
def test_lwr_with_an_arbitrary_weight_function(dim, k, ball):
    """`weightfun` may be any function of the normalised distance (lwr.jl:58,136).  A closure cannot cross the C-ABI: the
    device searches, the host evaluates delta and the weights, the device solves the normal equations
    (gss_lwr_predict_weights).  Against the oracle, 1e-9; the default weight through this route equals the in-kernel one."""
    import gss
    rng = np.random.default_rng(80 + dim + k)
    x = rng.uniform(0, 60, (500, dim))
    z = np.sin(x[:, 0] / 9.0) + 0.03 * x[:, -1] + 0.05 * rng.normal(size=500)
    c = rng.uniform(0, 60, (300, dim))
    wf = lambda h: 1.0 / (1.0 + 5.0 * h) ** 2 + 0.1 * np.cos(h)          # noqa: E731
    prob = gss.EstimationProblem(gss.georef({"z": z}, x), gss.PointSet(c), "z")
    kw = dict(maxneighbors=k, weightfun=wf)
    if ball is not None:
        kw["neighborhood"] = gss.MetricBall(ball)
    sol = gss.solve(prob, gss.LWRSolver(("z", kw)))
    rmu, rvar, rst = E.lwr(x, z, c, k, 1, wf, radius=ball)
    ok = rst == 0
    assert np.array_equal(np.isnan(sol["z"]), ~ok)
    assert np.max(np.abs(sol["z"][ok] - rmu[ok])) < 1e-9 and np.max(np.abs(sol["z_variance"][ok] - rvar[ok])) < 1e-9
    if ball is None:
        a = gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=k, weightfun=E.default_weightfun))))
        b = gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=k))))
        assert np.max(np.abs(a["z"] - b["z"])) < 1e-10 and np.max(np.abs(a["z_variance"] - b["z_variance"])) < 1e-10


def test_lwr_weight_function_that_only_takes_scalars():
    """lwr.jl:136 broadcasts `weightfun` over the neighbours' normalised distances, so a function written for one number
    (a branch, `math.exp`) is a legal `weightfun`: the host maps it over the neighbours that exist -- never over padding
    entries of a ball search -- and the result equals the vectorised form of the same function (1e-12)."""
    import math
    import gss
    rng = np.random.default_rng(17)
    x = rng.uniform(0, 60, (400, 2))
    z = np.cos(x[:, 0] / 7.0) + 0.02 * x[:, 1]
    c = rng.uniform(0, 60, (150, 2))
    prob = gss.EstimationProblem(gss.georef({"z": z}, x), gss.PointSet(c), "z")
    scalar = lambda h: math.exp(-3.0 * h * h) if h < 0.95 else 0.01          # noqa: E731
    vector = lambda h: np.where(h < 0.95, np.exp(-3.0 * h * h), 0.01)        # noqa: E731
    for kw in (dict(maxneighbors=15), dict(maxneighbors=25, neighborhood=gss.MetricBall(9.0))):
        a = gss.solve(prob, gss.LWRSolver(("z", dict(weightfun=scalar, **kw))))
        b = gss.solve(prob, gss.LWRSolver(("z", dict(weightfun=vector, **kw))))
        assert np.array_equal(np.isnan(a["z"]), np.isnan(b["z"]))         # (math.exp and np.exp may differ in the last bit)
        ok = np.isfinite(a["z"])
        assert ok.sum() > 100
        assert np.max(np.abs(a["z"][ok] - b["z"][ok])) < 1e-12
        assert np.max(np.abs(a["z_variance"][ok] - b["z_variance"][ok])) < 1e-12


def test_compositional_idw_on_the_device_replays_the_reference_assertions():
    """test/estimation/idw.jl:47-65 through `gss.solve` on the device: the three `aitchison(...) < 1e-2` assertions, and
    the whole field against the oracle's restatement in the compositions' own arithmetic (1e-12 relative).  The log-parts
    travel as value columns of ONE gss_idw_predict_cols call: one search, one weight vector per cell."""
    import gss
    data = [gss.Composition(0.1, 0.2), gss.Composition(0.3, 0.4), gss.Composition(0.5, 0.6)]
    coord = [(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)]
    grid = gss.CartesianGrid(100, 100)
    problem = gss.EstimationProblem(gss.georef({"z": data}, coord), grid, "z")
    sol = gss.solve(problem, gss.IDWSolver())
    S = sol["z"]
    lin = lambda i, j: (i - 1) + 100 * (j - 1)                      # noqa: E731  LinearIndices(size(grid))[i, j]
    assert gss.aitchison(S[lin(25, 25)], data[0]) < 1e-2
    assert gss.aitchison(S[lin(50, 75)], data[1]) < 1e-2
    assert gss.aitchison(S[lin(75, 50)], data[2]) < 1e-2
    mu, sd, st = E.idw_compositional(np.array(coord), np.array([c.parts for c in data]), grid.centroids())
    got = np.array([c.parts for c in S])
    assert np.max(np.abs(got / mu - 1.0)) < 1e-12 and np.max(np.abs(sol["z_distance"] - sd)) < 1e-12
    # a larger case with a neighbourhood, missing data and points without neighbours
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 50, (300, 2))
    parts = rng.uniform(0.1, 1.0, (300, 4))
    cs = [gss.Composition(p) for p in parts]
    cs[7] = None
    dom = gss.PointSet(rng.uniform(0, 50, (500, 2)))
    prob = gss.EstimationProblem(gss.georef({"c": cs}, x), dom, "c")
    sol = gss.solve(prob, gss.IDWSolver(("c", dict(maxneighbors=9, neighborhood=gss.MetricBall(3.0), exponent=2))))
    keep = np.arange(300) != 7
    mu, sd, st = E.idw_compositional(x[keep], parts[keep], dom.centroids(), maxneighbors=9, exponent=2.0, radius=3.0)
    assert st.any() and not st.all()
    for j, c in enumerate(sol["c"]):
        assert (c is None) == bool(st[j])
        if c is not None:
            assert np.max(np.abs(c.parts / mu[j] - 1.0)) < 1e-12


@pytest.mark.parametrize("k,n", [(7, 400), (40, 400), (90, 400), (400, 400), (64, 64)])
def test_value_columns_share_one_search_and_one_weight_vector(k, n):
    """gss_idw_predict_cols / gss_lwr_predict_cols: nz value columns on one search.  Every kernel family -- sixteen and
    sixty-four lanes per point, neighbour lists beyond 64, every sample a neighbour -- must give, column by column, what
    the single-column call gives (bit for bit where the same kernel serves both, 1e-9 where the single column runs on
    an instantiation of its own: the compiler contracts the sums differently), with a ball, duplicates of estimation points among the samples (zero distances)
    and device-resident arrays."""
    import torch
    from gss.engine import HipEngine
    rng = np.random.default_rng(k + n)
    x = rng.uniform(0, 40, (n, 3))
    Z = rng.normal(size=(5, n))
    c = np.concatenate([rng.uniform(0, 40, (700, 3)), x[:20]])
    for ball in (None, 9.0):
        same_kernel = k <= 64                       # the wave-per-point kernel serves one column and many alike; the
                                                    # list / all-samples kernels are instantiated per column count
        for fn, kw in ((HipEngine.idw, dict(exponent=2.0)), (HipEngine.idw, dict(exponent=1.5)),
                       (HipEngine.lwr, dict(weight=(0, 3.0, 2.0))), (HipEngine.lwr, dict(weight=(1, 0.0, 0.0)))):
            nmin = 2 if fn is HipEngine.idw else min(10, k)     # (a plane through four or five points is barely determined)
            mu, ax, st = fn(x, Z, c, k, nmin, radius=ball, **kw)
            assert mu.shape == (5, 720) and ax.shape == (720,)
            for col in range(5):
                m1, a1, s1 = fn(x, Z[col], c, k, nmin, radius=ball, **kw)
                assert np.array_equal(st, s1)
                ok = st == 0
                if same_kernel:
                    assert np.array_equal(mu[col][ok], m1[ok]) and np.array_equal(ax[ok], a1[ok])
                else:
                    # (rounding differences, amplified where a ball leaves an LWR fit nearly singular)
                    assert np.max(np.abs(mu[col][ok] - m1[ok])) < 1e-9
                    assert np.max(np.abs(ax[ok] - a1[ok]) / (1.0 + np.abs(a1[ok]))) < 1e-8
            md, ad, sd = fn(torch.as_tensor(x, device="cuda"), torch.as_tensor(Z, device="cuda"),
                            torch.as_tensor(c, device="cuda"), k, nmin, radius=ball, **kw)
            assert np.array_equal(md.cpu().numpy(), mu, equal_nan=True) and np.array_equal(sd.cpu().numpy(), st)


@pytest.mark.parametrize("solver", ["idw", "lwr"])
def test_variables_with_the_same_parameters_share_search_and_weights_through_solve(solver):
    """`solve` with three scalar variables on the same samples and the same parameters: one search, one weight vector per
    point, three value columns (one device call); a fourth variable with a missing value is estimated on its own.
    Same numbers as a solve per variable."""
    import gss
    from gss import _lib
    rng = np.random.default_rng(31)
    n, m = 3000, 20000
    xy = rng.uniform(0, 100, (n, 2))
    tab = {k: rng.normal(size=n) + i for i, k in enumerate("abcd")}
    tab["d"][5] = np.nan
    data = gss.georef(tab, xy)
    dom = gss.PointSet(rng.uniform(0, 100, (m, 2)))
    S = gss.IDWSolver if solver == "idw" else gss.LWRSolver
    params = dict(maxneighbors=12)
    _lib.profile_reset(); _lib.profile_enable(True)
    together = gss.solve(gss.EstimationProblem(data, dom, tuple("abcd")), S(*[(v, params) for v in "abcd"]))
    _lib.profile_enable(False)
    assert _lib.profile_read("knn")[1] == 2                              # {a, b, c} and d
    aux = "distance" if solver == "idw" else "variance"
    for v in "abcd":
        alone = gss.solve(gss.EstimationProblem(data, dom, v), S((v, params)))
        assert np.max(np.abs(together[v] - alone[v])) < 1e-9
        assert np.max(np.abs(together[f"{v}_{aux}"] - alone[f"{v}_{aux}"])) < 1e-9
