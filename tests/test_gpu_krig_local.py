"""Moving-neighbourhood kriging (K4 + K5) vs the oracle: neighbour indices bit-exact, means and
variances 1e-9 (well-conditioned models; Gaussian 1e-6)."""
import numpy as np
import pytest

from oracle import fftgs as offt, kriging as K
from oracle.variogram import Variogram

pytestmark = pytest.mark.gpu


def _vgs(kind, **kw):
    import gss
    ctor = dict(gaussian=gss.GaussianVariogram, exponential=gss.ExponentialVariogram,
                spherical=gss.SphericalVariogram, matern=gss.MaternVariogram)[kind]
    okw = dict(kw)
    kw = dict(kw)
    if "nu" in kw:
        kw["order"] = kw.pop("nu")
    if "radii" in kw:
        return ctor(gss.MetricBall(tuple(kw.pop("radii"))), **kw), Variogram(kind, **okw)
    return ctor(**kw), Variogram(kind, **okw)


@pytest.mark.parametrize("n,m,dim,k", [(5, 40, 1, 3), (100, 500, 2, 8), (3000, 600, 3, 64), (2049, 300, 3, 17),
                                       (64, 100, 2, 64), (5000, 257, 3, 64)])
def test_knn_indices_bit_exact(n, m, dim, k):
    from gss.engine import HipEngine
    rng = np.random.default_rng(n + k)
    x = rng.uniform(0, 100, (n, dim))
    c = rng.uniform(0, 100, (m, dim))
    c[:3] = x[:3]
    idx, cnt = HipEngine.knn_search(x, c, k)
    ridx, rcnt = K.knn_search(x, c, k)
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt)


@pytest.mark.parametrize("dim,k,ball", [(3, 64, False), (2, 10, False), (3, 24, True), (1, 7, False)])
def test_knn_pruned_equals_brute_force_on_large_clustered_data(monkeypatch, dim, k, ball):
    """The Morton-ordered, box-pruned search is an optimisation only: on 150k clustered samples with exact
    duplicates (ties) it must return the same neighbour lists as the exhaustive kernel (GSS_KNN_BRUTE=1)
    and as the float64 numpy ranking by (d2, index) on a subset."""
    from gss.engine import HipEngine
    rng = np.random.default_rng(17 + dim + k)
    n, m = 150_000, 5000          # > 4096 centres: stays on the indexed path (fewer go to the brute-force sweep)
    centres_of_mass = rng.uniform(0, 1000, (40, dim))
    x = centres_of_mass[rng.integers(0, 40, n)] + rng.normal(0, 15.0, (n, dim))
    x[1000:1200] = x[:200]                                   # exact duplicates -> ties broken by index
    c = np.concatenate([x[rng.integers(0, n, m // 2)] + rng.normal(0, 1.0, (m // 2, dim)),
                        rng.uniform(-200, 1200, (m - m // 2, dim))])  # includes centres far outside the data
    kw = dict(radius=12.0) if ball else {}
    idx, cnt = HipEngine.knn_search(x, c, k, **kw)
    monkeypatch.setenv("GSS_KNN_BRUTE", "1")
    bidx, bcnt = HipEngine.knn_search(x, c, k, **kw)
    monkeypatch.delenv("GSS_KNN_BRUTE")
    assert np.array_equal(cnt, bcnt) and np.array_equal(idx, bidx)
    ridx, rcnt = K.knn_search(x, c[::40], k, **kw)
    assert np.array_equal(idx[::40], ridx) and np.array_equal(cnt[::40], rcnt)
    if ball:
        assert cnt.min() == 0 and cnt.max() == k


@pytest.mark.parametrize("n,dim,k", [(3000, 3, 17), (70, 2, 5), (20000, 1, 9), (40000, 2, 33)])
def test_knn_index_built_on_host_and_on_device_give_the_same_neighbours(monkeypatch, n, dim, k):
    """The k-d ordering is an internal matter of the index: whether the host (nth_element) or the device (radix sort per
    level, knn_build.hip) produced it, the neighbour lists equal the oracle's.  Lattice points (ties along every axis)
    and exact duplicates included."""
    from gss.engine import HipEngine
    rng = np.random.default_rng(n + k)
    x = np.round(rng.uniform(0, 50, (n, dim)), 1)          # coarse lattice: many equal coordinates
    x[: n // 10] = x[n // 10: 2 * (n // 10)]                # duplicates
    c = rng.uniform(-5, 55, (600, dim))
    ridx, rcnt = K.knn_search(x, c[:150], k)
    for how in ("host", "device"):
        monkeypatch.setenv("GSS_KNN_BUILD", how)
        idx, cnt = HipEngine.knn_search(x, c, k)
        assert np.array_equal(idx[:150], ridx) and np.array_equal(cnt[:150], rcnt), how
        if how == "host":
            first = idx
        else:
            assert np.array_equal(idx, first)


def test_knn_few_queries_into_large_set_host_and_device():
    """m <= 4096 centres into n >= 32768 points takes the brute-force sweep (no host index build); same lists as the
    oracle ranking, and the same again when both point sets are CUDA tensors (the conditional-FFTGS cell lookup)."""
    import torch
    from gss.engine import HipEngine
    g = offt.grid_centroids((40, 40, 30))                 # 48 000 lattice cells: ties at every face / edge
    rng = np.random.default_rng(77)
    c = np.concatenate([rng.uniform(0, 40, (60, 3)), np.array([[20.0, 20.0, 15.0], [0.0, 0.0, 0.0], [39.5, 1.0, 29.5]])])
    for k in (1, 8):
        idx, cnt = HipEngine.knn_search(g, c, k)
        ridx, rcnt = K.knn_search(g, c, k)
        assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt)
        didx, dcnt = HipEngine.knn_search(torch.as_tensor(g, device="cuda"), torch.as_tensor(c, device="cuda"), k)
        assert np.array_equal(didx.cpu().numpy(), ridx) and np.array_equal(dcnt.cpu().numpy(), rcnt)
    with pytest.raises(ValueError):
        HipEngine.knn_search(torch.as_tensor(g, device="cuda"), c, 1)


def test_knn_lattice_ties_and_balls():
    from gss.engine import HipEngine
    g = offt.grid_centroids((12, 12))                     # lattice -> many exactly equal distances
    c = np.array([[6.0, 6.0], [0.0, 0.0], [5.5, 5.5], [11.9, 3.2]])
    for k in (1, 4, 9, 30):
        idx, cnt = HipEngine.knn_search(g, c, k)
        ridx, rcnt = K.knn_search(g, c, k)
        assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt)
    idx, cnt = HipEngine.knn_search(g, c, 20, radius=1.5)
    ridx, rcnt = K.knn_search(g, c, 20, radius=1.5)
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt) and cnt.max() < 20
    idx, cnt = HipEngine.knn_search(g, c, 20, radii=(3.0, 1.0))
    ridx, rcnt = K.knn_search(g, c, 20, radii=(3.0, 1.0))
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt)


CASES = [(K.OK, {}, "matern", dict(range=30.0, nu=1.5)), (K.SK, dict(mean=0.4), "exponential", dict(range=25.0)),
         (K.UK, dict(degree=1), "matern", dict(range=30.0, nu=1.5)),
         (K.UK, dict(degree=2), "spherical", dict(range=40.0, nugget=0.05))]


@pytest.mark.parametrize("variant,okw,kind,vkw", CASES)
@pytest.mark.parametrize("n,m,dim,k", [(400, 300, 3, 64), (200, 250, 2, 16), (60, 100, 1, 5)])
def test_local_kriging_matches_oracle(variant, okw, kind, vkw, n, m, dim, k):
    from gss.engine import KrigHandle
    if variant == K.UK and okw["degree"] == 2 and k < 12:
        pytest.skip("fewer neighbours than drift terms")
    gvg, ovg = _vgs(kind, **vkw)
    rng = np.random.default_rng(n * 3 + k)
    x = rng.uniform(0, 100, (n, dim))
    if dim == 1:       # uniform samples on a line contain near-duplicates (cond ~ 1e12): use a jittered lattice
        x = (np.linspace(0, 100, n) + rng.uniform(-0.3, 0.3, n))[:, None]
    z = rng.normal(size=n) + 0.02 * x[:, 0]
    x0 = rng.uniform(0, 100, (m, dim))
    x0[:2] = x[:2]
    h = KrigHandle(gvg, variant, x, z, mean=okw.get("mean"), degree=okw.get("degree"), factor=False)
    mu, var, st, idx, cnt = h.predict_knn(x0, k, return_idx=True)
    h.close()
    rmu, rvar, rst, ridx, rcnt = K.approxsolve(variant, ovg, x, z, x0, k, mean=okw.get("mean") or 0.0,
                                               degree=okw.get("degree"), return_idx=True)
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt) and np.array_equal(st, rst)
    assert np.max(np.abs(mu - rmu)) < 1e-9 * max(1.0, np.max(np.abs(rmu)))
    assert np.max(np.abs(var - rvar)) < 1e-9
    assert np.allclose(mu[:2], z[:2], atol=1e-9) or vkw.get("nugget", 0) > 0


def test_ball_search_and_minneighbors_missing():
    from gss.engine import KrigHandle
    gvg, ovg = _vgs("exponential", range=20.0)
    rng = np.random.default_rng(77)
    x = rng.uniform(0, 50, (150, 2))
    z = rng.normal(size=150)
    x0 = np.vstack([rng.uniform(0, 50, (200, 2)), [[500.0, 500.0], [80.0, 80.0]]])
    h = KrigHandle(gvg, K.OK, x, z, factor=False)
    mu, var, st, idx, cnt = h.predict_knn(x0, 10, minneighbors=3, radius=6.0, return_idx=True)
    rmu, rvar, rst, ridx, rcnt = K.approxsolve(K.OK, ovg, x, z, x0, 10, minneighbors=3, radius=6.0, return_idx=True)
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt) and np.array_equal(st, rst)
    assert st[-1] == 1 and st[-2] == 1 and np.isnan(mu[-1])               # krig.jl:213-214
    ok = st == 0
    assert ok.sum() > 50 and (~ok).sum() >= 2
    assert np.max(np.abs(mu[ok] - rmu[ok])) < 1e-9 and np.max(np.abs(var[ok] - rvar[ok])) < 1e-9
    h.close()


@pytest.mark.parametrize("variant,okw", [(K.UK, dict(degree=1)), (K.SK, dict(mean=0.3)), (K.OK, {})])
def test_mixed_neighbour_counts_inside_a_workgroup(variant, okw):
    """The moving-neighbourhood kernel works on four points per workgroup and shares the diagonal-tile factorisation
    between their waves; with a search ball over samples of very uneven density neighbouring domain points get anything
    from 0 to 64 neighbours (0 to 4 tile steps, `missing` points included) in the same workgroup, and the number of
    points is not a multiple of four."""
    from gss.engine import KrigHandle
    gvg, ovg = _vgs("matern", range=30.0, nu=1.5, nugget=0.02)
    rng = np.random.default_rng(4242)
    x = np.vstack([rng.uniform(0, 100, (600, 3)), rng.normal(50.0, 4.0, (1400, 3)), rng.normal(20.0, 1.5, (500, 3))])
    z = rng.normal(size=len(x)) + 0.01 * x[:, 0]
    x0 = rng.uniform(-10, 110, (1003, 3))
    h = KrigHandle(gvg, variant, x, z, mean=okw.get("mean"), degree=okw.get("degree"), factor=False)
    mu, var, st, idx, cnt = h.predict_knn(x0, 64, minneighbors=6, radius=9.0, return_idx=True)
    h.close()
    rmu, rvar, rst, ridx, rcnt = K.approxsolve(variant, ovg, x, z, x0, 64, mean=okw.get("mean") or 0.0,
                                               degree=okw.get("degree"), minneighbors=6, radius=9.0, return_idx=True)
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt) and np.array_equal(st, rst)
    tiles = (cnt + 15) // 16
    assert set(np.unique(tiles[st == 0])) >= {1, 2, 3, 4} and (st == 1).sum() > 50 and cnt.min() == 0
    ok = st == 0
    assert np.max(np.abs(mu[ok] - rmu[ok])) < 1e-8 and np.max(np.abs(var[ok] - rvar[ok])) < 1e-8
    assert np.all(np.isnan(mu[st != 0]))


def test_k_equals_n_reproduces_global():
    from gss.engine import KrigHandle
    gvg, ovg = _vgs("matern", range=30.0, nu=1.5)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 100, (48, 3))
    z = rng.normal(size=48)
    x0 = rng.uniform(0, 100, (300, 3))
    hg = KrigHandle(gvg, K.OK, x, z)
    mu_g, var_g, _ = hg.predict_global(x0)
    hl = KrigHandle(gvg, K.OK, x, z, factor=False)
    mu_l, var_l, st = hl.predict_knn(x0, 48)
    assert not st.any() and np.max(np.abs(mu_g - mu_l)) < 1e-9 and np.max(np.abs(var_g - var_l)) < 1e-9


def test_reference_nearest_and_local_cases_through_solve():
    """test/estimation/krig.jl:39-72 through the solver front-end (atol 1e-3 as in the reference)."""
    import gss
    data = gss.georef({"z": [1.0, 0.0, 1.0]}, [(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
    grid = gss.CartesianGrid((100, 100), (0.5, 0.5), (1.0, 1.0))
    vg = gss.GaussianVariogram(range=35.0, nugget=0.0)
    for params in (dict(variogram=vg, maxneighbors=3),
                   dict(variogram=vg, maxneighbors=3, neighborhood=gss.MetricBall(100.0))):
        sol = gss.solve(gss.EstimationProblem(data, grid, "z"), gss.KrigingSolver(("z", params)))
        Z = gss.asarray(sol, "z")
        assert abs(Z[24, 24] - 1.0) < 1e-3 and abs(Z[49, 74] - 0.0) < 1e-3 and abs(Z[74, 49] - 1.0) < 1e-3
    with pytest.warns(UserWarning, match="Invalid maximum number of neighbors. Adjusting to 3..."):
        sol = gss.solve(gss.EstimationProblem(data, grid, "z"),
                        gss.KrigingSolver(("z", dict(variogram=vg, maxneighbors=7))))
    assert np.all(np.isfinite(sol["z"]))
    # custom path (test/estimation/krig.jl:78-90): the results come back in traversal order (krig.jl:179-183)
    kw = dict(variogram=vg, maxneighbors=3, neighborhood=gss.MetricBall(100.0))
    lin = gss.solve(gss.EstimationProblem(data, grid, "z"), gss.KrigingSolver(("z", kw)))
    mg = gss.solve(gss.EstimationProblem(data, grid, "z"), gss.KrigingSolver(("z", dict(kw, path="multigrid"))))
    order = gss.solvers.multigrid_order((100, 100))
    assert np.array_equal(mg["z"], lin["z"][order]) and np.array_equal(mg["z_variance"], lin["z_variance"][order])


def test_config5_shape_small():
    """BASELINE config 5 at reduced m: UK degree 1, 5 000 3-D data, k = 64, Matern-3/2."""
    from gss.engine import KrigHandle
    gvg, ovg = _vgs("matern", range=30.0, nu=1.5)
    x = np.random.default_rng(6).uniform(0, 100, (5000, 3))
    z = 1.0 + 0.03 * x[:, 0] - 0.02 * x[:, 1] + 0.01 * x[:, 2] + np.random.default_rng(60).normal(size=5000)
    x0 = np.random.default_rng(7).uniform(0, 100, (400, 3))
    h = KrigHandle(gvg, K.UK, x, z, degree=1, factor=False)
    mu, var, st, idx, cnt = h.predict_knn(x0, 64, return_idx=True)
    rmu, rvar, rst, ridx, rcnt = K.approxsolve(K.UK, ovg, x, z, x0, 64, degree=1, return_idx=True)
    assert np.array_equal(idx, ridx) and not st.any()
    assert np.max(np.abs(mu - rmu)) < 1e-9 * max(1.0, np.max(np.abs(rmu))) and np.max(np.abs(var - rvar)) < 1e-9


def test_host_arrays_in_pieces_equal_device_arrays_moving_neighbourhood():
    """Host arrays beyond 131 072 points travel piece by piece beside the computation (HostPipe): means, variances,
    status, neighbour indices and counts equal those of one call on device arrays, bit for bit."""
    import torch
    import gss
    from gss.engine import KrigHandle, UK
    rng = np.random.default_rng(21)
    x = rng.uniform(0, 100, (2000, 3))
    z = rng.normal(size=2000)
    m = 131072 * 2 + 777
    x0 = rng.uniform(0, 100, (m, 3))
    h = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5), UK, x, z, degree=1)
    dev = h.predict_knn(torch.as_tensor(x0, device="cuda"), 24, radius=40.0, return_idx=True)
    host = h.predict_knn(x0, 24, radius=40.0, return_idx=True)
    assert len(dev) == len(host) == 5
    for a, b in zip(dev, host):
        assert isinstance(b, np.ndarray) and np.array_equal(a.cpu().numpy(), b, equal_nan=True)


@pytest.mark.parametrize("dim,k,kw", [(3, 16, {}), (3, 1, {}), (2, 4, dict(radius=3.0)), (2, 16, dict(radii=(6.0, 2.0))),
                                      (1, 9, {}), (3, 12, dict(radius=7.5))])
def test_small_neighbour_counts_equal_the_exhaustive_search(monkeypatch, dim, k, kw):
    """Up to sixteen neighbours (the IDW / LWR range): the indexed search returns the same lists, bit for bit, as the
    exhaustive kernel -- random and lattice data (ties by index), duplicates, balls and ellipsoids, queries far outside
    the data, a query count that is not a multiple of sixteen."""
    from gss.engine import HipEngine
    rng = np.random.default_rng(100 * dim + k)
    n, m = 30_000, 6007
    x = rng.uniform(0, 100, (n, dim))
    x[5000:9000] = np.round(x[5000:9000])                   # lattice points: equal distances, ties by index
    x[20000:20300] = x[:300]                                  # exact duplicates
    c = np.concatenate([rng.uniform(0, 100, (m - 2000, dim)), np.round(rng.uniform(0, 100, (1500, dim))) + 0.5,
                        rng.uniform(-300, 400, (500, dim))])
    idx, cnt = HipEngine.knn_search(x, c, k, **kw)
    monkeypatch.setenv("GSS_KNN_BRUTE", "1")
    bidx, bcnt = HipEngine.knn_search(x, c, k, **kw)
    monkeypatch.delenv("GSS_KNN_BRUTE")
    assert np.array_equal(cnt, bcnt) and np.array_equal(idx, bidx)


@pytest.mark.parametrize("k", [24, 64, 100])
@pytest.mark.parametrize("kind,vkw", [("spherical", dict(radii=(40.0, 25.0, 15.0), nugget=0.05)),
                                       ("matern", dict(range=30.0, nu=1.5, nugget=0.1)),
                                       ("exponential", dict(radii=(30.0, 30.0, 10.0), nugget=0.02))])
def test_estimation_at_data_locations_with_a_nugget(kind, vkw, k):
    """C(0) = sill, C(0+) = sill - nugget: at a data location the covariance to the coincident sample is the sill, and
    the moving-neighbourhood kernels decide that by `d2 <= 0` on coordinates they have SCALED (radii of the model's ball,
    the model's own scale).  The scaled coordinates of the sample and of the estimation point must therefore be the same
    rounded products -- a multiply contracted into the difference leaves a residual of 1e-32 and the nugget is lost
    (round 4: caught by the spherical case of test_local_kriging_matches_oracle once the isotropic models were scaled
    too).  Every kernel family: 24 and 64 neighbours (wave per point), 100 (tile kernel)."""
    from gss.engine import KrigHandle
    gvg, ovg = _vgs(kind, **vkw)
    rng = np.random.default_rng(k)
    x = rng.uniform(0, 100, (500, 3))
    z = rng.normal(size=500) + 0.01 * x[:, 1]
    x0 = np.concatenate([x[:40], rng.uniform(0, 100, (60, 3))])
    h = KrigHandle(gvg, K.OK, x, z, factor=False)
    mu, var, st = h.predict_knn(x0, k)[:3]
    h.close()
    rmu, rvar, rst = K.approxsolve(K.OK, ovg, x, z, x0, k)[:3]
    assert np.array_equal(st, rst)
    assert np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9
    assert np.max(np.abs(mu[:40] - z[:40])) < 1e-9            # C(0) = sill on both sides of the system: exact at the data
