"""maxneighbors between 65 and n - 1.  The reference accepts any count (src/ui.jl:16-23 clamps only above n;
src/estimation/krig.jl:201-210 sizes its buffer by it); on the device the search runs in passes of 64 (each pass
bounded below by the last key of the pass before) and the estimators switch to their any-k kernels (one workgroup per
point for kriging, one thread per point walking the list for IDW / LWR).  Same bars as the k <= 64 paths: neighbour
indices bit-exact, estimates 1e-9 against the oracle."""
import numpy as np
import pytest

from oracle import idw_lwr as E, kriging as K
from oracle.variogram import Variogram

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m,dim,k,ball", [(300, 200, 2, 65, None), (1000, 150, 3, 128, None), (700, 100, 1, 130, None),
                                            (2500, 120, 3, 200, None), (2000, 150, 2, 150, 9.0),
                                            (400, 50, 3, 399, None)])
def test_knn_any_k_indices_bit_exact(n, m, dim, k, ball):
    from gss.engine import HipEngine
    rng = np.random.default_rng(n + k)
    x = rng.uniform(0, 100, (n, dim))
    x[50:60] = x[:10]                                   # exact duplicates: ties broken by index across pass boundaries
    c = rng.uniform(0, 100, (m, dim))
    c[:3] = x[:3]
    kw = {} if ball is None else dict(radius=ball)
    idx, cnt = HipEngine.knn_search(x, c, k, **kw)
    ridx, rcnt = K.knn_search(x, c, k, **kw)
    assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx)
    if ball is not None:
        assert cnt.min() < k           # the ball cuts some lists short (a pass comes back short, the next finds nothing)


CASES = [(K.OK, {}, "matern", dict(range=30.0, nu=1.5)),
         (K.SK, dict(mean=0.3), "exponential", dict(range=25.0, sill=2.0, nugget=0.1)),
         (K.UK, dict(degree=1), "spherical", dict(range=40.0, nugget=0.05)),
         (K.UK, dict(degree=2), "matern", dict(range=35.0, nu=2.5, nugget=0.02))]


@pytest.mark.parametrize("variant,okw,kind,vkw", CASES)
@pytest.mark.parametrize("n,m,dim,k", [(500, 120, 3, 65), (900, 100, 3, 128), (1500, 60, 2, 170), (1200, 40, 3, 260),
                                       (1400, 50, 3, 192), (1600, 40, 2, 256)])
def test_local_kriging_with_more_than_64_neighbours(variant, okw, kind, vkw, n, m, dim, k):
    """65 .. 256 neighbours: the register-distributed tile kernel (4 waves up to 128, 8 waves up to 256; 128, 192 and 256
    fill their tile counts exactly, 65 / 170 leave the last tile ragged and waves without a column pair); beyond
    256 (k = 260 here) the scalar kernel with its system in a per-workgroup slab."""
    import gss
    from gss.engine import KrigHandle
    ctor = dict(exponential=gss.ExponentialVariogram, spherical=gss.SphericalVariogram, matern=gss.MaternVariogram)[kind]
    gkw = dict(vkw)
    if "nu" in gkw:
        gkw["order"] = gkw.pop("nu")
    gvg, ovg = ctor(**gkw), Variogram(kind, **vkw)
    rng = np.random.default_rng(n * 3 + k)
    x = rng.uniform(0, 100, (n, dim))
    z = rng.normal(size=n) + 0.02 * x[:, 0]
    x0 = rng.uniform(0, 100, (m, dim))
    x0[:2] = x[:2]
    h = KrigHandle(gvg, variant, x, z, mean=okw.get("mean"), degree=okw.get("degree"), factor=False)
    mu, var, st, idx, cnt = h.predict_knn(x0, k, return_idx=True)
    h.close()
    rmu, rvar, rst, ridx, rcnt = K.approxsolve(variant, ovg, x, z, x0, k, mean=okw.get("mean") or 0.0,
                                               degree=okw.get("degree"), return_idx=True)
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt) and np.array_equal(st, rst) and not st.any()
    assert np.max(np.abs(mu - rmu)) < 1e-9 * max(1.0, np.max(np.abs(rmu)))
    assert np.max(np.abs(var - rvar)) < 1e-9
    assert np.allclose(mu[:2], z[:2], atol=1e-9) or vkw.get("nugget", 0) > 0


def test_ball_limited_large_neighbourhood_and_missing_points():
    """KBallSearch semantics with k > 64: neighbour counts vary from 0 to k inside one launch."""
    import gss
    from gss.engine import KrigHandle
    rng = np.random.default_rng(9)
    x = rng.uniform(0, 100, (3000, 2))
    z = rng.normal(size=3000)
    x0 = np.concatenate([rng.uniform(0, 100, (150, 2)), rng.uniform(300, 400, (10, 2))])
    h = KrigHandle(gss.ExponentialVariogram(range=20.0, nugget=0.05), K.OK, x, z, factor=False)
    mu, var, st, idx, cnt = h.predict_knn(x0, 100, minneighbors=3, radius=9.0, return_idx=True)
    h.close()
    rmu, rvar, rst, ridx, rcnt = K.approxsolve(K.OK, Variogram("exponential", range=20.0, nugget=0.05), x, z, x0, 100, 3,
                                               radius=9.0, return_idx=True)
    assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx) and np.array_equal(st, rst)
    assert cnt.max() > 64 and st[-10:].all() and cnt[-10:].max() == 0
    ok = st == 0
    assert np.max(np.abs(mu[ok] - rmu[ok])) < 1e-9 and np.max(np.abs(var[ok] - rvar[ok])) < 1e-9


def test_k_just_below_n_approaches_global_and_solve_api():
    """maxneighbors = n - 1 through the solver front-end (krig.jl:151-157 takes the approxsolve branch)."""
    import gss
    rng = np.random.default_rng(21)
    xy = rng.uniform(0, 50, (90, 2))
    z = rng.normal(size=90)
    dom = gss.PointSet(rng.uniform(0, 50, (40, 2)))
    prob = gss.EstimationProblem(gss.georef({"z": z}, xy), dom, "z")
    vg = gss.SphericalVariogram(range=20.0, nugget=0.1)
    sol = gss.solve(prob, gss.KrigingSolver(("z", dict(variogram=vg, maxneighbors=89))))
    rmu, rvar, rst = K.approxsolve(K.OK, Variogram("spherical", range=20.0, nugget=0.1), xy, z, dom.coords, 89)
    assert np.max(np.abs(sol["z"] - rmu)) < 1e-9 and np.max(np.abs(sol["z_variance"] - rvar)) < 1e-9


@pytest.mark.parametrize("n,m,dim,k", [(500, 300, 2, 65), (3000, 200, 3, 150), (400, 100, 1, 300)])
def test_idw_lwr_with_more_than_64_neighbours(n, m, dim, k):
    from gss.engine import HipEngine
    rng = np.random.default_rng(n + dim)
    x = rng.uniform(0, 100, (n, dim))
    z = np.sin(x[:, 0] / 17.0) + 0.01 * x.sum(axis=1) + 0.1 * rng.normal(size=n)
    c = rng.uniform(-5, 105, (m, dim))
    c[:4] = x[:4]
    for exponent in (1, 2.5):
        mu, sd, st = HipEngine.idw(x, z, c, k, 1, exponent)
        rmu, rsd, rst = E.idw(x, z, c, k, 1, exponent)
        assert np.array_equal(st, rst) and np.max(np.abs(mu - rmu)) < 1e-10 and np.max(np.abs(sd - rsd)) < 1e-10
        assert np.array_equal(mu[:4], z[:4]) and np.all(sd[:4] == 0.0)
    for spec, wf in (((0, 3.0, 2.0), E.default_weightfun), ((1, 0.0, 0.0), E.tricube)):
        mu, var, st = HipEngine.lwr(x, z, c, k, 1, spec)
        rmu, rvar, rst = E.lwr(x, z, c, k, 1, wf)
        assert np.array_equal(st, rst) and not st.any()
        assert np.max(np.abs(mu - rmu)) < 1e-9 * max(1.0, np.max(np.abs(rmu))) and np.max(np.abs(var - rvar)) < 1e-9
    # ball-limited
    mu, sd, st = HipEngine.idw(x, z, c, k, 2, 1, radius=12.0)
    rmu, rsd, rst = E.idw(x, z, c, k, 2, 1, radius=12.0)
    assert np.array_equal(st, rst)
    ok = st == 0
    assert np.max(np.abs(mu[ok] - rmu[ok])) < 1e-10


def test_many_neighbours_anisotropic_nested_and_external_drift():
    """The many-neighbour kernel's model-specific instantiations divide the coordinates by the radii of the model's
    ball once per point (anisotropic exponential model); a nested model and a Gaussian one take the general
    instantiation; an external drift supplies its own right-hand-side columns.  k = 100, against the oracle, 1e-9
    (Gaussian: 1e-6)."""
    import gss
    from gss.engine import KrigHandle
    from oracle.variogram import Nested
    rng = np.random.default_rng(33)
    x = rng.uniform(0, 100, (900, 3))
    z = np.sin(x[:, 0] / 20.0) + 0.02 * x[:, 1] + 0.2 * rng.normal(size=900)
    x0 = rng.uniform(5, 95, (60, 3))
    cases = [
        (gss.ExponentialVariogram(gss.MetricBall((40.0, 20.0, 10.0)), nugget=0.05),
         Variogram("exponential", radii=(40.0, 20.0, 10.0), nugget=0.05), 1e-9),
        (0.6 * gss.SphericalVariogram(range=30.0, nugget=0.05) + 0.4 * gss.ExponentialVariogram(range=60.0),
         Nested([(0.6, Variogram("spherical", range=30.0, nugget=0.05)), (0.4, Variogram("exponential", range=60.0))]), 1e-9),
        (gss.GaussianVariogram(range=15.0, nugget=0.05), Variogram("gaussian", range=15.0, nugget=0.05), 1e-6),
    ]
    for gvg, ovg, tol in cases:
        h = KrigHandle(gvg, K.OK, x, z, factor=False)
        mu, var, st = h.predict_knn(x0, 100)
        h.close()
        rmu, rvar, rst = K.approxsolve(K.OK, ovg, x, z, x0, 100)
        assert not st.any() and np.max(np.abs(mu - rmu)) < tol and np.max(np.abs(var - rvar)) < tol
    # 1-D (the general instantiation); a nugget keeps 100 collinear neighbours of a smooth model well conditioned
    x1 = rng.uniform(0, 100, (300, 1))
    z1 = rng.normal(size=300)
    h = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5, nugget=0.1), K.UK, x1, z1, degree=1, factor=False)
    mu, var, st = h.predict_knn(x0[:, :1], 100)
    h.close()
    rmu, rvar, rst = K.approxsolve(K.UK, Variogram("matern", range=30.0, nu=1.5, nugget=0.1), x1, z1, x0[:, :1], 100, degree=1)
    assert not st.any() and np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9
    fd = np.c_[np.ones(900), x[:, 0] / 100.0, (x[:, 1] / 100.0) ** 2]
    f0 = np.c_[np.ones(60), x0[:, 0] / 100.0, (x0[:, 1] / 100.0) ** 2]
    gvg, ovg = gss.MaternVariogram(range=30.0, order=1.5, nugget=0.02), Variogram("matern", range=30.0, nu=1.5, nugget=0.02)
    h = KrigHandle(gvg, K.EDK, x, z, drift_data=fd, factor=False)
    mu, var, st = h.predict_knn(x0, 100, drift_dom=f0)
    h.close()
    rmu, rvar, rst = K.approxsolve(K.EDK, ovg, x, z, x0, 100, drift_data=fd, drift_dom=f0)
    assert not st.any() and np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9


def test_block_support_in_moving_neighbourhoods_every_kernel():
    """`gss_krig_set_block_support` followed by `gss_krig_predict_knn`: the right-hand side of each local system is the
    covariance averaged over the cell around the point and C(V,V) replaces the sill in the variance (krig.jl:226 hands
    the cell to `predict`).  Neighbour counts 12 / 100 / 200 / 300 reach the in-register tile kernel, the tile columns
    over four and over eight waves, and the general kernel; the anisotropic model takes the instantiations that scale
    the coordinates by the ball's radii, the nested one the general covariance.  Against the oracle, 1e-9."""
    import gss
    from gss.engine import KrigHandle
    from oracle.variogram import Nested
    rng = np.random.default_rng(35)
    x = rng.uniform(0, 100, (700, 3))
    z = np.sin(x[:, 0] / 20.0) + 0.02 * x[:, 1] + 0.2 * rng.normal(size=700)
    x0 = rng.uniform(5, 95, (24, 3))
    cell, nsub = (4.0, 2.0, 1.0), 2
    cases = [
        (gss.ExponentialVariogram(gss.MetricBall((40.0, 20.0, 10.0)), nugget=0.05),
         Variogram("exponential", radii=(40.0, 20.0, 10.0), nugget=0.05)),
        (gss.MaternVariogram(range=30.0, order=1.5, nugget=0.05), Variogram("matern", range=30.0, nu=1.5, nugget=0.05)),
        (0.6 * gss.SphericalVariogram(range=30.0, nugget=0.05) + 0.4 * gss.ExponentialVariogram(range=60.0),
         Nested([(0.6, Variogram("spherical", range=30.0, nugget=0.05)), (0.4, Variogram("exponential", range=60.0))])),
    ]
    for gvg, ovg in cases:
        for variant, deg, k in ((K.OK, None, 12), (K.UK, 1, 100), (K.OK, None, 200), (K.UK, 1, 300)):
            okw = {} if deg is None else dict(degree=deg)
            h = KrigHandle(gvg, variant, x, z, factor=False, **okw)
            h.set_block_support(cell, nsub)
            mu, var, st = h.predict_knn(x0, k)
            h.close()
            rmu, rvar, rst = K.approxsolve(variant, ovg, x, z, x0, k, support=(cell, nsub), **okw)
            assert not st.any(), (ovg, k)
            assert np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9, (ovg, k)
    # the point-support answer differs (the test would pass vacuously otherwise)
    h = KrigHandle(cases[1][0], K.OK, x, z, factor=False)
    mu_p, var_p, _ = h.predict_knn(x0, 12)
    h.close()
    rmu, rvar, _ = K.approxsolve(K.OK, cases[1][1], x, z, x0, 12, support=(cell, nsub))
    assert np.max(np.abs(var_p - rvar)) > 1e-3


@pytest.mark.parametrize("variant,deg,kind,vkw,dim,k", [
    (K.UK, 1, "matern", dict(range=30.0, nu=1.5), 3, 512), (K.OK, None, "spherical", dict(range=40.0, nugget=0.05), 2, 700),
    (K.SK, None, "exponential", dict(range=25.0), 3, 768), (K.OK, None, "gaussian", dict(range=8.0, nugget=0.3), 3, 300)])
def test_slab_kernel_257_to_768_neighbours(variant, deg, kind, vkw, dim, k):
    """257 .. 768 neighbours run the tile algorithm with the block triangle in a per-workgroup slab of global memory
    (csrc/krig_slab.hip; round 4 -- until then the unblocked column sweep of krig_local_big_kernel, 25 x slower at 512).
    Against the oracle, 1e-9 (1e-6 for the Gaussian model), estimation points at data locations included (C(0) = sill);
    ragged neighbour counts under a ball; and the same call on the column sweep (GSS_KRIG_SLAB_OFF=1, read per call)."""
    import os
    import gss
    from gss.engine import KrigHandle
    ctor = dict(gaussian=gss.GaussianVariogram, exponential=gss.ExponentialVariogram, spherical=gss.SphericalVariogram,
                matern=gss.MaternVariogram)[kind]
    okw = dict(vkw)
    gkw = dict(vkw)
    if "nu" in gkw:
        gkw["order"] = gkw.pop("nu")
    gvg, ovg = ctor(**gkw), Variogram(kind, **okw)
    rng = np.random.default_rng(k + dim)
    n = 1600
    x = rng.uniform(0, 100, (n, dim))
    z = rng.normal(size=n) + 0.02 * x[:, 0]
    x0 = np.concatenate([x[:5], rng.uniform(0, 100, (45, dim))])
    tol = 1e-6 if kind == "gaussian" else 1e-9
    h = KrigHandle(gvg, variant, x, z, mean=0.3 if variant == K.SK else None, degree=deg, factor=False)
    mu, var, st = h.predict_knn(x0, k)[:3]
    rmu, rvar, rst = K.approxsolve(variant, ovg, x, z, x0, k, mean=0.3 if variant == K.SK else 0.0, degree=deg)[:3]
    assert np.array_equal(st, rst) and not st.any()
    assert np.max(np.abs(mu - rmu)) < tol * max(1.0, np.max(np.abs(rmu))) and np.max(np.abs(var - rvar)) < tol
    # a ball that leaves between 257 and k neighbours, and too few for some points
    radius = 38.0 if dim == 3 else 30.0
    mub, varb, stb = h.predict_knn(x0, k, minneighbors=280, radius=radius)[:3]
    rb = K.approxsolve(variant, ovg, x, z, x0, k, minneighbors=280, radius=radius, mean=0.3 if variant == K.SK else 0.0,
                       degree=deg)
    assert np.array_equal(stb, rb[2])
    okb = stb == 0
    if okb.any():
        assert np.max(np.abs(mub[okb] - rb[0][okb])) < tol * 10 and np.max(np.abs(varb[okb] - rb[1][okb])) < tol * 10
    try:
        os.environ["GSS_KRIG_SLAB_OFF"] = "1"
        mu2, var2, st2 = h.predict_knn(x0, k)[:3]
    finally:
        os.environ.pop("GSS_KRIG_SLAB_OFF", None)
    assert np.array_equal(st2, st) and np.max(np.abs(mu2 - mu)) < tol and np.max(np.abs(var2 - var)) < tol
    h.close()


@pytest.mark.parametrize("k", [63, 64, 65, 80, 81, 96, 97, 112, 113, 128, 129, 144, 240, 255, 256, 257, 272, 400, 767, 768,
                               769, 800])
def test_every_kernel_boundary_with_ragged_neighbour_counts(k):
    """maxneighbors on both sides of every switch of the moving-neighbourhood dispatch (64 | 65: one wave per point ->
    register tiles; 96 | 97 and 128 | 129: 3 -> 4 -> 8 waves; 256 | 257: tiles -> slab kernel; 768 | 769: slab -> scalar
    kernel), with a ball that leaves the counts anywhere between minneighbors and k inside one launch, universal kriging
    of degree 1, a nugget, and estimation points on data locations.  Indices bit-exact, estimates 1e-9."""
    import gss
    from gss.engine import KrigHandle
    rng = np.random.default_rng(1000 + k)
    n, m = max(1200, 2 * k), 48
    x = rng.uniform(0, 100, (n, 3))
    z = rng.normal(size=n) + 0.03 * x[:, 1]
    x0 = rng.uniform(5, 95, (m, 3))
    x0[:3] = x[:3]
    x0[-4:] = rng.uniform(400, 500, (4, 3))            # nothing inside the ball: missing
    # radius so that a typical ball holds about 0.8 k samples: some lists are full, most are cut short
    radius = (0.8 * k / n * 100.0 ** 3 * 3.0 / (4.0 * np.pi)) ** (1.0 / 3.0)
    vkw = dict(range=35.0, sill=1.3, nugget=0.05)
    h = KrigHandle(gss.SphericalVariogram(**vkw), K.UK, x, z, degree=1, factor=False)
    mu, var, st, idx, cnt = h.predict_knn(x0, k, minneighbors=5, radius=radius, return_idx=True)
    h.close()
    rmu, rvar, rst, ridx, rcnt = K.approxsolve(K.UK, Variogram("spherical", **vkw), x, z, x0, k, 5, degree=1,
                                               radius=radius, return_idx=True)
    assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx) and np.array_equal(st, rst)
    assert st[-4:].all() and cnt[:-4].min() < cnt[:-4].max()
    ok = st == 0
    assert ok.sum() >= m - 8
    assert np.max(np.abs(mu[ok] - rmu[ok])) < 1e-9 * max(1.0, np.max(np.abs(rmu[ok])))
    assert np.max(np.abs(var[ok] - rvar[ok])) < 1e-9
