"""the rest of the variogram zoo — SineHole and the non-stationary Power model.

Power has no sill: the oracle solves the variogram form [G F; F' 0] ([DEP] GeoStatsModels) while the
device uses the pseudo-covariance A - gamma(h); with the unbiasedness constraint both give the same estimates
for any A, which is what these tests check (1e-9 on well-conditioned cases)."""
import numpy as np
import pytest

import gss
from oracle import kriging as K
from oracle.variogram import Variogram, cov_pairwise, gamma_h, isstationary
from oracle_engine import OracleEngine


def test_model_formulas():
    h = np.array([0.0, 0.5, 1.0, 2.0, 4.0])
    s = Variogram("sinehole", sill=2.0, nugget=0.5, range=2.0)
    t = np.pi * h[1:] / 2.0
    assert gamma_h(s, h)[0] == 0.0 and np.allclose(gamma_h(s, h)[1:], 1.5 * (1 - np.sin(t) / t) + 0.5)
    p = Variogram("power", range=0.7, nu=1.5, nugget=0.1)
    assert gamma_h(p, h)[0] == 0.0 and np.allclose(gamma_h(p, h)[1:], 0.7 * h[1:] ** 1.5 + 0.1)
    assert isstationary(s) and not isstationary(p)
    assert not gss.PowerVariogram().isstationary() and gss.SineHoleVariogram(range=3.0).isstationary()
    with pytest.raises(ValueError):
        gss.PowerVariogram(exponent=2.0)


def test_power_kriging_is_exact_and_independent_of_the_pseudo_sill():
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 10, (30, 2))
    z = rng.normal(size=30)
    p = Variogram("power", range=0.7, nu=1.3, nugget=0.05)
    mu, var = K.exactsolve(K.OK, p, x, z, np.vstack([x[:3], rng.uniform(0, 10, (4, 2))]))
    assert np.allclose(mu[:3], z[:3], atol=1e-10) and np.allclose(var[:3], 0.0, atol=1e-9)    # interpolation
    from oracle.variogram import pairwise
    x0 = rng.uniform(0, 10, (6, 2))
    mu, var = K.exactsolve(K.OK, p, x, z, x0)
    for A in (30.0, 3000.0):
        lhs = np.zeros((31, 31))
        lhs[:30, :30] = A - pairwise(p, x)
        lhs[:30, 30] = lhs[30, :30] = 1.0
        rhs = np.ones((31, 6))
        rhs[:30] = A - pairwise(p, x, x0)
        w = np.linalg.solve(lhs, rhs)
        assert np.allclose(w[:30].T @ z, mu, atol=1e-9) and np.allclose(A - np.sum(rhs * w, axis=0), var, atol=1e-8)


def test_simulation_solvers_reject_the_power_model():          # fft.jl:91-93, lu.jl:110
    grid = gss.CartesianGrid(8, 8)
    prob = gss.SimulationProblem(grid, {"z": float}, 1)
    for solver in (gss.FFTGS(("z", dict(variogram=gss.PowerVariogram())), engine=OracleEngine),
                   gss.LUGS(("z", dict(variogram=gss.PowerVariogram())), engine=OracleEngine)):
        with pytest.raises((AssertionError, ValueError), match="stationary"):
            gss.solve(prob, solver)


# ------------------------------------------------------------------------------------------
# device
# ------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_sinehole_covariance_and_kriging_on_device():
    from gss.engine import HipEngine, KrigHandle, OK
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 20, (150, 3))
    b = rng.uniform(0, 20, (90, 3))
    gv, ov = gss.SineHoleVariogram(range=6.0, sill=1.4, nugget=0.2), Variogram("sinehole", range=6.0, sill=1.4, nugget=0.2)
    assert np.max(np.abs(HipEngine.cov_pairwise(gv, a, b) - cov_pairwise(ov, a, b))) < 1e-13
    assert np.max(np.abs(HipEngine.cov_pairwise(gv, a) - cov_pairwise(ov, a))) < 1e-13
    z = rng.normal(size=150)
    h = KrigHandle(gv, OK, a, z)
    mu, var, st = h.predict_global(b)
    rmu, rvar = K.exactsolve(K.OK, ov, a, z, b)
    assert not st.any() and np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9
    nested = gss.SineHoleVariogram(range=6.0, sill=0.7, nugget=0.1) + 0.5 * gss.ExponentialVariogram(range=9.0)
    from oracle.variogram import Nested
    onest = Nested([(1.0, Variogram("sinehole", range=6.0, sill=0.7, nugget=0.1)), (0.5, Variogram("exponential", range=9.0))])
    assert np.max(np.abs(HipEngine.cov_pairwise(nested, a, b) - cov_pairwise(onest, a, b))) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("variant,okw", [(K.OK, {}), (K.UK, dict(degree=1)), (K.UK, dict(degree=2))])
@pytest.mark.parametrize("dim,exponent", [(2, 1.0), (3, 1.5), (1, 0.6)])
def test_power_variogram_kriging_on_device(variant, okw, dim, exponent):
    from gss.engine import KrigHandle
    rng = np.random.default_rng(10 * dim + int(10 * exponent))
    n = 400 if dim > 1 else 60
    x = rng.uniform(0, 50, (n, dim))
    z = 0.05 * x[:, 0] + rng.normal(size=n)
    x0 = np.vstack([x[:5], rng.uniform(0, 50, (300, dim))])
    gv = gss.PowerVariogram(scaling=0.3, exponent=exponent, nugget=0.2)
    ov = Variogram("power", range=0.3, nu=exponent, nugget=0.2)
    h = KrigHandle(gv, variant, x, z, degree=okw.get("degree"))
    mu, var, st = h.predict_global(x0)
    rmu, rvar = K.exactsolve(variant, ov, x, z, x0, **okw)
    assert not st.any() and np.max(np.abs(mu - rmu)) < 1e-8 and np.max(np.abs(var - rvar)) < 1e-8
    assert np.max(np.abs(mu[:5] - z[:5])) < 1e-8                       # exact interpolation at the samples
    hl = KrigHandle(gv, variant, x, z, degree=okw.get("degree"), factor=False)
    k = 24 if dim > 1 else 8
    lmu, lvar, lst = hl.predict_knn(x0, k)
    rl = K.approxsolve(variant, ov, x, z, x0, k, **okw)
    assert np.array_equal(lst, rl[2]) and np.max(np.abs(lmu - rl[0])) < 1e-8 and np.max(np.abs(lvar - rl[1])) < 1e-8


@pytest.mark.gpu
def test_power_variogram_is_refused_where_the_reference_refuses_it():
    from gss import _lib
    from gss.engine import FFTGSHandle, KrigHandle, SK
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 10, (20, 2))
    with pytest.raises(_lib.GSSError, match="stationary"):
        KrigHandle(gss.PowerVariogram(), SK, x, rng.normal(size=20), mean=0.0)
    l = _lib.lib()
    import ctypes as C
    v = _lib.make_variogram("power", 2, sill=10.0, nugget=0.0, range=1.0, nu=1.0)
    h = C.c_void_p()
    dims = (C.c_int64 * 2)(16, 16)
    assert l.gss_fftgs_create(C.byref(h), C.byref(v), 2, dims, None, 0.0, 0, None) == _lib.ERR_INVALID
    assert "stationary" in _lib.last_error()
    sol = gss.solve(gss.EstimationProblem(gss.georef(dict(z=rng.normal(size=20)), x), gss.CartesianGrid(10, 10), "z"),
                    gss.KrigingSolver(("z", dict(variogram=gss.PowerVariogram(exponent=1.2, nugget=0.1)))))
    assert np.all(np.isfinite(sol["z"])) and np.all(sol["z_variance"] >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize("order", [1.0, 2.0, 3.0])
def test_matern_integer_orders_on_device(order):
    """MaternVariogram(order = 1) is the reference's default order ([DEP] Variography); integer orders need the
    modified Bessel functions K0 / K1 on the device (Chebyshev expansions, csrc/gss_internal.h).  Oracle: scipy kv."""
    from gss.engine import HipEngine, KrigHandle, OK
    rng = np.random.default_rng(int(order) + 40)
    a = rng.uniform(0, 60, (200, 3))
    b = np.vstack([a[:5], rng.uniform(0, 60, (120, 3)), a[:1] + 1e-7, a[:1] + 500.0])   # zero, tiny and huge lags
    gv = gss.MaternVariogram(range=25.0, order=order, sill=1.7, nugget=0.1)
    ov = Variogram("matern", range=25.0, nu=order, sill=1.7, nugget=0.1)
    assert np.max(np.abs(HipEngine.cov_pairwise(gv, a, b) - cov_pairwise(ov, a, b))) < 5e-14
    z = rng.normal(size=200)
    h = KrigHandle(gv, OK, a, z)
    mu, var, st = h.predict_global(b[:125])
    rmu, rvar = K.exactsolve(K.OK, ov, a, z, b[:125])
    assert not st.any() and np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9
    hl = KrigHandle(gv, OK, a, z, factor=False)
    lmu, lvar, lst = hl.predict_knn(b[:125], 20)
    r = K.approxsolve(K.OK, ov, a, z, b[:125], 20)
    assert np.max(np.abs(lmu - r[0])) < 1e-9 and np.max(np.abs(lvar - r[1])) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("order", [0.2, 0.7, 1.25, 2.6, 4.2, 9.5])
def test_matern_general_order_on_device(order):
    """Any positive Matern order, as [DEP] Variography allows: K_nu by Temme's series / Steed's continued fraction
    on the device (csrc/gss_internal.h gss_matern_general, tables from tools/gen_matern_general.py).  Oracle: scipy kv."""
    from gss.engine import HipEngine, KrigHandle, OK, UK
    rng = np.random.default_rng(int(order * 100))
    a = rng.uniform(0, 60, (150, 2))
    b = np.vstack([a[:5], rng.uniform(0, 60, (100, 2)), a[:1] + 1e-7, a[:1] + 800.0])     # zero, tiny and huge lags
    gv = gss.MaternVariogram(range=25.0, order=order, sill=1.3, nugget=0.05)
    ov = Variogram("matern", range=25.0, nu=order, sill=1.3, nugget=0.05)
    assert np.max(np.abs(HipEngine.cov_pairwise(gv, a, b) - cov_pairwise(ov, a, b))) < 2e-13
    z = rng.normal(size=150)
    h = KrigHandle(gv, OK, a, z)
    mu, var, st = h.predict_global(b[:105])
    rmu, rvar = K.exactsolve(K.OK, ov, a, z, b[:105])
    assert not st.any() and np.max(np.abs(mu - rmu)) < 1e-8 and np.max(np.abs(var - rvar)) < 1e-8
    hl = KrigHandle(gv, UK, a, z, degree=1, factor=False)
    lmu, lvar, lst = hl.predict_knn(b[:105], 24)
    r = K.approxsolve(K.UK, ov, a, z, b[:105], 24, degree=1)
    assert np.array_equal(lst, r[2]) and np.max(np.abs(lmu - r[0])) < 1e-8 and np.max(np.abs(lvar - r[1])) < 1e-8
    # nested with a second structure of general order
    from oracle.variogram import Nested
    gn = 0.6 * gss.GaussianVariogram(range=10.0) + 0.4 * gv
    on = Nested([(0.6, Variogram("gaussian", range=10.0)), (0.4, ov)])
    assert np.max(np.abs(HipEngine.cov_pairwise(gn, a, b) - cov_pairwise(on, a, b))) < 2e-13


@pytest.mark.gpu
def test_matern_order_outside_range_is_invalid():
    from gss.engine import KrigHandle, OK
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 10, (10, 2))
    for bad in (0.0, -1.0, 80.0):
        with pytest.raises(gss._lib.GSSError, match="must lie"):
            KrigHandle(gss.MaternVariogram(range=5.0, order=bad), OK, a, rng.normal(size=10))
