"""Randomised parity (hypothesis, fixed seed database off): small irregular inputs the hand-picked cases may miss --
odd sizes, duplicates, collinear and lattice points, k = 1 / k = n, single queries -- against the oracle."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import kriging as K
from oracle.variogram import Variogram

pytestmark = pytest.mark.gpu

# (GSS_TEST_EXAMPLES=n: a longer hunt, with fresh random draws instead of the fixed sequence)
_N = int(os.environ.get("GSS_TEST_EXAMPLES", "80"))
SET = dict(max_examples=_N, deadline=None, derandomize=_N == 80, suppress_health_check=list(HealthCheck))


def _points(rng, n, dim, style):
    if style == 0:
        x = rng.uniform(0, 100, (n, dim))
    elif style == 1:                                   # lattice with ties
        x = rng.integers(0, 7, (n, dim)).astype(np.float64) * 3.5
    elif style == 2:                                   # collinear
        x = np.outer(rng.uniform(0, 100, n), np.ones(dim))
    else:                                              # tight clusters + duplicates
        x = rng.normal(50, 0.5, (n, dim))
        x[: n // 3] = x[n // 3: 2 * (n // 3)]
    return x


@settings(**SET)
@given(st.integers(1, 400), st.integers(1, 3), st.integers(0, 3), st.integers(1, 64), st.integers(1, 70),
       st.integers(0, 10_000), st.booleans())
def test_knn_matches_oracle(n, dim, style, k, m, seed, ball):
    from gss.engine import HipEngine
    rng = np.random.default_rng(seed)
    x = _points(rng, n, dim, style)
    k = min(k, n)
    c = np.vstack([x[rng.integers(0, n, m // 2 + 1)], rng.uniform(-20, 120, (m - m // 2, dim))])
    kw = dict(radius=15.0) if ball else {}
    idx, cnt = HipEngine.knn_search(x, c, k, **kw)
    ridx, rcnt = K.knn_search(x, c, k, **kw)
    assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx)


@settings(**SET)
@given(st.integers(8, 300), st.integers(1, 3), st.sampled_from([(K.SK, {}), (K.OK, {}), (K.UK, dict(degree=1))]),
       st.integers(4, 64), st.integers(1, 40), st.integers(0, 10_000),
       st.sampled_from(["exponential", "matern", "spherical", "gaussian"]))
def test_moving_neighbourhood_kriging_matches_oracle(n, dim, var, k, m, seed, kind):
    from gss.engine import KrigHandle
    import gss
    variant, okw = var
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 100, (n, dim))
    if dim == 1:
        x = (np.linspace(0, 100, n) + rng.uniform(-0.2, 0.2, n))[:, None]
    z = rng.normal(size=n)
    k = min(k, n)
    nug = 0.05 if kind == "gaussian" else 0.01
    gv = dict(exponential=gss.ExponentialVariogram, matern=gss.MaternVariogram, spherical=gss.SphericalVariogram,
              gaussian=gss.GaussianVariogram)[kind](range=40.0, nugget=nug)
    ov = Variogram(kind, range=40.0, nugget=nug)
    x0 = rng.uniform(0, 100, (m, dim))
    h = KrigHandle(gv, variant, x, z, mean=0.2 if variant == K.SK else None, degree=okw.get("degree"), factor=False)
    mu, var_, st_ = h.predict_knn(x0, k, minneighbors=min(k, 6), radius=60.0)
    h.close()
    rmu, rvar, rst = K.approxsolve(variant, ov, x, z, x0, k, mean=0.2, degree=okw.get("degree"), minneighbors=min(k, 6),
                                   radius=60.0)
    assert np.array_equal(st_ == 1, rst == 1)                 # `missing` pattern is exact (krig.jl:213-214)
    ok = (st_ == 0) & (rst == 0)
    tol = 1e-6 if kind == "gaussian" else 1e-8
    # (a neighbourhood with as many samples as drift terms is fixed by the unbiasedness constraints alone and can be as
    #  ill conditioned as its geometry -- a nearly flat tetrahedron gives weights and variances in the thousands: the
    #  bar scales with the size of the numbers, |variance| standing in for the size of the weights)
    scale = np.maximum(1.0, np.abs(rvar[ok]))
    assert np.all(np.abs(mu[ok] - rmu[ok]) < tol * np.maximum(1.0, np.abs(rmu[ok])) * scale)
    assert np.all(np.abs(var_[ok] - rvar[ok]) < tol * scale)


@settings(**SET)
@given(st.integers(3, 420), st.integers(1, 3), st.sampled_from([(K.SK, {}), (K.OK, {}), (K.UK, dict(degree=1)),
                                                                 (K.UK, dict(degree=2))]),
       st.integers(1, 3000), st.integers(0, 10_000), st.sampled_from(["exponential", "matern", "spherical"]))
def test_global_kriging_matches_oracle(n, dim, var, m, seed, kind):
    """Every padding of the factor (n + nc against the 16- and 128-row tiles), launches of whole rounds and of split
    tails only, all kriging variants."""
    from gss.engine import KrigHandle
    import gss
    variant, okw = var
    nc = 0 if variant == K.SK else (1 if variant == K.OK else {1: dim + 1, 2: (dim + 1) * (dim + 2) // 2}[okw["degree"]])
    if n < nc + 3:
        n = nc + 3
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 100, (n, dim))
    if dim == 1:
        x = (np.linspace(0, 100, n) + rng.uniform(-0.1, 0.1, n))[:, None]
    z = rng.normal(size=n) + 0.01 * x[:, 0]
    gv = dict(exponential=gss.ExponentialVariogram, matern=gss.MaternVariogram,
              spherical=gss.SphericalVariogram)[kind](range=35.0, nugget=0.02)
    ov = Variogram(kind, range=35.0, nugget=0.02)
    x0 = rng.uniform(0, 100, (m, dim))
    x0[0] = x[0]
    h = KrigHandle(gv, variant, x, z, mean=0.2 if variant == K.SK else None, degree=okw.get("degree"))
    mu, var_, st_ = h.predict_global(x0)
    h.close()
    rmu, rvar = K.exactsolve(variant, ov, x, z, x0, mean=0.2, degree=okw.get("degree"))
    assert not st_.any()
    assert np.max(np.abs(mu - rmu)) < 1e-8 * max(1.0, np.max(np.abs(rmu))) and np.max(np.abs(var_ - rvar)) < 1e-8


@settings(**SET)
@given(st.integers(2, 300), st.integers(1, 3), st.integers(1, 64), st.integers(1, 60), st.integers(0, 10_000),
       st.sampled_from([1.0, 2.0, 0.7]), st.booleans())
def test_idw_lwr_match_oracle(n, dim, k, m, seed, exponent, all_samples):
    from gss.engine import HipEngine
    from oracle import idw_lwr as E
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 100, (n, dim))
    z = rng.normal(size=n)
    kk = None if all_samples else min(k, n)
    c = np.vstack([x[:1], rng.uniform(0, 100, (m, dim))])
    mu, sd, st_ = HipEngine.idw(x, z, c, n if kk is None else kk, 1, exponent)
    rmu, rsd, rst = E.idw(x, z, c, kk, 1, exponent)
    assert np.array_equal(st_, rst) and np.allclose(mu, rmu, rtol=1e-10, atol=1e-12) and np.allclose(sd, rsd, rtol=1e-12, atol=1e-12)
    assert mu[0] == z[0] and sd[0] == 0.0                     # idw.jl:131-134
    if n >= dim + 2:
        lmu, lvar, lst = HipEngine.lwr(x, z, c, n if kk is None else max(kk, min(n, dim + 2)), 1, (0, 3.0, 2.0))
        rlmu, rlvar, rlst = E.lwr(x, z, c, None if kk is None else max(kk, min(n, dim + 2)), 1)
        ok = (lst == 0) & (rlst == 0)
        assert np.array_equal(lst == 1, rlst == 1)
        scale = np.maximum(1.0, np.abs(rlmu[ok]))
        assert np.all(np.abs(lmu[ok] - rlmu[ok]) < 1e-6 * scale)
