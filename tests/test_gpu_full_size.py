"""BASELINE.json's full sizes on the device, checked through size-independent properties plus an oracle sample
(the oracle cannot finish 10^6 points in seconds): configs[1] global kriging (n = 1000, m = 10^6) and the per-GPU
share of configs[4] (n = 5000, k = 64, UK degree 1, m = 1.25 * 10^6)."""
import numpy as np
import pytest

from oracle import kriging as K
from oracle.variogram import Variogram

pytestmark = pytest.mark.gpu


def test_config2_global_kriging_one_million_points():
    import torch
    import gss
    from gss.engine import KrigHandle, OK
    n, m = 1000, 1_000_000
    x = np.random.default_rng(2).uniform(0.0, 100.0, (n, 3))
    z1 = np.random.default_rng(1002).normal(size=n)
    z2 = np.random.default_rng(77).normal(size=n)
    x0 = np.random.default_rng(3).uniform(0.0, 100.0, (m, 3))
    x0[:n] = x                                             # the first n domain points ARE the samples
    xd = torch.as_tensor(x0, device="cuda")
    vg = gss.MaternVariogram(range=30.0, order=1.5)
    outs = []
    for z in (z1, z2, z1 + 2.0 * z2):
        h = KrigHandle(vg, OK, x, z)
        mu, var, st = h.predict_global(xd)
        outs.append((mu.cpu().numpy(), var.cpu().numpy(), st.cpu().numpy()))
        h.close()
    (m1, v1, s1), (m2, v2, s2), (m3, v3, s3) = outs
    assert not s1.any() and not s3.any()
    # exact interpolation at the samples (test/estimation/krig.jl:35-37), zero variance there, variance <= sill
    assert np.max(np.abs(m1[:n] - z1)) < 1e-9 and np.max(v1[:n]) < 1e-9
    assert v1.min() >= 0.0 and v1.max() <= 1.0 + 1e-12
    # the estimator is linear in the data and its variance does not depend on them
    assert np.max(np.abs(m3 - (m1 + 2.0 * m2))) < 1e-9
    assert np.array_equal(v1, v2) and np.array_equal(v1, v3)
    # a separate call on a slice gives the same numbers up to summation order (a short launch runs as
    # (strip, row block) units whose partial sums are added in a different order)
    h = KrigHandle(vg, OK, x, z1)
    ms, vs, _ = h.predict_global(xd[500_000:500_777])
    h.close()
    assert np.max(np.abs(ms.cpu().numpy() - m1[500_000:500_777])) < 1e-12
    assert np.max(np.abs(vs.cpu().numpy() - v1[500_000:500_777])) < 1e-12
    # oracle on a sample
    sel = np.linspace(n, m - 1, 200).astype(np.int64)
    rmu, rvar = K.exactsolve(K.OK, Variogram("matern", range=30.0, nu=1.5), x, z1, x0[sel])
    assert np.max(np.abs(m1[sel] - rmu)) < 1e-9 and np.max(np.abs(v1[sel] - rvar)) < 1e-9


def test_config5_share_moving_neighbourhood():
    import torch
    import gss
    from gss.engine import KrigHandle, UK
    n, m, k = 5000, 1_250_000, 64
    x = np.random.default_rng(6).uniform(0, 100, (n, 3))
    trend = 1.0 + 0.03 * x[:, 0] - 0.02 * x[:, 1] + 0.01 * x[:, 2]
    x0 = np.random.default_rng(7).uniform(0, 100, (m, 3))
    xd = torch.as_tensor(x0, device="cuda")
    vg = gss.MaternVariogram(range=30.0, order=1.5)
    # universal kriging of degree 1 reproduces an affine field exactly, with any neighbourhood
    h = KrigHandle(vg, UK, x, trend, degree=1, factor=False)
    mu, var, st, idx, cnt = h.predict_knn(xd, k, return_idx=True)
    h.close()
    mu, var, st, idx, cnt = (t.cpu().numpy() for t in (mu, var, st, idx, cnt))
    assert not st.any() and np.all(cnt == k)
    assert np.max(np.abs(mu - (1.0 + 0.03 * x0[:, 0] - 0.02 * x0[:, 1] + 0.01 * x0[:, 2]))) < 1e-8
    assert var.min() >= 0.0
    # neighbour lists: distinct, sorted by distance, and identical to the oracle's on a sample
    sel = np.linspace(0, m - 1, 300).astype(np.int64)
    for p in sel[:50]:
        d2 = ((x[idx[p]] - x0[p]) ** 2).sum(axis=1)
        assert np.all(np.diff(d2) >= 0) and len(set(idx[p].tolist())) == k
    ridx, _ = K.knn_search(x, x0[sel], k)
    assert np.array_equal(idx[sel], ridx)
    z = trend + np.random.default_rng(60).normal(size=n)
    h = KrigHandle(vg, UK, x, z, degree=1, factor=False)
    mu2, var2, _ = h.predict_knn(xd[sel.tolist()], k)
    h.close()
    r = K.approxsolve(K.UK, Variogram("matern", range=30.0, nu=1.5), x, z, x0[sel], k, degree=1)
    assert np.max(np.abs(mu2.cpu().numpy() - r[0])) < 1e-9 and np.max(np.abs(var2.cpu().numpy() - r[1])) < 1e-9
    assert np.max(np.abs(var2.cpu().numpy() - var[sel])) < 1e-12      # the variance does not depend on the data


def test_config4_lugs_full_size():
    """configs[3]: 128 x 128 grid, 4096 conditioning cells, spherical range 20, 100 realisations.  Hard data exact
    (lu.jl:217); a zero-noise realisation is the conditional mean C21 C11^-1 z1 (lu.jl:136-138, numpy solve);
    realisations scatter around it with at most the prior variance."""
    import gss
    from gss.engine import LUGSHandle
    from oracle import fftgs as offt
    from oracle.variogram import cov_pairwise
    g = 128
    cent = offt.grid_centroids((g, g))
    N = g * g
    rng = np.random.default_rng(5)
    dl = np.sort(rng.permutation(N)[:4096])
    z1 = rng.normal(size=4096)
    h = LUGSHandle(gss.SphericalVariogram(range=20.0), cent, dl, z1)
    y, _ = h.realize(123, 0, 100)
    y0, _ = h.realize(0, 0, 1, noise=np.zeros((1, h.ns)))
    h.close()
    assert np.array_equal(y[:, dl], np.tile(z1, (100, 1))) and np.array_equal(y0[0, dl], z1)
    sl = np.setdiff1d(np.arange(N), dl)
    ovg = Variogram("spherical", range=20.0)
    C11 = cov_pairwise(ovg, cent[dl])
    rows = sl[:: len(sl) // 400]
    cm = cov_pairwise(ovg, cent[rows], cent[dl]) @ np.linalg.solve(C11, z1)
    assert np.max(np.abs(y0[0, rows] - cm)) < 1e-8
    dev = y[:, sl] - y0[0, sl]
    assert abs(dev.mean()) < 0.02 and 0.0 < dev.var() < 1.0 and np.abs(y[0] - y[1]).max() > 0.1


def test_large_data_set_global_universal_kriging():
    """n = 8 000 samples (the factor-and-inverse recursion runs five levels deep, W' is 0.5 GB), UK degree 1,
    60 000 domain points: oracle sample, exactness at samples and affine reproduction (SURVEY.md 8c KATs)."""
    import torch
    import gss
    from gss.engine import KrigHandle, UK
    n, m = 8000, 60_000
    rng = np.random.default_rng(8000)
    x = rng.uniform(0.0, 100.0, (n, 3))
    z = rng.normal(size=n)
    x0 = rng.uniform(0.0, 100.0, (m, 3))
    x0[:500] = x[:500]
    xd = torch.as_tensor(x0, device="cuda")
    vg = gss.MaternVariogram(range=30.0, order=1.5, nugget=0.01)
    h = KrigHandle(vg, UK, x, z, degree=1)
    mu, var, st = (t.cpu().numpy() for t in h.predict_global(xd))
    h.close()
    assert not st.any() and var.min() >= 0.0
    sel = np.concatenate([np.arange(5), np.linspace(500, m - 1, 25).astype(np.int64)])
    rmu, rvar = K.exactsolve(K.UK, Variogram("matern", range=30.0, nu=1.5, nugget=0.01), x, z, x0[sel], degree=1)
    assert np.max(np.abs(mu[sel] - rmu)) < 1e-9 and np.max(np.abs(var[sel] - rvar)) < 1e-9
    assert np.max(np.abs(mu[:500] - z[:500])) < 1e-8 and np.max(var[:500]) < 1e-8
    aff = 2.0 - 0.3 * x[:, 0] + 0.05 * x[:, 1] + 0.7 * x[:, 2]
    h = KrigHandle(vg, UK, x, aff, degree=1)
    mu2 = h.predict_global(xd)[0].cpu().numpy()
    h.close()
    assert np.max(np.abs(mu2 - (2.0 - 0.3 * x0[:, 0] + 0.05 * x0[:, 1] + 0.7 * x0[:, 2]))) < 1e-8
