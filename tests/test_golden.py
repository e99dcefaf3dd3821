"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the ORACLE -- the
Julia reference cannot be run here, see that script's header).

* not-gpu: the oracle still reproduces its frozen outputs (regression guard), and the reference's own
  assertions hold on the frozen vectors.
* gpu: the device reproduces the frozen vectors without importing the oracle."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_oracle_reproduces_golden_vectors():
    from oracle import fftgs, kriging as K, lugs
    from oracle.variogram import Variogram
    g = _load("krig_reference_2d.npz")
    mu, var = K.exactsolve(K.OK, Variogram("gaussian", range=35.0), g["x"], g["z"], g["grid"])
    assert np.allclose(mu, g["mu_global"], atol=1e-12) and np.allclose(var, g["var_global"], atol=1e-12)
    Z = g["mu_global"].reshape(100, 100).T
    assert abs(Z[24, 24] - 1) < 1e-3 and abs(Z[49, 74]) < 1e-3 and abs(Z[74, 49] - 1) < 1e-3      # krig.jl:35-37
    c1 = _load("krig_config1.npz")
    mu, var = K.exactsolve(K.OK, Variogram("gaussian", range=20.0, nugget=1e-6), c1["x"], c1["z"], c1["grid"])
    assert np.allclose(mu, c1["mu"], atol=1e-8) and np.allclose(var, c1["var"], atol=1e-8)
    f = _load("fftgs_24x20x16.npz")
    pre = fftgs.preprocess(Variogram("exponential", range=6.0, sill=1.5), (24, 20, 16), mean=0.5)
    assert np.allclose(fftgs.realize(pre, 4, 0, 2), f["z"], atol=1e-12)
    l = _load("lugs_line100.npz")
    p = lugs.preprocess(Variogram("spherical", range=10.0), fftgs.grid_centroids((100,)),
                        np.array([[0.0], [25.0], [50.0], [75.0], [100.0]]), np.array([0.0, 1.0, 0.0, 1.0, 0.0]))
    assert np.allclose(lugs.realize(p, 123, 0, 2)[0], l["y"], atol=1e-12)


def test_oracle_reproduces_idw_lwr_vectors():
    from oracle import idw_lwr as E
    g = _load("idw_lwr.npz")
    mu, sd, _ = E.idw(g["x3"], g["z3"], g["grid"], 3)
    assert np.allclose(mu, g["idw_mu"], atol=1e-13) and np.allclose(sd, g["idw_dist"], atol=1e-13)
    mu, var, st = E.lwr(g["x4"], g["z4"], g["grid"], 4)
    assert np.allclose(mu, g["lwr_mu4"], atol=1e-11) and np.allclose(var, g["lwr_var4"], atol=1e-11) and not st.any()
    # three non-collinear points: the local plane interpolates them, whatever the weights (lwr.jl:139-143)
    mu3 = g["lwr_mu3"].reshape(100, 100).T
    assert abs(mu3[24, 24] - 1.0) < 0.05 and np.all(np.isfinite(g["lwr_var3"]))
    mu, var, _ = E.lwr(g["xs"], g["zs"], g["dom"], 12)
    assert np.allclose(mu, g["lwr_mu_s"], atol=1e-11) and np.allclose(var, g["lwr_var_s"], atol=1e-11)


@pytest.mark.gpu
def test_device_reproduces_idw_lwr_vectors():
    from gss.engine import HipEngine
    g = _load("idw_lwr.npz")
    mu, sd, st = HipEngine.idw(g["x3"], g["z3"], g["grid"], 3)
    assert np.max(np.abs(mu - g["idw_mu"])) < 1e-12 and np.max(np.abs(sd - g["idw_dist"])) < 1e-12 and not st.any()
    for k, key in ((3, "3"), (4, "4")):
        mu, var, st = HipEngine.lwr(g["x4"], g["z4"], g["grid"], k)
        assert not st.any()
        assert np.max(np.abs(mu - g["lwr_mu" + key])) < 1e-8 and np.max(np.abs(var - g["lwr_var" + key])) < 1e-8
    mu, sd, _ = HipEngine.idw(g["xs"], g["zs"], g["dom"], 12, 1, 2.0)
    assert np.max(np.abs(mu - g["idw_mu_s"])) < 1e-12 and np.max(np.abs(sd - g["idw_dist_s"])) < 1e-12
    mu, var, _ = HipEngine.lwr(g["xs"], g["zs"], g["dom"], 12)
    assert np.max(np.abs(mu - g["lwr_mu_s"])) < 1e-10 and np.max(np.abs(var - g["lwr_var_s"])) < 1e-10


@pytest.mark.gpu
def test_device_reproduces_golden_vectors():
    import gss
    from gss.engine import FFTGSHandle, KrigHandle, LUGSHandle, OK, UK
    g = _load("krig_reference_2d.npz")
    vg = gss.GaussianVariogram(range=35.0, nugget=0.0)
    h = KrigHandle(vg, OK, g["x"], g["z"])
    mu, var, _ = h.predict_global(g["grid"])
    assert np.max(np.abs(mu - g["mu_global"])) < 1e-9 and np.max(np.abs(var - g["var_global"])) < 1e-9
    hl = KrigHandle(vg, OK, g["x"], g["z"], factor=False)
    mu, var, st, idx, cnt = hl.predict_knn(g["grid"], 3, return_idx=True)
    assert np.array_equal(idx, g["idx_nearest"])
    assert np.max(np.abs(mu - g["mu_nearest"])) < 1e-9 and np.max(np.abs(var - g["var_nearest"])) < 1e-9
    c1 = _load("krig_config1.npz")                                      # BASELINE config 1 (Gaussian: 1e-6)
    h1 = KrigHandle(gss.GaussianVariogram(range=20.0, nugget=1e-6), OK, c1["x"], c1["z"])
    mu, var, _ = h1.predict_global(c1["grid"])
    assert np.max(np.abs(mu - c1["mu"])) < 1e-6 and np.max(np.abs(var - c1["var"])) < 1e-6
    u = _load("krig_local_uk.npz")
    hu = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5), UK, u["x"], u["z"], degree=1, factor=False)
    mu, var, st, idx, cnt = hu.predict_knn(u["dom"], 16, return_idx=True)
    assert np.array_equal(idx, u["idx"]) and np.max(np.abs(mu - u["mu"])) < 1e-9 and np.max(np.abs(var - u["var"])) < 1e-9
    f = _load("fftgs_24x20x16.npz")
    fh = FFTGSHandle(gss.ExponentialVariogram(range=6.0, sill=1.5), (24, 20, 16), mean=0.5)
    assert np.max(np.abs(fh.spectrum() - f["F"])) < 1e-12 * f["F"].max()
    assert np.max(np.abs(fh.realize(4, 0, 2) - f["z"])) < 1e-9
    l = _load("lugs_line100.npz")
    lh = LUGSHandle(gss.SphericalVariogram(range=10.0), np.arange(100.0)[:, None] + 0.5, l["dlocs"], l["z1"])
    L22, d2 = lh.factor()
    assert np.max(np.abs(d2 - l["d2"])) < 1e-9 and np.max(np.abs(L22[50] - l["L22_row50"])) < 1e-9
    y, w = lh.realize(123, 0, 2)
    assert np.max(np.abs(y - l["y"])) < 1e-9 and np.max(np.abs(w - l["w"])) < 1e-12
