"""Global-neighbourhood kriging on the device vs the oracle.

Tolerances: well-conditioned models (Matern / exponential / spherical) 1e-9
relative on the mean and 1e-9 absolute (sill = 1) on the variance; Gaussian-variogram systems
1e-6 absolute."""
import numpy as np
import pytest

from oracle import fftgs as offt, kriging as K
from oracle.variogram import Variogram, cov_pairwise

pytestmark = pytest.mark.gpu


def _vg(kind, **kw):
    import gss
    ctor = dict(gaussian=gss.GaussianVariogram, exponential=gss.ExponentialVariogram,
                spherical=gss.SphericalVariogram, matern=gss.MaternVariogram, cubic=gss.CubicVariogram,
                pentaspherical=gss.PentasphericalVariogram)[kind]
    radii = kw.pop("radii", None)
    if radii is not None:
        return ctor(gss.MetricBall(tuple(radii)), **kw)
    if "nu" in kw:
        kw["order"] = kw.pop("nu")
    return ctor(**kw)


CASES = [("gaussian", dict(range=12.0, nugget=0.05)), ("exponential", dict(range=20.0, sill=2.0)),
         ("spherical", dict(range=15.0, nugget=0.1)), ("matern", dict(range=30.0, nu=0.5)),
         ("matern", dict(range=30.0, nu=1.5)), ("matern", dict(range=30.0, nu=2.5, sill=3.0, nugget=0.5)),
         ("cubic", dict(range=25.0)), ("pentaspherical", dict(range=25.0)),
         ("gaussian", dict(radii=(20.0, 5.0, 10.0), nugget=0.01))]


@pytest.mark.parametrize("kind,kw", CASES)
@pytest.mark.parametrize("dim", [1, 2, 3])
def test_cov_pairwise_matches_oracle(kind, kw, dim):
    from gss.engine import HipEngine
    kw = dict(kw)
    if "radii" in kw:
        kw["radii"] = kw["radii"][:dim]
        if dim == 1:
            pytest.skip("isotropic in 1-D")
    rng = np.random.default_rng(3)
    a = rng.uniform(0, 50, (77, dim))
    b = rng.uniform(0, 50, (301, dim))
    b[:5] = a[:5]                                  # exact zero lags
    got = HipEngine.cov_pairwise(_vg(kind, **kw), a, b)
    ref = cov_pairwise(Variogram(kind, **kw), a, b)
    assert np.max(np.abs(got - ref)) < 5e-15 * max(1.0, kw.get("sill", 1.0))
    sym = HipEngine.cov_pairwise(_vg(kind, **kw), a)
    assert np.array_equal(sym, sym.T)


def _run(variant, kind, kw, n, m, dim, seed, **okw):
    from gss.engine import KrigHandle
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 100, (n, dim))
    z = rng.normal(size=n)
    x0 = rng.uniform(0, 100, (m, dim))
    x0[:3] = x[:3]
    h = KrigHandle(_vg(kind, **dict(kw)), variant, x, z, mean=okw.get("mean"), degree=okw.get("degree"),
                   drift_data=okw.get("drift_data"))
    mu, var, st = h.predict_global(x0, okw.get("drift_dom"))
    h.close()
    rmu, rvar = K.exactsolve(variant, Variogram(kind, **kw), x, z, x0, mean=okw.get("mean") or 0.0,
                             degree=okw.get("degree"), drift_data=okw.get("drift_data"),
                             drift_dom=okw.get("drift_dom"))
    assert not st.any()
    return mu, var, rmu, rvar, z


@pytest.mark.parametrize("variant,okw", [(K.SK, dict(mean=0.7)), (K.OK, {}), (K.UK, dict(degree=1)),
                                         (K.UK, dict(degree=2))])
@pytest.mark.parametrize("n,m,dim", [(3, 10, 2), (50, 1000, 1), (200, 777, 2), (1000, 3000, 3), (1029, 300, 3)])
def test_global_kriging_matches_oracle(variant, okw, n, m, dim):
    if variant == K.UK and n < 12:
        pytest.skip("fewer samples than drift terms")
    mu, var, rmu, rvar, z = _run(variant, "matern", dict(range=30.0, nu=1.5), n, m, dim, seed=n + dim, **okw)
    assert np.max(np.abs(mu - rmu)) < 1e-9 * max(1.0, np.max(np.abs(rmu)))
    assert np.max(np.abs(var - rvar)) < 1e-9
    assert np.allclose(mu[:3], z[:3], atol=1e-9) and np.all(var[:3] < 1e-9)     # exact at data


def test_external_drift_matches_oracle():
    rng = np.random.default_rng(8)
    n, m = 120, 500
    dd = rng.normal(size=(n, 2))
    d0 = rng.normal(size=(m, 2))
    mu, var, rmu, rvar, _ = _run(K.EDK, "exponential", dict(range=25.0), n, m, 2, 5, drift_data=dd, drift_dom=d0)
    # x0[:3] == x[:3] but drift values differ there, so no exactness check
    assert np.max(np.abs(mu - rmu)) < 1e-9 * max(1.0, np.max(np.abs(rmu))) and np.max(np.abs(var - rvar)) < 1e-9


def test_gaussian_variogram_config1():
    """BASELINE config 1: OK, 100 2-D data -> 64x64 grid, Gaussian range 20 nugget 1e-6."""
    from gss.engine import KrigHandle
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 64, (100, 2))
    z = rng.normal(size=100)
    g = offt.grid_centroids((64, 64))
    kw = dict(range=20.0, sill=1.0, nugget=1e-6)
    h = KrigHandle(_vg("gaussian", **kw), K.OK, x, z)
    mu, var, st = h.predict_global(g)
    rmu, rvar = K.exactsolve(K.OK, Variogram("gaussian", **kw), x, z, g)
    assert np.max(np.abs(mu - rmu)) < 1e-6 and np.max(np.abs(var - rvar)) < 1e-6


def test_device_resident_inputs_match_host_path_and_chunking(monkeypatch):
    import torch
    from gss.engine import KrigHandle
    rng = np.random.default_rng(21)
    x = rng.uniform(0, 100, (300, 3))
    z = rng.normal(size=300)
    x0 = rng.uniform(0, 100, (5000, 3))
    vg = _vg("matern", range=30.0, nu=1.5)
    h = KrigHandle(vg, K.OK, x, z)
    mu, var, _ = h.predict_global(x0)
    tmu, tvar, tst = h.predict_global(torch.as_tensor(x0, device="cuda"))
    torch.cuda.synchronize()
    assert np.array_equal(tmu.cpu().numpy(), mu) and np.array_equal(tvar.cpu().numpy(), var)
    h.close()
    monkeypatch.setenv("GSS_KRIG_WS_MB", "1")                 # forces many chunks
    h2 = KrigHandle(vg, K.OK, x, z)
    mu2, var2, _ = h2.predict_global(x0)
    assert np.array_equal(mu2, mu) and np.array_equal(var2, var)


def test_not_positive_definite_is_reported():
    from gss import _lib
    from gss.engine import KrigHandle
    x = np.array([[0.0, 0.0], [0.0, 0.0], [1.0, 1.0]])      # duplicate sample, no nugget
    with pytest.raises(_lib.GSSError) as e:
        KrigHandle(_vg("gaussian", range=5.0), K.OK, x, np.zeros(3))
    assert e.value.code == _lib.ERR_NOT_POSDEF


def test_reference_2d_problem_through_solve_api():
    """test/estimation/krig.jl:22-37 through the solver front-end."""
    import gss
    data = gss.georef({"z": [1.0, 0.0, 1.0]}, [(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
    grid = gss.CartesianGrid((100, 100), (0.5, 0.5), (1.0, 1.0))
    problem = gss.EstimationProblem(data, grid, "z")
    solver = gss.KrigingSolver(("z", dict(variogram=gss.GaussianVariogram(range=35.0, nugget=0.0))))
    sol = gss.solve(problem, solver)
    Z = gss.asarray(sol, "z")
    assert abs(Z[24, 24] - 1.0) < 1e-3 and abs(Z[49, 74] - 0.0) < 1e-3 and abs(Z[74, 49] - 1.0) < 1e-3
    assert sol["z_variance"].shape == (10000,) and np.all(sol["z_variance"] >= 0)


def test_factor_state_can_be_shipped_between_handles():
    """The multi-GPU path broadcasts the factor state (W' and the dual weights) from rank 0 into handles created
    with GSS_KRIG_NO_FACTOR (bench.py --factor-broadcast).  On one GPU: copy it between two handles."""
    import torch
    from gss.engine import KrigHandle
    rng = np.random.default_rng(12)
    x = rng.uniform(0, 100, (257, 3))
    z = rng.normal(size=257)
    x0 = rng.uniform(0, 100, (1000, 3))
    vg = _vg("matern", range=30.0, nu=1.5)
    a = KrigHandle(vg, K.UK, x, z, degree=1)
    b = KrigHandle(vg, K.UK, x, z, degree=1, factor=False)
    from gss import _lib
    with pytest.raises(_lib.GSSError, match="no factor"):
        b.predict_global(x0)
    ta, tb = a.factor_tensor(), b.factor_tensor()
    assert ta.is_cuda and ta.dtype == torch.float64 and ta.shape == tb.shape
    tb.copy_(ta)                      # stands in for dist.broadcast(t, src=0) over RCCL
    torch.cuda.synchronize()
    b.adopt_factor()
    mu_a, var_a, _ = a.predict_global(x0)
    mu_b, var_b, _ = b.predict_global(x0)
    assert np.array_equal(mu_a, mu_b) and np.array_equal(var_a, var_b)


def test_edge_sizes_empty_domain_single_datum_large_system():
    from gss.engine import KrigHandle
    vg = _vg("exponential", range=25.0)
    ovg = Variogram("exponential", range=25.0)
    # a single datum: OK returns the datum everywhere with variance from the 2x2 system
    h = KrigHandle(vg, K.OK, np.array([[10.0, 10.0]]), np.array([2.5]))
    mu, var, st = h.predict_global(np.array([[10.0, 10.0], [40.0, 50.0]]))
    rmu, rvar = K.exactsolve(K.OK, ovg, np.array([[10.0, 10.0]]), np.array([2.5]), np.array([[10.0, 10.0], [40.0, 50.0]]))
    assert np.allclose(mu, [2.5, 2.5]) and np.allclose(mu, rmu) and np.allclose(var, rvar, atol=1e-12)
    # empty domain
    mu0, var0, st0 = h.predict_global(np.empty((0, 2)))
    assert mu0.shape == (0,) and var0.shape == (0,)
    h.close()
    # 3 001 x 3 001 system: several recursion levels, N1 not a multiple of the tile sizes
    rng = np.random.default_rng(3000)
    x = rng.uniform(0, 200, (3000, 3))
    z = rng.normal(size=3000)
    x0 = rng.uniform(0, 200, (513, 3))
    hb = KrigHandle(_vg("matern", range=30.0, nu=1.5), K.OK, x, z)
    mu, var, st = hb.predict_global(x0)
    rmu, rvar = K.exactsolve(K.OK, Variogram("matern", range=30.0, nu=1.5), x, z, x0)
    assert np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9
    hb.close()


def test_nested_variograms_all_paths():
    """gamma = gamma1 + 2 gamma2 + 0.5 gamma3 with per-structure anisotropy: pairwise
    covariance, global and moving-neighbourhood kriging, FFTGS spectrum and LUGS factor against the oracle."""
    import gss
    from gss.engine import FFTGSHandle, HipEngine, KrigHandle, LUGSHandle
    from oracle import fftgs as OF, lugs as OL
    from oracle.variogram import Nested
    g = gss.SphericalVariogram(range=10.0, nugget=0.1) + 2.0 * gss.ExponentialVariogram(range=30.0) \
        + 0.5 * gss.GaussianVariogram(gss.MetricBall((20.0, 5.0)))
    o = Nested([(1.0, Variogram("spherical", range=10.0, nugget=0.1)), (2.0, Variogram("exponential", range=30.0)),
                (0.5, Variogram("gaussian", radii=(20.0, 5.0)))])
    rng = np.random.default_rng(31)
    x = rng.uniform(0, 60, (150, 2))
    z = rng.normal(size=150)
    x0 = rng.uniform(0, 60, (400, 2))
    x0[:3] = x[:3]
    got = HipEngine.cov_pairwise(g, x, x0)
    assert np.max(np.abs(got - cov_pairwise(o, x, x0))) < 2e-14 and abs(got[0, 0] - 3.5) < 1e-15
    h = KrigHandle(g, K.OK, x, z)
    mu, var, _ = h.predict_global(x0)
    rmu, rvar = K.exactsolve(K.OK, o, x, z, x0)
    assert np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9
    hl = KrigHandle(g, K.UK, x, z, degree=1, factor=False)
    mu, var, st, idx, _ = hl.predict_knn(x0, 12, return_idx=True)
    rmu, rvar, rst, ridx, _ = K.approxsolve(K.UK, o, x, z, x0, 12, degree=1, return_idx=True)
    assert np.array_equal(idx, ridx) and np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9
    f = FFTGSHandle(g, (32, 24), mean=0.0)
    pre = OF.preprocess(o, (32, 24))
    assert np.max(np.abs(f.spectrum() - pre.F.ravel())) < 1e-11 * pre.F.max()
    assert np.max(np.abs(f.realize(3, 0, 1) - OF.realize(pre, 3, 0, 1))) < 1e-8
    cent = OF.grid_centroids((12, 10))
    lh = LUGSHandle(g, cent, [3, 40], [0.5, -0.5])
    p = OL.preprocess(o, cent, cent[[3, 40]], [0.5, -0.5])
    L22, d2 = lh.factor()
    assert np.max(np.abs(L22 - p.L22)) < 1e-9 and np.max(np.abs(d2 - p.d2)) < 1e-9



def test_calls_on_different_streams_are_ordered():
    """The library recycles scratch memory (buffer pool, the R workspace of the quadratic form) across calls; a call
    on another stream is chained behind the previous one on the device (gss.h, `stream`), so handles driven from
    different torch streams give the results of sequential execution."""
    import torch
    import gss
    from gss.engine import KrigHandle, OK
    rng = np.random.default_rng(77)
    xa, xb = rng.uniform(0, 100, (700, 3)), rng.uniform(0, 100, (900, 3))
    za, zb = rng.normal(size=700), rng.normal(size=900)
    x0 = torch.as_tensor(rng.uniform(0, 100, (200_000, 3)), device="cuda")
    va, vb = gss.MaternVariogram(range=30.0, order=1.5), gss.ExponentialVariogram(range=25.0)
    ha, hb = KrigHandle(va, OK, xa, za), KrigHandle(vb, OK, xb, zb)
    ref_a = [t.clone() for t in ha.predict_global(x0)]
    ref_b = [t.clone() for t in hb.predict_global(x0)]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for it in range(3):
        with torch.cuda.stream(s1):
            oa = ha.predict_global(x0)
        with torch.cuda.stream(s2):
            ob = hb.predict_global(x0)
        outs.append((oa, ob))
    torch.cuda.synchronize()
    for oa, ob in outs:
        assert all(torch.equal(p, q) for p, q in zip(oa, ref_a))
        assert all(torch.equal(p, q) for p, q in zip(ob, ref_b))
    ha.close()
    hb.close()


def test_asynchronous_fit_same_results_and_status_at_the_first_prediction():
    """GSS_KRIG_ASYNC_FIT: the constructor returns once the fit is queued on the library's fit stream, the first global
    prediction assembles its right-hand sides beside it and joins before the quadratic form -- bit-identical results;
    a covariance matrix that is not positive definite is reported by that prediction (by the constructor otherwise);
    the factor buffer (broadcast) and the batched means wait for the fit too."""
    import torch
    import gss
    from gss import _lib
    from gss.engine import KrigHandle, OK, UK
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 100, (700, 3))
    z = rng.normal(size=700)
    x0 = torch.as_tensor(rng.uniform(0, 100, (40000, 3)), device="cuda")
    vg = gss.MaternVariogram(range=30.0, order=1.5)
    for variant, kw in ((OK, {}), (UK, dict(degree=1))):
        a = KrigHandle(vg, variant, x, z, **kw).predict_global(x0)
        h = KrigHandle(vg, variant, x, z, async_fit=True, **kw)
        b = h.predict_global(x0)
        c = h.predict_global(x0)                       # the fit has been joined: an ordinary call
        assert all(torch.equal(u, v) and torch.equal(u, w) for u, v, w in zip(a, b, c))
        h.close()
    h = KrigHandle(vg, OK, x, z, async_fit=True)
    assert h.factor_tensor().numel() > 0               # waits for the fit
    zb = torch.as_tensor(rng.normal(size=(3, 700)), device="cuda")
    ref = KrigHandle(vg, OK, x, z).predict_global_batch(x0[:500], zb)
    got = KrigHandle(vg, OK, x, z, async_fit=True).predict_global_batch(x0[:500], zb)
    assert torch.equal(ref, got)
    xd = np.vstack([x[:50], x[:1]])                    # a duplicate sample: singular covariance matrix
    zd = np.r_[z[:50], z[0]]
    with pytest.raises(_lib.GSSError, match="positive definite"):
        KrigHandle(gss.GaussianVariogram(range=30.0), OK, xd, zd)
    hbad = KrigHandle(gss.GaussianVariogram(range=30.0), OK, xd, zd, async_fit=True)
    with pytest.raises(_lib.GSSError, match="positive definite"):
        hbad.predict_global(x0[:1000])
    hbad.close()


def test_host_arrays_in_pieces_equal_device_arrays():
    """Host arrays beyond 131 072 points are handed over in pieces whose transfers overlap the computation of their
    neighbours (krig.hip, HOST_PIPE_POINTS): same results as one call on device arrays, also with a last piece that is
    not full, external drifts riding along, and with the switch off."""
    import os
    import subprocess
    import sys
    import torch
    import gss
    from gss.engine import KrigHandle, OK, EDK
    rng = np.random.default_rng(12)
    x = rng.uniform(0, 100, (300, 3))
    z = rng.normal(size=300)
    m = 131072 * 2 + 4321
    x0 = rng.uniform(0, 100, (m, 3))
    vg = gss.SphericalVariogram(range=40.0, nugget=0.05)
    h = KrigHandle(vg, OK, x, z)
    dev = h.predict_global(torch.as_tensor(x0, device="cuda"))
    host = h.predict_global(x0)
    for a, b in zip(dev, host):
        assert isinstance(b, np.ndarray) and np.array_equal(a.cpu().numpy(), b)
    fd, f0 = rng.normal(size=(300, 2)), rng.normal(size=(m, 2))
    he = KrigHandle(vg, EDK, x, z, drift_data=fd)
    dev = he.predict_global(torch.as_tensor(x0, device="cuda"), torch.as_tensor(f0, device="cuda"))
    host = he.predict_global(x0, f0)
    for a, b in zip(dev, host):
        assert np.array_equal(a.cpu().numpy(), b)
    code = ("import numpy as np, gss; from gss.engine import KrigHandle, OK\n"
            "rng = np.random.default_rng(12); x = rng.uniform(0, 100, (300, 3)); z = rng.normal(size=300)\n"
            "x0 = rng.uniform(0, 100, (%d, 3))\n"
            "mu, var, st = KrigHandle(gss.SphericalVariogram(range=40.0, nugget=0.05), OK, x, z).predict_global(x0)\n"
            "np.save(%r, np.stack([mu, var]))\n")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o.npy")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, GSS_HOST_PIPELINE="0",
                   PYTHONPATH=os.pathsep.join([os.path.join(root, "geostatssolvers.jl_amd"), os.environ.get("PYTHONPATH", "")]))
        subprocess.run([sys.executable, "-c", code % (m, out)], check=True, env=env, timeout=300)
        ref = np.load(out)
    h2 = KrigHandle(vg, OK, x, z)
    mu, var, _ = h2.predict_global(x0)
    assert np.array_equal(ref[0], mu) and np.array_equal(ref[1], var)


def test_host_arrays_in_pieces_without_a_status_array():
    """The C-ABI accepts status = NULL; the piece-by-piece hand-over of host arrays then moves means and variances only."""
    import ctypes as C
    import gss
    from gss import _lib
    from gss.engine import KrigHandle, OK
    rng = np.random.default_rng(31)
    x = rng.uniform(0, 50, (200, 2))
    z = rng.normal(size=200)
    m = 131072 + 999
    x0 = np.ascontiguousarray(rng.uniform(0, 50, (m, 2)))
    h = KrigHandle(gss.ExponentialVariogram(range=15.0), OK, x, z)
    ref = h.predict_global(x0)
    mean = np.full(m, np.nan)
    var = np.full(m, np.nan)
    rc = _lib.lib().gss_krig_predict_global(h._h, x0.ctypes.data_as(C.c_void_p), None, m, mean.ctypes.data_as(C.c_void_p),
                                            var.ctypes.data_as(C.c_void_p), None, 0, None)
    assert rc == 0
    assert np.array_equal(mean, ref[0]) and np.array_equal(var, ref[1])


@pytest.mark.parametrize("variant,okw,dim", [(K.OK, {}, 2), (K.SK, dict(mean=0.4), 3), (K.UK, dict(degree=1), 2), (K.OK, {}, 1)])
def test_block_support_matches_oracle(variant, okw, dim):
    """`support = ("block", nsub)`: right-hand sides regularised over the grid cells by the midpoint rule (krig.jl:180
    passes the cell to predictprob; gss.h gss_krig_set_block_support) against the oracle's restatement, 1e-9; the block
    mean equals the average of the point estimates at the samples (weights are linear in the right-hand side)."""
    from gss.engine import KrigHandle
    from oracle.kriging import block_samples
    rng = np.random.default_rng(70 + dim)
    x = rng.uniform(0, 30, (120, dim))
    z = rng.normal(size=120)
    dims = {1: (40,), 2: (12, 10), 3: (6, 5, 4)}[dim]
    spacing = {1: (0.75,), 2: (2.5, 3.0), 3: (5.0, 6.0, 7.5)}[dim]
    cent = offt.grid_centroids(dims, (0.0,) * dim, spacing)
    gvg, ovg = _vg("matern", range=12.0, nu=1.5, nugget=0.05), Variogram("matern", range=12.0, nu=1.5, nugget=0.05)
    h = KrigHandle(gvg, variant, x, z, mean=okw.get("mean"), degree=okw.get("degree"))
    mu_pt, var_pt, _ = h.predict_global(cent)
    h.set_block_support(spacing, 3)
    mu, var, st = h.predict_global(cent)
    rmu, rvar = K.exactsolve(variant, ovg, x, z, cent, support=(spacing, 3), **{k: v for k, v in okw.items()})
    assert not st.any() and np.max(np.abs(mu - rmu)) < 1e-9 and np.max(np.abs(var - rvar)) < 1e-9
    assert np.max(np.abs(mu - mu_pt)) > 1e-6                         # it is a different estimate
    off = block_samples(dim, spacing, 3)
    pts = (cent[:5, None, :] + off[None, :, :]).reshape(-1, dim)
    h.set_block_support(None, 0)                                     # back to point support
    mu_s, _, _ = h.predict_global(pts)
    assert np.max(np.abs(mu[:5] - mu_s.reshape(5, -1).mean(axis=1))) < 1e-10
    mu_again, var_again, _ = h.predict_global(cent)
    assert np.array_equal(mu_again, mu_pt) and np.array_equal(var_again, var_pt)
    h.close()


def test_block_support_through_the_solver_and_refusals():
    import gss
    from gss import _lib
    from gss.engine import KrigHandle
    rng = np.random.default_rng(9)
    xy = rng.uniform(0, 64, (100, 2))
    z = rng.normal(size=100)
    grid = gss.CartesianGrid(64, 64)                                  # the shape of BASELINE configs[0]
    prob = gss.EstimationProblem(gss.georef({"z": z}, xy), grid, "z")
    vg = gss.SphericalVariogram(range=20.0)
    sol = gss.solve(prob, gss.KrigingSolver(("z", dict(variogram=vg, support=("block", 2)))))
    rmu, rvar = K.exactsolve(K.OK, Variogram("spherical", range=20.0), xy, z, grid.centroids(), support=((1.0, 1.0), 2))
    assert np.max(np.abs(sol["z"] - rmu)) < 1e-9 and np.max(np.abs(sol["z_variance"] - rvar)) < 1e-9
    with pytest.raises(ValueError, match="Cartesian grid"):
        gss.solve(gss.EstimationProblem(gss.georef({"z": z}, xy), gss.PointSet(xy + 0.5), "z"),
                  gss.KrigingSolver(("z", dict(variogram=vg, support="block"))))
    # moving neighbourhoods regularise the same way (krig.jl:226 hands the cell to predictprob as well): 8 and 80
    # neighbours here; tests/test_gpu_large_neighbourhoods.py takes every kernel and model path
    for k, d in ((8, 1), (80, None)):
        kw = dict(variogram=vg, support=("block", 2), maxneighbors=k)
        okw = {}
        if d is not None:
            kw["degree"], okw["degree"] = d, d
        sol = gss.solve(prob, gss.KrigingSolver(("z", kw)))
        rmu, rvar, rst = K.approxsolve(K.UK if d else K.OK, Variogram("spherical", range=20.0), xy, z, grid.centroids(), k,
                                       support=((1.0, 1.0), 2), **okw)
        assert np.max(np.abs(sol["z"] - rmu)) < 1e-9 and np.max(np.abs(sol["z_variance"] - rvar)) < 1e-9, k
    h = KrigHandle(vg, K.UK, xy, z, degree=2)
    with pytest.raises(_lib.GSSError, match="degree <= 1"):
        h.set_block_support((1.0, 1.0), 3)
    h.close()


@pytest.mark.parametrize("extra", [dict(), dict(mean=0.4), dict(degree=1), dict(degree=2)])
def test_variables_sharing_one_kriging_system_through_solve(extra):
    """`solve` with several variables on the same samples and the same variogram object: one fit, one assembly, one
    quadratic form; the other variables are a batched product with the same weights.  Means 1e-9 against a solve per
    variable and against the oracle, variances identical to the first variable's."""
    import gss
    rng = np.random.default_rng(12)
    n, m = 400, 5000
    xy = rng.uniform(0, 100, (n, 3))
    tab = {k: rng.normal(size=n) + i for i, k in enumerate("abc")}
    data = gss.georef(tab, xy)
    dom = gss.PointSet(rng.uniform(0, 100, (m, 3)))
    vg = gss.MaternVariogram(range=30.0, order=1.5, nugget=0.05)
    params = dict(variogram=vg, **extra)
    from gss import _lib
    _lib.profile_reset(); _lib.profile_enable(True)
    together = gss.solve(gss.EstimationProblem(data, dom, ("a", "b", "c")),
                         gss.KrigingSolver(("a", params), ("b", params), ("c", params)))
    _lib.profile_enable(False)
    assert _lib.profile_read("krig_quadform")[1] == 1                       # one quadratic-form pass for three variables
    ovg = Variogram("matern", range=30.0, nu=1.5, nugget=0.05)
    variant = K.SK if "mean" in extra else (K.UK if "degree" in extra else K.OK)
    for v in "abc":
        alone = gss.solve(gss.EstimationProblem(data, dom, v), gss.KrigingSolver((v, params)))
        assert np.max(np.abs(together[v] - alone[v])) < 1e-9
        assert np.max(np.abs(together[f"{v}_variance"] - alone[f"{v}_variance"])) < 1e-12
        rmu, rvar = K.exactsolve(variant, ovg, xy, tab[v], dom.coords[:300], mean=extra.get("mean", 0.0),
                                 degree=extra.get("degree"))[:2]
        assert np.max(np.abs(together[v][:300] - rmu)) < 1e-9 and np.max(np.abs(together[f"{v}_variance"][:300] - rvar)) < 1e-9
