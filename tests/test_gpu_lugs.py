"""LUGS on the device vs the oracle (lu.jl:76-224 restated with LAPACK).

Tolerances: factors L22 / d2 and realisations from supplied normals 1e-9 absolute (sill 1,
spherical / exponential models, cond(C) <~ 1e4); Gaussian-variogram cases are property-checked only."""
import numpy as np
import pytest

from oracle import fftgs as offt, lugs as O, philox
from oracle.variogram import Variogram, cov_pairwise

pytestmark = pytest.mark.gpu


def _mk(kind, **kw):
    import gss
    ctor = dict(gaussian=gss.GaussianVariogram, exponential=gss.ExponentialVariogram,
                spherical=gss.SphericalVariogram)[kind]
    radii = kw.pop("radii", None)
    return ctor(gss.MetricBall(tuple(radii)), **kw) if radii is not None else ctor(**kw)


@pytest.mark.parametrize("dims,ndata", [((100,), 5), ((100,), 0), ((30, 20), 40), ((12, 10, 8), 100), ((70, 70), 0),
                                        ((70, 60), 1500)])   # two ragged data panels, three ragged simulation panels
def test_factor_and_realisations_match_oracle(dims, ndata):
    from gss.engine import LUGSHandle
    cent = offt.grid_centroids(dims)
    N = cent.shape[0]
    rng = np.random.default_rng(N + ndata)
    kw = dict(range=0.2 * max(dims) + 3.0, nugget=0.02)
    dlocs = np.sort(rng.choice(N, ndata, replace=False)) if ndata else np.empty(0, dtype=np.int64)
    z1 = rng.normal(size=ndata)
    h = LUGSHandle(_mk("spherical", **kw), cent, dlocs, z1, mean=0.75)
    ovg = Variogram("spherical", **kw)
    p = O.preprocess(ovg, cent, cent[dlocs] if ndata else None, z1 if ndata else None, mean=0.75)
    assert np.array_equal(p.dlocs, dlocs) and h.ns == p.slocs.size
    L22, d2 = h.factor()
    assert np.array_equal(np.triu(L22, 1), np.zeros_like(L22))
    assert np.max(np.abs(L22 - p.L22)) < 1e-9 and np.max(np.abs(d2 - p.d2)) < 1e-9
    w = rng.normal(size=(3, h.ns))
    y, wout = h.realize(0, 0, 3, noise=w)
    assert np.array_equal(wout, w)
    for r in range(3):
        ref, _ = O.lusim(p, w[r])
        assert np.max(np.abs(y[r] - ref)) < 1e-9
        assert np.array_equal(y[r][dlocs], z1)                       # lu.jl:217
    # device Philox normals: same draws as the oracle up to libm rounding
    y2, w2 = h.realize(123, 5, 2)
    ry, rw = O.realize(p, 123, 5, 2)
    assert np.max(np.abs(w2 - rw)) < 1e-12 and np.max(np.abs(y2 - ry)) < 1e-9
    h.close()


def test_even_panel_count_with_a_ragged_last_panel():
    """ns = 3 969 = 3 x 1 024 + 897: four panels, the last one ragged (not a multiple of 16) -- the case in which the
    padded copies of the last panel's factor-and-inverse need more scratch than a full panel while the helper stream
    still reads the panel column behind it (dense_la.hip, panel_scratch_doubles).  Against the oracle, 1e-9."""
    from gss.engine import LUGSHandle
    cent = offt.grid_centroids((63, 63))
    kw = dict(range=9.0, nugget=0.02)
    for _ in range(2):          # twice: the second run finds the pool's blocks in a different state
        h = LUGSHandle(_mk("spherical", **kw), cent, [], [], 0.0)
        L22, _ = h.factor()
        h.close()
        ref = np.linalg.cholesky(cov_pairwise(Variogram("spherical", **kw), cent))
        assert np.max(np.abs(L22 - ref)) < 1e-9


def test_cosimulation_matches_oracle():
    from gss.engine import LUGSHandle
    cent = offt.grid_centroids((500,))                                # test/simulation/lu.jl:33-45
    h1 = LUGSHandle(_mk("spherical", range=10.0), cent, [], [], 0.0)
    h2 = LUGSHandle(_mk("exponential", range=10.0), cent, [], [], 0.0)
    p1 = O.preprocess(Variogram("spherical", range=10.0), cent)
    p2 = O.preprocess(Variogram("exponential", range=10.0), cent)
    y1, w1 = h1.realize(7, 0, 4)
    y2, w2 = h2.realize(8, 0, 4, rho=0.95, w1=w1)
    ry1, rw1 = O.realize(p1, 7, 0, 4)
    ry2, rw2 = O.realize(p2, 8, 0, 4, rho=0.95, w1=rw1)
    assert np.max(np.abs(y1 - ry1)) < 1e-9 and np.max(np.abs(y2 - ry2)) < 1e-9
    assert np.max(np.abs(w2 - rw2)) < 1e-12                            # w_out is the raw second draw (lu.jl:223)


def test_statistics_and_device_outputs():
    import torch
    from gss.engine import LUGSHandle
    cent = offt.grid_centroids((16, 16))
    vg = Variogram("exponential", range=6.0, sill=2.0)
    h = LUGSHandle(_mk("exponential", range=6.0, sill=2.0), cent, [], [], mean=3.0)
    y, _ = h.realize(11, 0, 20000, device=True)
    torch.cuda.synchronize()
    C = cov_pairwise(vg, cent)
    emp = torch.cov(y.T).cpu().numpy()
    assert np.max(np.abs(emp - C)) < 0.08 and abs(float(y.mean()) - 3.0) < 0.02
    L22, _ = h.factor()
    assert np.max(np.abs(L22 @ L22.T - C)) < 1e-10


def test_reference_cases_through_solve_api():
    import gss
    S = gss.georef({"z": [0.0, 1.0, 0.0, 1.0, 0.0]}, np.array([[0.0], [25.0], [50.0], [75.0], [100.0]]))
    D = gss.CartesianGrid(100)
    # conditional (test/simulation/lu.jl:8-16)
    sol = gss.solve(gss.SimulationProblem(S, D, "z", 2),
                    gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0))), rng=123))
    assert len(sol) == 2 and sol[0].z.shape == (100,)
    cent = D.centroids()
    near = [int(np.argmin(np.abs(cent[:, 0] - c))) for c in (0.0, 25.0, 50.0, 75.0, 100.0)]
    assert np.allclose(sol[1].z[near], [0.0, 1.0, 0.0, 1.0, 0.0])
    # unconditional (lu.jl:18-27)
    sol = gss.solve(gss.SimulationProblem(D, ("z", float), 2),
                    gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0))), rng=123))
    assert np.all(np.isfinite(sol["z"][0]))
    # co-simulation, as written (lu.jl:29-39): the Gaussian model with its nugget left at 0
    D5 = gss.CartesianGrid(500)
    solver = gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0))),
                      ("y", dict(variogram=gss.GaussianVariogram(range=10.0))),
                      (("z", "y"), dict(correlation=0.95)), rng=123)
    sol = gss.solve(gss.SimulationProblem(D5, (("z", float), ("y", float)), 1), solver)
    assert sol[0].z.shape == (500,) and sol[0].y.shape == (500,) and np.all(np.isfinite(sol[0].y))
    # custom factorization (test/simulation/lu.jl:66-76): both run, as in the reference
    for fact in ("lu", "cholesky"):
        sol = gss.solve(gss.SimulationProblem(S, D, "z", 1),
                        gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0), factorization=fact)), rng=123))
        assert np.all(np.isfinite(sol[0].z)) and np.allclose(sol[0].z[near], [0.0, 1.0, 0.0, 1.0, 0.0])
    with pytest.warns(UserWarning, match="mean can only be specified in unconditional simulation"):
        gss.solve(gss.SimulationProblem(S, D, "z", 1),
                  gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=10.0), mean=1.0)), rng=1))


@pytest.mark.parametrize("ball", [None, (20.0, 5.0)])
def test_reference_gaussian_100x100_cases_run_as_written(ball):
    """test/simulation/lu.jl:41-52 (`GaussianVariogram(range=10.0)`) and :54-64 (`GaussianVariogram(MetricBall((20.,5.)))`)
    on `CartesianGrid(100, 100)`, three realisations, nugget left at 0 -- as written.  They run because the Gaussian model
    is evaluated with `nugget + 1e-6` (gss/variograms.py) exactly as in the oracle.  cond(C) ~ 1e10, so the
    factors are compared through what is backward stable: rows of L22 L22' against C to 1e-9, and the realisations
    against the oracle's (same Philox normals, LAPACK factor) to 1e-4 absolute."""
    import gss
    from gss.engine import LUGSHandle
    grid = gss.CartesianGrid(100, 100)
    gvg = gss.GaussianVariogram(range=10.0) if ball is None else gss.GaussianVariogram(gss.MetricBall(ball))
    ovg = Variogram("gaussian", range=10.0) if ball is None else Variogram("gaussian", radii=ball)
    sol = gss.solve(gss.SimulationProblem(grid, ("z", float), 3), gss.LUGS(("z", dict(variogram=gvg)), rng=123))
    zs = np.stack(sol["z"])
    assert zs.shape == (3, 10000) and np.all(np.isfinite(zs)) and 0.3 < zs.var() < 2.5
    cent = grid.centroids()
    p = O.preprocess(ovg, cent)
    ref, _ = O.realize(p, 123, 0, 3)
    assert np.max(np.abs(zs - ref)) < 1e-4
    h = LUGSHandle(gvg, cent, [], [], 0.0)
    L22, _ = h.factor()
    h.close()
    for i in np.random.default_rng(1).integers(0, 10000, 6):
        assert np.max(np.abs(L22[i] @ L22.T - cov_pairwise(ovg, cent[i:i + 1], cent)[0])) < 1e-9


def test_gaussian_without_the_regularisation_is_reported_not_silently_wrong():
    """`regularize=False` restores the bare model: the same 100 x 100 lattice is then numerically singular and the
    device says so (GSS_ERR_NOT_POSDEF), as LAPACK does in the oracle."""
    import gss
    from gss import _lib
    with pytest.raises(_lib.GSSError) as e:
        gss.solve(gss.SimulationProblem(gss.CartesianGrid(100, 100), ("z", float), 1),
                  gss.LUGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0, regularize=False))), rng=123))
    assert e.value.code == _lib.ERR_NOT_POSDEF


def test_2d_100x100_runs_with_properties():
    """The reference's largest LUGS case size (N = 10^4): L22 L22' == C checked through random probes."""
    import torch
    from gss.engine import LUGSHandle
    cent = offt.grid_centroids((100, 100))
    vg = Variogram("spherical", range=10.0)
    h = LUGSHandle(_mk("spherical", range=10.0), cent, [], [], 0.0)
    L22, d2 = h.factor()
    rows = np.random.default_rng(0).integers(0, h.ns, 6)
    for i in rows:
        assert np.max(np.abs(L22[i] @ L22.T - cov_pairwise(vg, cent[i:i + 1], cent)[0])) < 1e-10
    assert not d2.any()
    y, _ = h.realize(3, 0, 3)
    assert y.shape == (3, 10000) and np.all(np.isfinite(y)) and abs(y.var() - 1.0) < 0.3   # 3 correlated fields: loose sanity bound
    h.close()


def test_batched_global_prediction_matches_loop():
    from gss.engine import KrigHandle
    import gss
    from oracle import kriging as K
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 100, (150, 2))
    x0 = rng.uniform(0, 100, (700, 2))
    zb = rng.normal(size=(5, 150))
    for variant, mean in ((K.SK, 0.3), (K.OK, None)):
        h = KrigHandle(gss.ExponentialVariogram(range=25.0), variant, x, zb[0], mean=mean)
        out = h.predict_global_batch(x0, zb)
        h.close()
        for b in range(5):
            ref, _ = K.exactsolve(variant, Variogram("exponential", range=25.0), x, zb[b], x0, mean=mean or 0.0)
            assert np.max(np.abs(out[b] - ref)) < 1e-9


@pytest.mark.parametrize("variant,okw,dim,nb", [(2, dict(degree=1), 3, 37), (1, {}, 1, 16), (0, dict(mean=-0.4), 2, 17)])
def test_batched_means_more_vectors_than_one_pass_and_drifts(variant, okw, dim, nb):
    """The fused means-only kernel takes sixteen data vectors per pass: 37 vectors need three passes; universal kriging
    adds the drift rows; a nested model goes through the generic instantiation."""
    from gss.engine import KrigHandle
    import gss
    from oracle import kriging as K
    from oracle.variogram import Nested
    rng = np.random.default_rng(40 + nb)
    x = rng.uniform(0, 100, (120, dim))
    if dim == 1:
        x = (np.linspace(0, 100, 120) + rng.uniform(-0.2, 0.2, 120))[:, None]
    x0 = rng.uniform(0, 100, (1001, dim))
    zb = rng.normal(size=(nb, 120))
    gv = 0.7 * gss.SphericalVariogram(range=40.0, nugget=0.02) + 0.3 * gss.ExponentialVariogram(range=15.0)
    ov = Nested([(0.7, Variogram("spherical", range=40.0, nugget=0.02)), (0.3, Variogram("exponential", range=15.0))])
    kvar = {0: K.SK, 1: K.OK, 2: K.UK}[variant]
    h = KrigHandle(gv, kvar, x, zb[0], mean=okw.get("mean"), degree=okw.get("degree"))
    out = h.predict_global_batch(x0, zb)
    h.close()
    assert out.shape == (nb, 1001)
    for b in sorted({0, 15, min(16, nb - 1), nb - 1}):
        ref, _ = K.exactsolve(kvar, ov, x, zb[b], x0, mean=okw.get("mean") or 0.0, degree=okw.get("degree"))
        assert np.max(np.abs(out[b] - ref)) < 1e-9


def test_state_broadcast_adopt():
    """lu.jl:76 runs once: a handle created with GSS_LUGS_NO_FACTOR refuses to realise, receives (L22, d2) from
    another handle (what torch.distributed.broadcast does between ranks) and then reproduces its realisations."""
    import gss
    from gss._lib import GSSError
    from gss.engine import LUGSHandle
    cent = offt.grid_centroids((30, 17))
    dlocs = np.sort(np.random.default_rng(3).permutation(30 * 17)[:40])
    z1 = np.random.default_rng(4).normal(size=40)
    vg = gss.SphericalVariogram(range=9.0, nugget=0.05)
    a = LUGSHandle(vg, cent, dlocs, z1)
    b = LUGSHandle(vg, cent, dlocs, z1, factor=False)
    with pytest.raises(GSSError, match="no factor"):
        b.realize(1, 0, 1)
    tb = b.state_tensor()
    tb.copy_(a.state_tensor())
    b.adopt_state()
    ya, wa = a.realize(7, 2, 3)
    yb, wb = b.realize(7, 2, 3)
    assert np.array_equal(ya, yb) and np.array_equal(wa, wb)
    a.close()
    b.close()


@pytest.mark.parametrize("dims,ndata", [((100,), 5), ((100,), 0), ((24, 18), 30), ((40, 33), 200)])
def test_lu_factorization_matches_oracle(dims, ndata):
    """`factorization = lu` (lu.jl:70,107): L11 and L22 are `lu(Symmetric(.)).L`, the unit lower factor of LAPACK's
    partial-pivot LU with the permutation dropped -- reproduced as the reference computes it (oracle: scipy's getrf).
    Same tolerance as the Cholesky path: factors and realisations 1e-9."""
    from gss.engine import LUGSHandle
    import gss
    cent = offt.grid_centroids(dims)
    N = cent.shape[0]
    rng = np.random.default_rng(N + ndata)
    dlocs = np.sort(rng.permutation(N)[:ndata])
    z1 = rng.normal(size=ndata)
    kw = dict(range=0.3 * dims[0], nugget=0.05)
    h = LUGSHandle(gss.SphericalVariogram(**kw), cent, dlocs, z1, factorization="lu")
    p = O.preprocess(Variogram("spherical", **kw), cent, cent[dlocs] if ndata else None, z1 if ndata else None,
                     factorization="lu")
    l22, d2 = h.factor()
    assert np.allclose(np.diag(l22), 1.0) and np.max(np.abs(np.triu(l22, 1))) == 0.0
    assert np.max(np.abs(l22 - p.L22)) < 1e-9 and np.max(np.abs(d2 - p.d2)) < 1e-9
    ns = N - ndata
    noise = np.random.default_rng(1).normal(size=(2, ns))
    y, _ = h.realize(0, 0, 2, noise=noise)
    for r in range(2):
        ref, _ = O.lusim(p, noise[r])
        assert np.max(np.abs(y[r] - ref)) < 1e-9
    h.close()
