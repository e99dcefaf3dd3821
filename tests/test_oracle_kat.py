"""Known-answer tests that pin the oracle independently of any implementation."""
import numpy as np
import pytest

from oracle import fftgs, kriging as K, lugs, philox
from oracle.variogram import Variogram, cov_h, cov_pairwise, gamma_h

RNG = np.random.default_rng(11)


def test_philox_random123_known_answers():
    kats = [((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for c, k, want in kats:
        o = philox.philox4x32_10(np.array([c[0]], dtype=np.uint32), c[1], c[2], c[3], k[0], k[1])
        assert " ".join("%08x" % int(x[0]) for x in o) == want


def test_philox_uniform_range_and_moments():
    u = philox.uniform(7, 3, 200001)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
    w = philox.normal(7, 3, 200000)
    assert abs(w.mean()) < 1e-2 and abs(w.var() - 1) < 1e-2
    assert not np.array_equal(philox.uniform(7, 4, 16), u[:16])


def test_variogram_shapes():
    h = np.array([0.0, 1e-9, 5.0, 10.0, 30.0])
    for kind in ("gaussian", "exponential", "spherical", "matern", "cubic", "pentaspherical"):
        vg = Variogram(kind, sill=2.0, nugget=0.25, range=10.0, nu=1.5)
        g = gamma_h(vg, h)
        assert g[0] == 0.0 and abs(g[1] - 0.25) < 1e-6          # nugget jump at h > 0
        assert np.all(np.diff(g) >= -1e-12) and g[-1] <= 2.0 + 1e-12
        assert np.allclose(cov_h(vg, h), 2.0 - g)
    # practical range: ~95% of the sill at h = range
    for kind in ("gaussian", "exponential"):
        assert abs(gamma_h(Variogram(kind, range=10.0, regularize=False), 10.0) - (1 - np.exp(-3))) < 1e-15
    # the Gaussian model is evaluated with nugget + 1e-6 ([RECALL] Variography); nothing else is
    g = Variogram("gaussian", range=10.0, sill=2.0, nugget=0.25)
    assert g.nugget == 0.25 and abs(gamma_h(g, 1e-12) - (0.25 + 1e-6)) < 1e-15 and gamma_h(g, 0.0) == 0.0
    assert abs(gamma_h(g, 10.0) - ((2.0 - 0.25 - 1e-6) * (1 - np.exp(-3)) + 0.25 + 1e-6)) < 1e-15
    assert gamma_h(Variogram("exponential", range=10.0), 1e-12) < 1e-11
    assert gamma_h(Variogram("spherical", range=10.0), 10.0) == 1.0


def test_matern_closed_forms_match_bessel_form():
    h = np.linspace(0.01, 40, 200)
    for nu in (0.5, 1.5, 2.5):
        a = gamma_h(Variogram("matern", range=10.0, nu=nu), h)
        b = gamma_h(Variogram("matern", range=10.0, nu=nu + 1e-12), h)    # generic Bessel branch
        assert np.max(np.abs(a - b)) < 1e-9


def test_anisotropic_ball_distance():
    vg = Variogram("gaussian", radii=(20.0, 5.0), regularize=False)
    c = cov_pairwise(vg, np.array([[0.0, 0.0]]), np.array([[20.0, 0.0], [0.0, 5.0], [0.0, 0.0]]))
    assert abs(c[0, 0] - np.exp(-3)) < 1e-15 and abs(c[0, 1] - np.exp(-3)) < 1e-15 and c[0, 2] == 1.0


@pytest.mark.parametrize("variant,kw", [(K.SK, dict(mean=0.3)), (K.OK, {}), (K.UK, dict(degree=1)),
                                        (K.UK, dict(degree=2))])
def test_kriging_is_exact_at_data_locations(variant, kw):
    x = RNG.uniform(0, 50, (40, 2))
    z = RNG.normal(size=40)
    vg = Variogram("spherical", range=25.0)
    mu, var = K.exactsolve(variant, vg, x, z, x, **kw)
    assert np.allclose(mu, z, atol=1e-9) and np.allclose(var, 0, atol=1e-9)


def test_ok_weights_sum_to_one_and_two_point_closed_form():
    vg = Variogram("exponential", range=10.0)
    x = np.array([[0.0], [4.0]])
    z = np.array([1.0, 3.0])
    mu, var = K.exactsolve(K.OK, vg, x, z, np.array([[1.0]]))
    c = lambda h: float(cov_h(vg, h))
    # closed form: lambda1 = 1/2 + (c01 - c02) / (2 (c0 - c12))
    l1 = 0.5 + (c(1.0) - c(3.0)) / (2 * (c(0.0) - c(4.0)))
    assert abs(mu[0] - (l1 * 1.0 + (1 - l1) * 3.0)) < 1e-12
    # adding a constant to the data shifts the OK estimate by that constant (weights sum to 1)
    mu2, _ = K.exactsolve(K.OK, vg, x, z + 10.0, np.array([[1.0]]))
    assert abs(mu2[0] - mu[0] - 10.0) < 1e-12


def test_sk_single_datum_closed_form():
    vg = Variogram("gaussian", range=10.0, sill=2.0)
    mu, var = K.exactsolve(K.SK, vg, np.array([[0.0, 0.0]]), np.array([5.0]), np.array([[3.0, 4.0]]), mean=1.0)
    c0, ch = 2.0, float(cov_h(vg, 5.0))
    assert abs(mu[0] - (1.0 + ch / c0 * 4.0)) < 1e-12 and abs(var[0] - (c0 - ch * ch / c0)) < 1e-12


def test_uk_reproduces_affine_field():
    x = RNG.uniform(0, 100, (30, 3))
    f = lambda p: 2.0 + 0.5 * p[:, 0] - 0.25 * p[:, 1] + 0.1 * p[:, 2]
    x0 = RNG.uniform(0, 100, (20, 3))
    mu, _ = K.exactsolve(K.UK, Variogram("matern", range=30.0, nu=1.5), x, f(x), x0, degree=1)
    assert np.allclose(mu, f(x0), atol=1e-8)


def test_moving_neighbourhood_with_all_neighbours_equals_global():
    x = RNG.uniform(0, 50, (25, 2))
    z = RNG.normal(size=25)
    x0 = RNG.uniform(0, 50, (15, 2))
    vg = Variogram("exponential", range=20.0)
    a, av = K.exactsolve(K.OK, vg, x, z, x0)
    b, bv, st = K.approxsolve(K.OK, vg, x, z, x0, 25)
    assert np.allclose(a, b, atol=1e-10) and np.allclose(av, bv, atol=1e-10) and not st.any()


def test_knn_order_ties_and_ball():
    x = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [3.0, 0.0]])
    idx, cnt = K.knn_search(x, np.array([[0.0, 0.0]]), 4)
    assert idx[0].tolist() == [0, 1, 2, 3] and cnt[0] == 4            # equal distances -> ascending index
    idx, cnt = K.knn_search(x, np.array([[0.0, 0.0]]), 5, radius=1.0)
    assert cnt[0] == 4 and idx[0, 4] == -1                             # d <= r inclusive
    mu, var, st = K.approxsolve(K.OK, Variogram("gaussian", range=5.0, nugget=0.1), x, np.arange(5.0),
                                np.array([[10.0, 10.0]]), 3, minneighbors=1, radius=2.0)
    assert st[0] == 1 and np.isnan(mu[0])                              # krig.jl:213-214


def test_fftgs_invariants():
    vg = Variogram("exponential", range=6.0, sill=2.5)
    pre = fftgs.preprocess(vg, (32, 24), mean=1.5)
    assert pre.F.flat[0] == 0.0
    # |fft(fftshift(C))| does not depend on where C is placed (property 1)
    cent = fftgs.grid_centroids((32, 24))
    C0 = cov_h(vg, np.linalg.norm(cent - cent[0], axis=1)).reshape(24, 32)
    lagx = np.minimum(np.arange(32), 32 - np.arange(32))
    lagy = np.minimum(np.arange(24), 24 - np.arange(24))
    Cw = cov_h(vg, np.hypot(lagx[None, :], lagy[:, None]))
    Fw = np.sqrt(np.abs(np.fft.fftn(Cw)))
    Fw.flat[0] = 0
    assert np.allclose(pre.F, Fw, atol=1e-12) and C0.shape == Cw.shape
    z = fftgs.realize(pre, 4, 0, 3)
    N = 32 * 24
    for r in range(3):
        zc = z[r] - 1.5
        assert abs(np.sum(zc * zc) / (N - 1) - 2.5) < 1e-10           # fft.jl:169-170
        amp = np.abs(np.fft.fftn(zc.reshape(24, 32)))
        k = amp / np.maximum(pre.F, 1e-300)
        sel = pre.F > 1e-8
        assert np.allclose(k[sel], k[sel][0], rtol=1e-8)               # |FFT(Z - mu)| proportional to F
    # Parseval: the rescale is the same constant for every realisation (property 2)
    s = np.sqrt(2.5 * N * (N - 1) / np.sum(pre.F ** 2))
    U = philox.uniform(4, 0, N).reshape(24, 32)
    P = pre.F * np.exp(1j * np.angle(np.fft.fftn(U)))
    assert np.allclose(z[0] - 1.5, s * np.real(np.fft.ifftn(P)).ravel(), atol=1e-12)


def test_fftgs_conditional_honours_data_at_cells():
    vg = Variogram("gaussian", range=10.0)
    xd = np.array([[25.0, 25.0], [50.0, 75.0], [75.0, 50.0]]) * 0.3
    zd = np.array([1.0, -1.0, 1.0])
    pre = fftgs.preprocess(vg, (30, 30), data_coords=xd, data_vals=zd)
    z = fftgs.solvesingle(pre, philox.uniform(1, 0, 900))
    # the conditioning data sit at cell centroids in solvesingle but at their own coordinates in
    # preprocess (fft.jl:112 vs :180-184), so the match is approximate by the cell offset only
    assert np.max(np.abs(z[pre.dinds] - zd)) < 0.2


def test_lugs_invariants():
    vg = Variogram("spherical", range=10.0)
    cent = fftgs.grid_centroids((60,))
    xd = np.array([[0.0], [25.0], [50.0]])
    zd = np.array([0.0, 1.0, 0.0])
    p = lugs.preprocess(vg, cent, xd, zd)
    C = cov_pairwise(vg, cent)
    d, s = p.dlocs, p.slocs
    schur = C[np.ix_(s, s)] - C[np.ix_(s, d)] @ np.linalg.solve(C[np.ix_(d, d)], C[np.ix_(d, s)])
    assert np.allclose(p.L22 @ p.L22.T, schur, atol=1e-10)
    assert np.allclose(p.d2, C[np.ix_(s, d)] @ np.linalg.solve(C[np.ix_(d, d)], p.z1), atol=1e-10)
    y, w = lugs.realize(p, 3, 0, 4)
    assert np.array_equal(y[:, d], np.tile(p.z1, (4, 1)))              # lu.jl:217
    # unconditional: empirical covariance -> C
    pu = lugs.preprocess(vg, cent[:12], mean=2.0)
    yu, _ = lugs.realize(pu, 5, 0, 20000)
    emp = np.cov(yu.T)
    assert np.max(np.abs(emp - C[:12, :12])) < 0.05 and abs(yu.mean() - 2.0) < 0.03
    # co-simulation: correlation of the driving normals
    y1, w1 = lugs.realize(pu, 9, 0, 4000)
    y2, w2 = lugs.realize(pu, 9, 0, 4000, var_index=1, rho=0.95, w1=w1)
    assert abs(np.corrcoef(y1[:, 3], y2[:, 3])[0, 1] - 0.95) < 0.02


def test_nested_variogram_is_the_weighted_sum():
    from oracle.variogram import Nested
    a = Variogram("spherical", range=10.0, nugget=0.1)
    b = Variogram("exponential", range=30.0, radii=None)
    c = Variogram("gaussian", radii=(20.0, 5.0))
    nv = Nested([(1.0, a), (2.0, b), (0.5, c)])
    assert abs(nv.sill - 3.5) < 1e-15 and abs(nv.nugget - 0.1) < 1e-15
    x = RNG.uniform(0, 40, (7, 2))
    y = RNG.uniform(0, 40, (9, 2))
    y[0] = x[0]
    ref = cov_pairwise(a, x, y) + 2.0 * cov_pairwise(b, x, y) + 0.5 * cov_pairwise(c, x, y)
    assert np.allclose(cov_pairwise(nv, x, y), ref, atol=1e-15) and abs(cov_pairwise(nv, x, y)[0, 0] - 3.5) < 1e-15
    mu, var = K.exactsolve(K.OK, nv, x, RNG.normal(size=7), x)
    assert np.all(var < 1e-9)                                    # still exact at the data (zero lag = total sill)


def test_block_support_kats():
    """Block support (krig.jl:180 hands the grid cell to predictprob): implementation-independent facts of
    the regularised system.  (1) The kriging weights are linear in the right-hand side, so the block mean equals the
    average of the point means at the sample points -- exactly; (2) block -> point as the cell shrinks; (3) the block
    variance is the point formula with C(V, V) in place of the sill and is never above the average point variance."""
    from oracle.kriging import block_samples
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 20, (25, 2))
    z = rng.normal(size=25)
    cent = rng.uniform(2, 18, (7, 2))
    for variant, kw, vg in ((K.OK, {}, Variogram("spherical", range=8.0, nugget=0.1)),
                            (K.SK, dict(mean=0.3), Variogram("exponential", range=6.0)),
                            (K.UK, dict(degree=1), Variogram("matern", range=9.0, nu=1.5))):
        cell, nsub = (1.0, 2.0), 3
        mu_b, var_b = K.exactsolve(variant, vg, x, z, cent, support=(cell, nsub), **kw)
        off = block_samples(2, cell, nsub)
        assert off.shape == (9, 2) and np.allclose(off.mean(axis=0), 0.0)
        pts = (cent[:, None, :] + off[None, :, :]).reshape(-1, 2)
        mu_p, var_p = K.exactsolve(variant, vg, x, z, pts, **kw)
        assert np.max(np.abs(mu_b - mu_p.reshape(7, 9).mean(axis=1))) < 1e-12          # (1)
        assert np.all(var_b <= var_p.reshape(7, 9).mean(axis=1) + 1e-12)                # (3)
        mu_s, var_s = K.exactsolve(variant, vg, x, z, cent, support=((1e-7, 1e-7), 2), **kw)
        mu_0, var_0 = K.exactsolve(variant, vg, x, z, cent, **kw)
        tol = 1e-5 if vg.nugget == 0 else np.inf     # with a nugget C(V, V) -> sill - nugget, not sill: only the mean converges
        assert np.max(np.abs(mu_s - mu_0)) < 1e-6 and np.max(np.abs(var_s - var_0)) < tol   # (2)


def test_float_division_trick_of_the_generic_fft_passes():
    """csrc/fftgs_generic.h, `g_div`: a / b as trunc((a + 1/2) * (1 / b)) in single precision, for every operand pair a pass
    can produce (items below 4 096, line counts and strides up to 2 048).  IEEE single precision on the host is what the
    device computes (no fast-math): exhaustive."""
    a = np.arange(0, 4096, dtype=np.int32)
    for b in range(1, 2049):
        inv = np.float32(1.0) / np.float32(b)
        q = ((a.astype(np.float32) + np.float32(0.5)) * inv).astype(np.int32)
        assert np.array_equal(q, a // b), b
