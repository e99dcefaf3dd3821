"""SGS on the device (gss_sgs_create / gss_sgs_realize) vs the oracle's per-realisation path loop
(seq.jl:76-141).  Neighbour lists bit-exact; realisations 1e-9 (well-conditioned models: the recursion
propagates rounding along the path); conditioning cells exact."""
import numpy as np
import pytest

from oracle import fftgs as offt, philox, sgs as S
from oracle.variogram import Variogram, cov_h

pytestmark = pytest.mark.gpu


def _vg(kind, **kw):
    import gss
    ctor = dict(spherical=gss.SphericalVariogram, exponential=gss.ExponentialVariogram,
                matern=gss.MaternVariogram)[kind]
    okw = dict(kw)
    if "nu" in kw:
        kw = dict(kw)
        kw["order"] = kw.pop("nu")
    return ctor(**kw), Variogram(kind, **okw)


CASES = [
    # dims, variogram, mean, k, minneighbors, ball, ndata, random path
    ((40, 30), ("spherical", dict(range=9.0, sill=1.3)), 0.5, 10, 1, dict(radius=8.0), 6, False),
    ((300,), ("exponential", dict(range=12.0)), 0.0, 6, 1, {}, 4, False),
    ((12, 10, 8), ("matern", dict(range=6.0, nu=1.5)), -1.0, 16, 2, {}, 9, True),
    ((35, 35), ("spherical", dict(range=7.0, nugget=0.1)), 0.0, 12, 3, dict(radii=(9.0, 4.0)), 0, True),
    ((9, 7), ("exponential", dict(range=4.0)), 2.0, 63, 1, {}, 2, False),           # k == N: everything simulated
]


@pytest.mark.parametrize("mask_after", [False, True])
@pytest.mark.parametrize("dims,vgspec,mean,k,nmin,ball,nd,rpath", CASES)
def test_realisations_match_oracle(dims, vgspec, mean, k, nmin, ball, nd, rpath, mask_after):
    """Both readings of `search!(..., mask=simulated)` (seq.jl:105): the k nearest among the simulated cells, and the
    k nearest cells of the whole domain filtered by the mask (GSS_SGS_MASK_AFTER_SEARCH, the front-ends' default)."""
    from gss.engine import SGSHandle
    gvg, ovg = _vg(vgspec[0], **vgspec[1])
    cent = offt.grid_centroids(dims)
    N = cent.shape[0]
    rng = np.random.default_rng(N + k)
    dl = np.sort(rng.choice(N, nd, replace=False)) if nd else np.empty(0, dtype=np.int64)
    zd = rng.normal(size=nd)
    path = rng.permutation(N) if rpath else None
    h = SGSHandle(gvg, cent, path, dl, zd, mean, k, nmin, ball.get("radius"), ball.get("radii"),
                  mask_after_search=mask_after)
    z = h.realize(42, 3, 3)
    ref = S.realize(ovg, mean, cent, path, dl, zd, 42, 3, 3, maxneighbors=k, minneighbors=nmin,
                    mask_after_search=mask_after, **ball)
    assert np.max(np.abs(z - ref)) < 1e-9
    if nd:
        assert np.array_equal(z[:, dl], np.tile(zd, (3, 1)))                  # test/simulation/sgs.jl:18-20
    # the same normals passed in explicitly give the same field; realisation r depends on (seed, r) only
    eps = np.stack([philox.normal(42, 3 + r, N) for r in range(3)])
    assert np.max(np.abs(h.realize(0, 0, 3, noise=eps) - z)) < 1e-9      # device vs numpy Box-Muller: ulps
    assert np.array_equal(h.realize(42, 4, 1)[0], z[1])
    h.close()


@pytest.mark.parametrize("mask_after", [True, False])
@pytest.mark.parametrize("dims,k,rpath,nd", [((30, 20), 80, False, 5), ((14, 12, 6), 100, True, 0), ((40, 25), 200, True, 7)])
def test_more_than_64_neighbours(dims, k, rpath, nd, mask_after):
    """seq.jl:91-98 accepts any maxneighbors.  Beyond 64 the search -- unmasked for `mask = simulated` applied after
    the search (the front-ends' default), masked otherwise -- runs in passes of 64; the weights come from the
    one-workgroup-per-node kernel (covariance triangle in LDS up to 180 neighbours, in HBM beyond: k = 200), the sweep
    from its plain variant.  Oracle = the reference's loop; same 1e-9 as the 64-neighbour kernels."""
    from gss.engine import SGSHandle
    gvg, ovg = _vg("spherical", range=0.4 * max(dims), sill=1.2, nugget=0.1)
    cent = offt.grid_centroids(dims)
    N = cent.shape[0]
    rng = np.random.default_rng(N + k)
    dl = np.sort(rng.choice(N, nd, replace=False)) if nd else np.empty(0, dtype=np.int64)
    zd = rng.normal(size=nd)
    path = rng.permutation(N) if rpath else None
    h = SGSHandle(gvg, cent, path, dl, zd, 0.3, k, 1, mask_after_search=mask_after)
    z = h.realize(7, 0, 3)
    idx, nc, w, sg = h.weights()
    h.close()
    ref = S.realize(ovg, 0.3, cent, path, dl, zd, 7, 0, 3, maxneighbors=k, minneighbors=1, mask_after_search=mask_after)
    assert np.max(np.abs(z - ref)) < 1e-9
    assert nc.max() > 64 or k > N // 2          # the lists really are longer than one pass of the search
    if nd:
        assert np.array_equal(z[:, dl], np.tile(zd, (3, 1)))
    # one visiting order per realisation on the same kernels
    paths = np.stack([np.random.default_rng([9, r]).permutation(N) for r in range(2)])
    hp = SGSHandle(gvg, cent, paths, dl, zd, 0.3, k, 1, path_base=0, mask_after_search=mask_after)
    zp = hp.realize(7, 0, 2)
    hp.close()
    for r in range(2):
        refp = S.realize(ovg, 0.3, cent, paths[r], dl, zd, 7, r, 1, maxneighbors=k, mask_after_search=mask_after)[0]
        assert np.max(np.abs(zp[r] - refp)) < 1e-9


@pytest.mark.parametrize("distance", ["cityblock", "chebyshev"])
@pytest.mark.parametrize("mask_after", [True, False])
def test_search_distance_parameter(distance, mask_after):
    """`distance` of SeqSim's neighbour search (seq.jl:91-98; [DEP] Distances.jl Cityblock / Chebyshev): the masked and
    the unmasked indexed search rank by that metric's key; off-lattice cell centres keep the keys free of ties.
    Through the handle (12 and 70 neighbours) and through solve; a ball does not combine with it (ui.jl:25-31)."""
    import gss
    from gss import _lib
    from gss.engine import SGSHandle
    gvg, ovg = _vg("exponential", range=9.0, sill=0.8)
    rng = np.random.default_rng(17)
    cent = offt.grid_centroids((26, 19)) + rng.uniform(-0.3, 0.3, (26 * 19, 2))
    N = cent.shape[0]
    dl = np.sort(rng.choice(N, 8, replace=False))
    zd = rng.normal(size=8)
    path = rng.permutation(N)
    for k in (12, 70):
        h = SGSHandle(gvg, cent, path, dl, zd, 0.0, k, 1, mask_after_search=mask_after, distance=distance)
        z = h.realize(5, 0, 2)
        h.close()
        ref = S.realize(ovg, 0.0, cent, path, dl, zd, 5, 0, 2, maxneighbors=k, mask_after_search=mask_after,
                        distance=distance)
        assert np.max(np.abs(z - ref)) < 1e-9
        refe = S.realize(ovg, 0.0, cent, path, dl, zd, 5, 0, 2, maxneighbors=k, mask_after_search=mask_after)
        assert np.max(np.abs(ref - refe)) > 1e-6                  # the metric does change the neighbourhoods
    grid = gss.CartesianGrid(20, 15)
    prob = gss.SimulationProblem(gss.georef({"z": zd[:3]}, grid.centroids()[[7, 111, 250]]), grid, "z", 2)
    kw = dict(variogram=gvg, maxneighbors=9, distance=distance)
    sol = gss.solve(prob, gss.SGS(("z", kw), rng=4, mask="after" if mask_after else "during"))
    sole = gss.solve(prob, gss.SGS(("z", dict(kw, distance="euclidean")), rng=4, mask="after" if mask_after else "during"))
    assert np.all(np.isfinite(sol["z"][1])) and np.max(np.abs(sol["z"][1] - sole["z"][1])) > 1e-6
    with pytest.raises(_lib.GSSError, match="ball"):
        SGSHandle(gvg, cent, None, dl, zd, 0.0, 8, 1, 5.0, distance=distance)
    with pytest.raises(_lib.GSSError, match="Haversine only with"):      # no masked exhaustive search
        SGSHandle(gvg, cent, None, dl, zd, 0.0, 8, 1, distance=("haversine", 1.0))


def test_haversine_search_distance_with_the_mask_after_search():
    """`distance = Haversine(r)` (seq.jl:91-98; the reference accepts any Distances.jl metric): the key has no box bounds,
    so the search is exhaustive -- available for the reading in which the mask filters the search result (the front-ends'
    default).  (longitude, latitude) cells near 60 degrees north, where a degree of longitude is half a degree of
    latitude: the neighbourhoods differ from the Euclidean ones.  12 and 70 neighbours, handle and solve."""
    import gss
    from gss.engine import SGSHandle
    gvg, ovg = _vg("exponential", range=6.0, sill=0.8)
    rng = np.random.default_rng(23)
    cent = offt.grid_centroids((24, 18), (10.0, 55.0), (0.5, 0.5)) + rng.uniform(-0.1, 0.1, (24 * 18, 2))
    N = cent.shape[0]
    dl = np.sort(rng.choice(N, 8, replace=False))
    zd = rng.normal(size=8)
    path = rng.permutation(N)
    dist = ("haversine", 6371.0)
    for k in (12, 70):
        h = SGSHandle(gvg, cent, path, dl, zd, 0.0, k, 1, mask_after_search=True, distance=dist)
        z = h.realize(5, 0, 2)
        h.close()
        ref = S.realize(ovg, 0.0, cent, path, dl, zd, 5, 0, 2, maxneighbors=k, mask_after_search=True, distance=dist)
        assert np.max(np.abs(z - ref)) < 1e-9
        refe = S.realize(ovg, 0.0, cent, path, dl, zd, 5, 0, 2, maxneighbors=k, mask_after_search=True)
        assert np.max(np.abs(ref - refe)) > 1e-6
    grid = gss.CartesianGrid((20, 15), (10.0, 55.0), (0.5, 0.5))
    prob = gss.SimulationProblem(gss.georef({"z": zd[:3]}, grid.centroids()[[7, 111, 250]]), grid, "z", 2)
    sol = gss.solve(prob, gss.SGS(("z", dict(variogram=gvg, maxneighbors=9, distance=dist)), rng=4))
    assert np.all(np.isfinite(sol["z"][1]))
    with pytest.raises(NotImplementedError, match="mask='after'"):
        gss.solve(prob, gss.SGS(("z", dict(variogram=gvg, maxneighbors=9, distance=dist)), rng=4, mask="during"))


def test_solver_front_end_with_ninety_neighbours():
    import gss
    grid = gss.CartesianGrid((40, 40), (0.5, 0.5), (1.0, 1.0))
    pts = [(10.0, 10.0), (20.0, 30.0), (30.0, 20.0)]
    data = gss.georef(dict(z=[1.0, 0.0, 1.0]), pts)
    solver = gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=15.0), maxneighbors=90)), rng=3)
    sol = gss.solve(gss.SimulationProblem(data, grid, "z", 2), solver)
    li = lambda i, j: (j - 1) * 40 + (i - 1)
    for r in sol["z"]:
        assert np.all(np.isfinite(r)) and r[li(10, 10)] == 1.0 and r[li(20, 30)] == 0.0 and r[li(30, 20)] == 1.0


def test_weights_are_the_simple_kriging_weights():
    from gss.engine import SGSHandle
    gvg, ovg = _vg("spherical", range=10.0)
    cent = offt.grid_centroids((30, 20))
    N = cent.shape[0]
    dl = np.array([77, 300, 512])
    h = SGSHandle(gvg, cent, None, dl, np.array([1.0, -1.0, 0.5]), 0.0, 8, 1)
    idx, nc, w, sg = h.weights()
    from oracle.variogram import cov_pairwise
    sim = np.zeros(N, dtype=bool)
    sim[dl] = True
    for node in range(N):
        if node in dl:
            assert nc[node] == 0
            continue
        if node in (0, 1, 150, 599):
            cand = np.flatnonzero(sim)
            d2 = ((cent[cand] - cent[node]) ** 2)
            d2 = d2[:, 0] + d2[:, 1]
            nb = cand[np.argsort(d2, kind="stable")[:8]]
            assert nc[node] == nb.size and np.array_equal(idx[node, :nb.size], nb)
            lam = np.linalg.solve(cov_pairwise(ovg, cent[nb]), cov_pairwise(ovg, cent[nb], cent[node][None])[:, 0])
            assert np.allclose(w[node, :nb.size], lam, atol=1e-11)
            assert np.isclose(sg[node] ** 2, 1.0 - lam @ cov_pairwise(ovg, cent[nb], cent[node][None])[:, 0], atol=1e-11)
        sim[node] = True
    h.close()


def test_statistics_and_conditioning_at_scale():
    """256 x 256 cells, 128 realisations in one sweep: hard data exact, marginal moments and the lag-1
    covariance of the model within sampling error (size-independent properties)."""
    from gss.engine import SGSHandle
    gvg, ovg = _vg("exponential", range=15.0, sill=2.0)
    cent = offt.grid_centroids((256, 256))
    N = cent.shape[0]
    rng = np.random.default_rng(5)
    h = SGSHandle(gvg, cent, rng.permutation(N), None, None, 1.0, 16, 1)
    z = h.realize(9, 0, 128)
    h.close()
    assert abs(z.mean() - 1.0) < 0.05 and abs(z.var() - 2.0) < 0.1
    g = z.reshape(128, 256, 256) - 1.0
    for lag in (1, 4):
        emp = np.mean(g[:, :, lag:] * g[:, :, :-lag])
        assert abs(emp - float(cov_h(ovg, np.array(float(lag))))) < 0.08
    dl = np.sort(rng.choice(N, 200, replace=False))
    zd = rng.normal(size=200)
    hc = SGSHandle(gvg, cent, None, dl, zd, 0.0, 12, 1, 40.0)
    zc = hc.realize(1, 0, 8)
    hc.close()
    assert np.array_equal(zc[:, dl], np.tile(zd, (8, 1))) and np.all(np.isfinite(zc))


def test_solver_through_solve_and_errors():
    import gss
    from gss import _lib
    grid = gss.CartesianGrid((100, 100), (0.5, 0.5), (1.0, 1.0))          # test/simulation/sgs.jl:2-20
    data = gss.georef(dict(z=[1.0, 0.0, 1.0]), [(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
    solver = gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=35.0), neighborhood=gss.MetricBall(30.0))),
                     rng=2017)
    sol1 = gss.solve(gss.SimulationProblem(data, grid, "z", 3), solver)
    sol2 = gss.solve(gss.SimulationProblem(grid, {"z": float}, 3), solver)
    reals = sol1["z"]
    li = lambda i, j: (j - 1) * 100 + (i - 1)
    assert all(r[li(25, 25)] == 1.0 and r[li(50, 75)] == 0.0 and r[li(75, 50)] == 1.0 for r in reals)
    assert len(sol2["z"]) == 3 and np.all(np.isfinite(sol2["z"][2]))
    cent = grid.centroids()
    with pytest.raises(_lib.GSSError, match="at most 1024"):
        from gss.engine import SGSHandle
        SGSHandle(gss.SphericalVariogram(range=35.0), cent, None, None, None, 0.0, 1025)
    with pytest.raises(_lib.GSSError, match="not a permutation"):
        SGSHandle(gss.SphericalVariogram(range=35.0), cent[:10], np.zeros(10, dtype=np.int64), None, None, 0.0, 3)


def test_one_visiting_order_per_realisation_matches_oracle(monkeypatch):
    """RandomPath semantics (seq.jl:99-102: `traverse` runs inside solvesingle, so every realisation walks its own
    permutation): gss_sgs_create_paths runs stage A once per path and the sweep gives each realisation a lane of its
    own.  Oracle = the reference's loop with that realisation's path.  Through the handle, through `solve` with
    path=("random", seed) and across handle boundaries (blocks of realisations)."""
    import gss
    from gss.engine import SGSHandle
    from gss import solvers
    gvg, ovg = _vg("spherical", range=7.0, sill=1.3, nugget=0.05)
    cent = offt.grid_centroids((18, 13))
    N = cent.shape[0]
    rng = np.random.default_rng(3)
    dl = np.sort(rng.choice(N, 9, replace=False))
    zd = rng.normal(size=9)
    paths = np.stack([np.random.default_rng([5, r]).permutation(N) for r in range(4, 9)])
    h = SGSHandle(gvg, cent, paths, dl, zd, 0.4, 7, 1, path_base=4)
    z = h.realize(21, 5, 3)                                   # realisations 5, 6, 7 <-> rows 1, 2, 3
    with pytest.raises(Exception, match="no visiting order"):
        h.realize(21, 8, 2)
    h.close()
    for i, r in enumerate((5, 6, 7)):
        ref = S.realize(ovg, 0.4, cent, paths[r - 4], dl, zd, 21, r, 1, maxneighbors=7)[0]
        assert np.max(np.abs(z[i] - ref)) < 1e-9 and np.array_equal(z[i][dl], zd)
    # solver front-end, blocks of two realisations per handle
    monkeypatch.setattr(solvers, "SGS_PATHS_PER_HANDLE", 2)
    grid = gss.CartesianGrid(18, 13)
    data = gss.georef({"z": zd[:3]}, cent[dl[:3]])
    sol = gss.solve(gss.SimulationProblem(data, grid, "z", 5),
                    gss.SGS(("z", dict(variogram=gvg, mean=0.4, path=("random", 5), maxneighbors=7)), rng=21))
    loop = gss.simulate_with_generic_loop(gss.SimulationProblem(data, grid, "z", 5),
                                          gss.SGS(("z", dict(variogram=gvg, mean=0.4, path=("random", 5), maxneighbors=7)),
                                                  rng=21))
    for r in range(5):
        ref = S.realize(ovg, 0.4, cent, np.random.default_rng([5, r]).permutation(N), dl[:3], zd[:3], 21, r, 1,
                        maxneighbors=7, mask_after_search=True)[0]     # front-end default: mask = "after"
        assert np.max(np.abs(sol["z"][r] - ref)) < 1e-9 and np.array_equal(sol["z"][r], loop["z"][r])
    assert np.max(np.abs(sol["z"][0] - sol["z"][1])) > 1e-3


def test_level_schedule_is_bit_identical_to_the_walk_along_the_path():
    """Stage B runs level by level over the dependency graph of the shared visiting order (sgs.hip,
    sgs_level_sweep_kernel): the same sums in the same order as one wave walking the path (GSS_SGS_LEVELS=0, in a
    child process), for 16 and for 80 neighbours, with conditioning data, an odd number of realisations, both
    readings of the mask, and with one visiting order per realisation (every order's levels in one schedule)."""
    import os
    import subprocess
    import sys
    import tempfile
    code = ("import sys, numpy as np, gss\n"
            "from gss.engine import SGSHandle\n"
            "out = sys.argv[1]\n"
            "res = []\n"
            "for dims, k, after in (((96, 80), 16, False), ((64, 50), 80, True), ((20, 18, 10), 12, True)):\n"
            "    N = int(np.prod(dims)); rng = np.random.default_rng(N)\n"
            "    g = np.meshgrid(*[np.arange(d) + 0.5 for d in dims], indexing='ij')\n"
            "    cent = np.stack([a.ravel(order='F') for a in g], 1)\n"
            "    dl = np.sort(rng.choice(N, 25, replace=False)); zd = rng.normal(size=25)\n"
            "    h = SGSHandle(gss.SphericalVariogram(range=12.0, nugget=0.05), cent, rng.permutation(N), dl, zd, 0.3, k, 1,\n"
            "                  mask_after_search=after)\n"
            "    res.append(h.realize(7, 2, 67)); h.close()\n"
            "paths = np.stack([rng.permutation(N) for _ in range(5)])     # one visiting order per realisation\n"
            "h = SGSHandle(gss.SphericalVariogram(range=12.0, nugget=0.05), cent, paths, dl, zd, 0.3, 10, 1)\n"
            "res.append(h.realize(7, 0, 5)); res.append(h.realize(7, 1, 3)); h.close()\n"
            "np.savez(out, *res)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for sw in ("1", "0"):
            out = os.path.join(d, f"o{sw}.npz")
            env = dict(os.environ, GSS_SGS_LEVELS=sw, PYTHONPATH=os.pathsep.join(
                [os.path.join(root, "geostatssolvers.jl_amd"), os.environ.get("PYTHONPATH", "")]))
            subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
            with np.load(out) as f:
                outs.append([f[k] for k in f.files])
    assert len(outs[0]) == 5
    for a, b in zip(*outs):
        assert a.shape == b.shape and np.array_equal(a, b)


def test_short_levels_in_one_launch_are_bit_identical_to_both_other_sweeps():
    """The row-by-row order (LinearPath) has thousands of short levels; `sgs_level_team_kernel` sweeps them in ONE launch
    with the field in a layout of its own (a workgroup per group of realisations, workgroup barriers between levels,
    lists and normals staged by LDS-direct loads).  Forced on (GSS_SGS_TEAM=1) it must give, bit for bit, the fields of
    the launch-per-level sweep (GSS_SGS_TEAM=0) and of the wave that walks the path (GSS_SGS_LEVELS=0): 2-D and 3-D
    grids, 4 ... 20 neighbours (10, the reference's default, and 7: lists that are not a multiple of four), conditioning data, realisation counts that do not fill the last group, a random order
    (long levels: several chunks and rounds per level), supplied normals."""
    import os
    import subprocess
    import sys
    import tempfile
    code = ("import sys, numpy as np, gss\n"
            "from gss.engine import SGSHandle\n"
            "out = sys.argv[1]\n"
            "res = []\n"
            "for dims, k, R, order in (((96, 80), 16, 67, None), ((70, 50), 12, 8, None), ((24, 18, 10), 20, 130, None),\n"
            "                          ((64, 64), 4, 9, None), ((90, 70), 16, 33, 'random'), ((40, 30), 8, 3, 'noise'),\n"
            "                          ((60, 45), 10, 5, None), ((50, 40), 7, 3, None)):\n"
            "    N = int(np.prod(dims)); rng = np.random.default_rng(N)\n"
            "    g = np.meshgrid(*[np.arange(d) + 0.5 for d in dims], indexing='ij')\n"
            "    cent = np.stack([a.ravel(order='F') for a in g], 1)\n"
            "    dl = np.sort(rng.choice(N, 25, replace=False)); zd = rng.normal(size=25)\n"
            "    path = rng.permutation(N) if order == 'random' else None\n"
            "    h = SGSHandle(gss.SphericalVariogram(range=12.0, nugget=0.05), cent, path, dl, zd, 0.3, k, 1)\n"
            "    noise = rng.normal(size=(R, N)) if order == 'noise' else None\n"
            "    res.append(h.realize(7, 2, R, noise=noise)); h.close()\n"
            "np.savez(out, *res)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for tag, extra in (("team", dict(GSS_SGS_TEAM="1")), ("launches", dict(GSS_SGS_TEAM="0")),
                           ("walk", dict(GSS_SGS_LEVELS="0"))):
            out = os.path.join(d, f"{tag}.npz")
            env = dict(os.environ, PYTHONPATH=os.pathsep.join(
                [os.path.join(root, "geostatssolvers.jl_amd"), os.environ.get("PYTHONPATH", "")]), **extra)
            subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=600)
            with np.load(out) as f:
                outs.append([f[k] for k in f.files])
    assert len(outs[0]) == 8
    for a, b, c in zip(*outs):
        assert a.shape == b.shape == c.shape and np.isfinite(a).all()
        assert np.array_equal(a, b) and np.array_equal(a, c)
    # the team kernel's own fields against the oracle (seq.jl:102-135), directly and not through the other sweeps:
    # the reference's default of 10 neighbours and a list length that is not a multiple of four, row-by-row order
    for case, (dims, k, R) in ((6, ((60, 45), 10, 5)), (7, ((50, 40), 7, 3))):
        N = int(np.prod(dims))
        rng = np.random.default_rng(N)
        g = np.meshgrid(*[np.arange(d) + 0.5 for d in dims], indexing="ij")
        cent = np.stack([a.ravel(order="F") for a in g], 1)
        dl = np.sort(rng.choice(N, 25, replace=False))
        zd = rng.normal(size=25)
        ref = S.realize(Variogram("spherical", range=12.0, nugget=0.05), 0.3, cent, None, dl, zd, 7, 2, R, maxneighbors=k)
        assert np.max(np.abs(outs[0][case] - ref)) < 1e-9
        assert np.array_equal(outs[0][case][:, dl], np.broadcast_to(zd, (R, 25)))
