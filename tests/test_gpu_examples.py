"""examples/quickstart.py stays runnable: every solver of the twin on the reference's own test inputs."""
import os
import runpy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_quickstart_runs_and_its_results_are_sane(capsys):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ns = runpy.run_path(os.path.join(root, "examples", "quickstart.py"))
    out = ns["out"]
    assert set(out) >= {"kriging_global", "kriging_nearest", "kriging_local", "kriging_uk", "idw", "lwr", "fftgs",
                        "fftgs_cond", "lugs", "lugs_corr", "sgs"}
    for k in ("kriging_global", "kriging_nearest", "kriging_local", "kriging_uk"):
        mu, var = out[k]
        assert mu.shape == (100,) and np.isfinite(mu).all() and (var > -1e-9).all()
    assert np.array_equal(out["kriging_nearest"][0], out["kriging_local"][0])      # the ball holds every sample
    assert out["fftgs"].shape == (3, 10000) and abs(out["fftgs"].var() - 1.0) < 0.05
    assert out["fftgs_cond"].shape == (100, 10000) and out["lugs"].shape == (2, 100) and out["sgs"].shape == (2, 10000)
    assert np.isfinite(out["idw"]).all() and np.isfinite(out["lwr"]).all()
    assert "SGS: 2 realisations, value at the cell of the datum z = 1: [1. 1.]" in capsys.readouterr().out


@pytest.mark.parametrize("view", [False, True])
def test_domain_points_formed_on_the_device_change_nothing(view, monkeypatch):
    """`solve` on (a view of) a Cartesian grid forms the domain points in HBM from 200 000 cells on and brings the table
    back through page-locked memory: bit-identical to the host path for kriging (global, moving neighbourhood, two
    variables sharing the system), IDW and LWR."""
    import gss
    rng = np.random.default_rng(8)
    grid = gss.CartesianGrid((90, 80, 30), (1.0, -2.0, 0.5), (0.5, 1.5, 2.0))          # 216 000 cells
    dom = gss.view(grid, np.sort(rng.choice(grid.nelements(), 205_000, replace=False))) if view else grid
    lo, hi = np.array([1.0, -2.0, 0.5]), np.array([1.0 + 45.0, -2.0 + 120.0, 0.5 + 60.0])
    xyz = rng.uniform(lo, hi, (300, 3))
    data = gss.georef({"a": rng.normal(size=300), "b": rng.normal(size=300)}, xyz)
    vg = gss.SphericalVariogram(range=25.0, nugget=0.1)
    solvers = [gss.KrigingSolver(("a", dict(variogram=vg)), ("b", dict(variogram=vg))),
               gss.KrigingSolver(("a", dict(variogram=vg, maxneighbors=12, degree=1))),
               gss.IDWSolver(("a", dict(maxneighbors=8)), ("b", dict(maxneighbors=8))),
               gss.LWRSolver(("a", dict(maxneighbors=10)))]
    for solver in solvers:
        prob = gss.EstimationProblem(data, dom, tuple(solver.vparams))
        monkeypatch.setenv("GSS_SOLVE_DEVICE_POINTS", "200000")
        dev = gss.solve(prob, solver)
        monkeypatch.setenv("GSS_SOLVE_DEVICE_POINTS", "0")
        host = gss.solve(prob, solver)
        assert set(dev.names()) == set(host.names())
        for name in dev.names():
            assert np.array_equal(dev[name], host[name], equal_nan=True), name
