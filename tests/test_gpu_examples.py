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
