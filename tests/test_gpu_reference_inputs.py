"""The reference's own test INPUTS, unmodified, through `gss.solve` on the device and through the oracle.

/root/reference/test/estimation/krig.jl:6-19 (1-D problem, global / nearest / local), test/simulation/fft.jl:3-12 (isotropic
and anisotropic Gaussian models on 100 x 100) and :24-32 (conditional, 100 realisations).  All of them use
`GaussianVariogram` with the nugget at 0; both sides evaluate it with `nugget + 1e-6` (gss/variograms.py).
The LUGS inputs (test/simulation/lu.jl) are in tests/test_gpu_lugs.py.  The reference asserts nothing numerical on
these inputs (shapes only), so the check is device == oracle at the Gaussian-model tolerance 1e-6."""
import numpy as np
import pytest

from oracle import fftgs as OF, kriging as K, philox
from oracle.variogram import Variogram

pytestmark = pytest.mark.gpu


def test_kriging_1d_problem_all_three_solvers():           # test/estimation/krig.jl:6-19
    import gss
    x = np.arange(0.0, 101.0, 10.0)[:, None]
    z = np.array([0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.4, 0.3, 0.2, 0.1, 0.0])
    problem = gss.EstimationProblem(gss.georef({"z": z}, x), gss.CartesianGrid(100), "z")
    vg = gss.GaussianVariogram(range=35.0, nugget=0.0)
    ovg = Variogram("gaussian", range=35.0, nugget=0.0)
    g = OF.grid_centroids((100,))
    sols = [gss.solve(problem, gss.KrigingSolver(("z", p))) for p in
            (dict(variogram=vg), dict(variogram=vg, maxneighbors=3),
             dict(variogram=vg, maxneighbors=3, neighborhood=gss.MetricBall(100.0)))]
    rg = K.exactsolve(K.OK, ovg, x, z, g)
    rn = K.approxsolve(K.OK, ovg, x, z, g, 3)
    rl = K.approxsolve(K.OK, ovg, x, z, g, 3, radius=100.0)
    for sol, ref in zip(sols, (rg, rn, rl)):
        assert np.max(np.abs(sol["z"] - ref[0])) < 1e-6 and np.max(np.abs(sol["z_variance"] - ref[1])) < 1e-6


@pytest.mark.parametrize("ball", [None, (20.0, 5.0)])
def test_fftgs_gaussian_100x100_as_written(ball):            # test/simulation/fft.jl:3-12
    import gss
    gvg = gss.GaussianVariogram(range=10.0) if ball is None else gss.GaussianVariogram(gss.MetricBall(ball))
    ovg = Variogram("gaussian", range=10.0) if ball is None else Variogram("gaussian", radii=ball)
    sol = gss.solve(gss.SimulationProblem(gss.CartesianGrid(100, 100), ("z", float), 3),
                    gss.FFTGS(("z", dict(variogram=gvg)), rng=2019))
    pre = OF.preprocess(ovg, (100, 100))
    ref = OF.realize(pre, 2019, 0, 3)
    assert np.max(np.abs(np.stack(sol["z"]) - ref)) < 1e-6


def test_fftgs_conditional_100_realisations_as_written():    # test/simulation/fft.jl:24-32
    import gss
    coords = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)])
    vals = [1.0, -1.0, 1.0]
    problem = gss.SimulationProblem(gss.georef({"z": vals}, coords), gss.CartesianGrid(100, 100), ("z", float), 100)
    sol = gss.solve(problem, gss.FFTGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0))), rng=2022))
    assert len(sol["z"]) == 100 and sol[0].z.shape == (10000,)
    pre = OF.preprocess(Variogram("gaussian", range=10.0), (100, 100), data_coords=coords, data_vals=vals)
    for r in (0, 57, 99):
        ref = OF.solvesingle(pre, philox.uniform(2022, r, 10000))
        assert np.max(np.abs(sol[r].z - ref)) < 1e-6
