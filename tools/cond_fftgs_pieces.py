"""Pieces of the conditional FFTGS preprocess / solve at the section-8f.1 shape (128^3 cells, 1 000 data, 16 realisations)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch, gss
from gss.engine import HipEngine, KrigHandle, SK
from gss.solvers import _centroids_device


def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best, r


e, R = 128, 16
rng = np.random.default_rng(9)
grid = gss.CartesianGrid((e, e, e))
xd = rng.uniform(0.0, float(e), (1000, 3)); zd = rng.normal(size=1000)
vg = gss.ExponentialVariogram(range=20.0)
cdev = _centroids_device(grid)
ms, kh = t(lambda: KrigHandle(vg, SK, xd, zd, mean=0.0)); print("krig create %.2f ms" % ms)
zb1 = torch.as_tensor(zd[None, :], device="cuda")
ms, _ = t(lambda: kh.predict_global_batch(cdev, zb1)); print("means, 1 vector %.2f ms" % ms)
zb = torch.as_tensor(rng.normal(size=(R, 1000)), device="cuda")
ms, _ = t(lambda: kh.predict_global_batch(cdev, zb)); print("means, %d vectors %.2f ms" % (R, ms))
xdd = torch.as_tensor(xd, device="cuda")
ms, _ = t(lambda: HipEngine.knn_search(cdev, xdd, 1)); print("nearest cell of the data (search over the cells) %.2f ms" % ms)
h = HipEngine.FFTGS(vg, grid.dims, grid.spacing, 0.0)
ms, z = t(lambda: h.realize(3, 0, R, device=True)); print("%d unconditional realisations %.2f ms" % (R, ms))
ms, _ = t(lambda: z.cpu()); print("results to the host %.2f ms" % ms)
