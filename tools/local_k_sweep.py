"""Moving-neighbourhood kriging over neighbour counts: search (K4) and systems (K5) per 1.25e6 points, UK degree 1,
5 000 3-D data, Matern-3/2 (the configs[4] data).  python3 tools/local_k_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
import gss
from gss import _lib
from gss.engine import KrigHandle
rng = np.random.default_rng(6)
x = rng.uniform(0, 100, (5000, 3)); z = x @ np.array([0.01, -0.02, 0.005]) + rng.normal(size=5000)
m = 1_250_000
x0 = torch.as_tensor(np.random.default_rng(7).uniform(0, 100, (m, 3)), device="cuda")
for k in (4, 8, 16, 32, 48, 64):
    h = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5), 2, x, z, degree=1, factor=False)
    h.predict_knn(x0[:1000], k)
    torch.cuda.synchronize()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    h.predict_knn(x0, k, device=True) if "device" in KrigHandle.predict_knn.__code__.co_varnames else h.predict_knn(x0, k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    knn = _lib.profile_read("knn"); kl = _lib.profile_read("krig_local")
    print("k=%2d  search %.2f ms  systems %.2f ms  call %.1f ms" % (k, knn[0], kl[0], 1e3 * dt), flush=True)
    h.close()
