"""Timing sweep of the K4 neighbour search (profile scope "knn") over k and n; prints one line per case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
import gss
from gss import _lib
from gss.engine import HipEngine
m = 1_250_000
x0 = torch.as_tensor(np.random.default_rng(17).uniform(0, 100, (m, 3)), device="cuda")
for n in (5000, 50000, 500000):
    x = torch.as_tensor(np.random.default_rng(16).uniform(0, 100, (n, 3)), device="cuda")
    z = torch.zeros(n, dtype=torch.float64, device="cuda")
    for k in (1, 4, 16, 64):
        HipEngine.idw(x, z, x0[:10000], k)
        _lib.profile_reset(); _lib.profile_enable(True)
        HipEngine.idw(x, z, x0, k)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        print(f"n={n} k={k} knn_ms={_lib.profile_read('knn')[0]:.2f} est_ms={_lib.profile_read('idw')[0]:.2f}", flush=True)
