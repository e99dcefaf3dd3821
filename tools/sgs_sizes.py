"""SGS create / realize times over grid sizes and visiting orders (level schedule): python3 tools/sgs_sizes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
import gss
from gss.engine import SGSHandle
vg = gss.SphericalVariogram(range=20.0)
for dims, k, R, order in (((256, 256), 16, 64, "linear"), ((256, 256), 16, 64, "random"), ((1024, 1024), 12, 64, "linear"),
                          ((1024, 1024), 12, 64, "random"), ((2048, 2048), 8, 64, "random"), ((96, 96, 96), 16, 128, "random"),
                          ((96, 96, 96), 16, 128, "linear")):
    N = int(np.prod(dims))
    g = np.meshgrid(*[np.arange(d) + 0.5 for d in dims], indexing="ij")
    cent = np.stack([a.ravel(order="F") for a in g], 1)
    rng = np.random.default_rng(N)
    dl = np.sort(rng.choice(N, 50, replace=False)); zd = rng.normal(size=50)
    path = None if order == "linear" else rng.permutation(N)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = SGSHandle(vg, cent, path, dl, zd, 0.0, k, 1, 25.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    z = h.realize(1, 0, R, device=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    h.realize(1, 0, R, out=z)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("%-16s k=%2d R=%3d %-6s create %8.1f ms  realize %8.1f ms  (%.2f G cells/s)  std %.3f"
          % ("x".join(map(str, dims)), k, R, order, 1e3 * (t1 - t0), 1e3 * (t3 - t2), N * R / (t3 - t2) / 1e9, float(z.std())), flush=True)
    h.close(); del z
