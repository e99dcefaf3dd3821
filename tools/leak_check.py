"""Repeated solves of every kind: free device memory must settle (released blocks are cached, not leaked)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch, gss
from gss.engine import KrigHandle, HipEngine, FFTGSHandle, LUGSHandle, SGSHandle, OK, UK
rng = np.random.default_rng(0)
g = np.meshgrid(np.arange(64) + 0.5, np.arange(64) + 0.5, indexing="ij"); cent = np.stack([a.ravel(order="F") for a in g], 1)
dl = np.arange(0, 4096, 16); zd = rng.normal(size=dl.size)
vg = gss.MaternVariogram(range=30.0, order=1.5)
free = []
for it in range(60):
    n = 500 + 37 * (it % 7)
    x = rng.uniform(0, 100, (n, 3)); z = rng.normal(size=n); x0 = rng.uniform(0, 100, (200_000 + 1000 * (it % 5), 3))
    h = KrigHandle(vg, OK, x, z, async_fit=bool(it % 2)); h.predict_global(x0); h.predict_knn(x0[:50000], 24); h.close()
    HipEngine.idw(x, z, x0, 12)
    f = FFTGSHandle(gss.ExponentialVariogram(range=10.0), (64, 64, 64) if it % 2 else (96, 80)); f.realize(it, 0, 2); f.close()
    l = LUGSHandle(gss.SphericalVariogram(range=10.0), cent, dl, zd); l.realize(it, 0, 5); l.close()
    s = SGSHandle(gss.SphericalVariogram(range=10.0), cent, rng.permutation(4096) if it % 2 else None, dl, zd, 0.0, 12, 1, 20.0)
    s.realize(it, 0, 8); s.close()
    torch.cuda.synchronize()
    free.append(torch.cuda.mem_get_info()[0] / 2**20)
    if it % 10 == 9:
        print("iteration %3d: free %.0f MiB" % (it + 1, free[-1]), flush=True)
drift = free[19] - free[-1]
print("free memory after 20 iterations %.0f MiB, after 60 %.0f MiB (drift %.0f MiB)" % (free[19], free[-1], drift))
assert drift < 64, "device memory keeps shrinking"
