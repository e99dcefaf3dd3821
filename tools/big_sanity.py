"""Sizes far above the BASELINE configs, for the record: FFTGS on 1024^3 cells (8.6 GB per realisation), on bricks with
unequal power-of-two edges, and global kriging with 16 384 data.  python3 tools/big_sanity.py (GPU box)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import gss
from gss import _lib
from gss.engine import FFTGSHandle, KrigHandle
# FFTGS at 1024^3 (8.6 GB per realisation) and a brick 1024 x 512 x 256
for dims in ((1024, 1024, 1024), (1024, 512, 256), (32, 64, 1024)):
    N = int(np.prod(dims))
    t0 = time.perf_counter()
    f = FFTGSHandle(gss.ExponentialVariogram(range=40.0), dims)
    out = torch.empty((1, N), dtype=torch.float64, device="cuda")
    f.realize(4, 0, 1, out=out); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for r in range(3): f.realize(4, r, 1, out=out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    var = float((out[0] * out[0]).sum() / (N - 1)); mean = float(out[0].mean())
    print("fftgs", dims, "create+first %.2f s, %.1f ms per realisation, variance %.12f mean %.2e" % (t1 - t0, (t2 - t1) / 3 * 1e3, var, mean), flush=True)
    f.close(); del out
    print("pool bytes", _lib.stat("pool_bytes"), flush=True)
# global kriging with 16 384 data, 2e5 points
rng = np.random.default_rng(3)
x = rng.uniform(0, 100, (16384, 3)); z = rng.normal(size=16384); x0 = rng.uniform(0, 100, (200000, 3))
t0 = time.perf_counter()
h = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5, nugget=0.01), 1, x, z)
mu, var, st = h.predict_global(x0)
t1 = time.perf_counter()
print("krig n=16384 m=2e5: %.2f s, status any %s, mu range %.3f..%.3f var range %.3e..%.3e" % (t1 - t0, bool(st.any()), mu.min(), mu.max(), var.min(), var.max()), flush=True)
# exactness at the data: predict at the first 1000 data points
mu2, var2, _ = h.predict_global(x[:1000])
print("at the first 1000 data: max|mu - z| %.2e, max var %.2e (exact interpolation, nugget or not)" % (np.abs(mu2 - z[:1000]).max(), var2.max()))
h.close()
