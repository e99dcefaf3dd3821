"""SGS with one random visiting order per realisation (seq.jl:99-102): create (stage A per order) and realize times.
python3 tools/sgs_paths_time.py [edge] [paths]   (GSS_SGS_LEVELS=0: one lane walks each order)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
import gss
from gss.engine import SGSHandle
e = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = np.meshgrid(np.arange(e) + 0.5, np.arange(e) + 0.5, indexing="ij")
cent = np.stack([a.ravel(order="F") for a in g], 1)
N = e * e
rng = np.random.default_rng(1)
dl = np.sort(rng.choice(N, 100, replace=False)); zd = rng.normal(size=100)
paths = np.stack([rng.permutation(N) for _ in range(P)])
vg = gss.SphericalVariogram(range=35.0)
SGSHandle(vg, cent[:5000], None, None, None, 0.0, 16, 1, 30.0).close()
torch.cuda.synchronize()
t0 = time.perf_counter()
h = SGSHandle(vg, cent, paths, dl, zd, 0.0, 16, 1, 30.0)
torch.cuda.synchronize()
t1 = time.perf_counter()
z = h.realize(3, 0, P, device=True)
torch.cuda.synchronize()
t2 = time.perf_counter()
h.realize(3, 0, P, out=z)
torch.cuda.synchronize()
t3 = time.perf_counter()
print("%d x %d cells, %d visiting orders: create %.1f ms, realize %.1f ms (first %.1f), checksum %.6f"
      % (e, e, P, 1e3 * (t1 - t0), 1e3 * (t3 - t2), 1e3 * (t2 - t1), float(z.sum())))
