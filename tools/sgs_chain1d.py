import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch, gss
from gss import _lib
from gss.engine import SGSHandle
for N in (4096, 65536):
    cent = (np.arange(N) + 0.5).reshape(-1, 1)
    dl = np.array([0]); zd = np.array([0.3])
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h = SGSHandle(gss.SphericalVariogram(range=20.0), cent, None, dl, zd, 0.0, 4, 1, 25.0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("1-D chain N=%d create %.1f ms" % (N, 1e3 * (t1 - t0)), flush=True)
        h.close()
