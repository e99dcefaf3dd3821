"""Host-array calls against device-array calls (configs[1], configs[4], IDW shapes):
python3 tools/host_path_times.py   (GSS_HOST_PIPELINE=0: one copy in, one copy out, as before round 2)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
import gss
from gss.engine import KrigHandle, HipEngine, OK, UK


def best(f, reps=4):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts)


rng = np.random.default_rng(5)
vg = gss.MaternVariogram(range=30.0, order=1.5)
x = rng.uniform(0, 100, (1000, 3)); z = rng.normal(size=1000)
x0 = rng.uniform(0, 100, (1_000_000, 3)); x0d = torch.as_tensor(x0, device="cuda")
h = KrigHandle(vg, OK, x, z)
print("global kriging 1e6 points: device %.2f ms, host %.2f ms" % (best(lambda: h.predict_global(x0d)), best(lambda: h.predict_global(x0))))
x = rng.uniform(0, 100, (5000, 3)); z = rng.normal(size=5000)
x0 = rng.uniform(0, 100, (1_250_000, 3)); x0d = torch.as_tensor(x0, device="cuda")
h = KrigHandle(vg, UK, x, z, degree=1, factor=False)
print("moving-neighbourhood kriging k=64, 1.25e6 points: device %.2f ms, host %.2f ms" % (best(lambda: h.predict_knn(x0d, 64)), best(lambda: h.predict_knn(x0, 64))))
x = rng.uniform(0, 100, (50000, 3)); z = rng.normal(size=50000); xd = torch.as_tensor(x, device="cuda"); zd = torch.as_tensor(z, device="cuda")
print("IDW k=16, 1.25e6 points: device %.2f ms, host %.2f ms" % (best(lambda: HipEngine.idw(xd, zd, x0d, 16)), best(lambda: HipEngine.idw(x, z, x0, 16))))
