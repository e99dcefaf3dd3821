"""Cold-start times of a process: library load, first fit, first prediction, first FFTGS / LUGS / SGS / IDW call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
t0 = time.perf_counter()
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t1 = time.perf_counter(); print("torch + device context %.0f ms" % (1e3 * (t1 - t0)))
import gss
from gss import _lib
from gss.engine import KrigHandle, HipEngine, FFTGSHandle, LUGSHandle, SGSHandle, OK
_lib.lib()
t2 = time.perf_counter(); print("import gss + dlopen %.0f ms" % (1e3 * (t2 - t1)))


def timed(name, f):
    torch.cuda.synchronize(); a = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print("%-34s %8.1f ms" % (name, 1e3 * (time.perf_counter() - a)), flush=True)
    return r


rng = np.random.default_rng(0)
x = rng.uniform(0, 100, (1000, 3)); z = rng.normal(size=1000); x0 = rng.uniform(0, 100, (100000, 3))
vg = gss.MaternVariogram(range=30.0, order=1.5)
h = timed("first kriging fit", lambda: KrigHandle(vg, OK, x, z))
timed("first global prediction", lambda: h.predict_global(x0))
timed("second global prediction", lambda: h.predict_global(x0))
timed("first moving-neighbourhood call", lambda: h.predict_knn(x0, 32))
timed("second moving-neighbourhood call", lambda: h.predict_knn(x0, 32))
timed("first IDW call", lambda: HipEngine.idw(x, z, x0, 16))
f = timed("first FFTGS create (128^3)", lambda: FFTGSHandle(gss.ExponentialVariogram(range=10.0), (128, 128, 128)))
timed("first FFTGS realisation", lambda: f.realize(1, 0, 1, device=True))
g = np.meshgrid(np.arange(64) + 0.5, np.arange(64) + 0.5, indexing="ij"); cent = np.stack([a.ravel(order="F") for a in g], 1)
dl = np.arange(0, 4096, 16); zd = rng.normal(size=dl.size)
l = timed("first LUGS create (64 x 64)", lambda: LUGSHandle(gss.SphericalVariogram(range=10.0), cent, dl, zd))
timed("first LUGS realisations", lambda: l.realize(1, 0, 10))
s = timed("first SGS create (64 x 64)", lambda: SGSHandle(gss.SphericalVariogram(range=10.0), cent, None, dl, zd, 0.0, 16, 1, 20.0))
timed("first SGS realisations", lambda: s.realize(1, 0, 10))
