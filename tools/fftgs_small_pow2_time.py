"""Small power-of-two 3-D grids: the fused pipeline (one realisation per launch) against the generic passes with batched
realisations (GSS_FFTGS_PATH=generic).  python3 tools/fftgs_small_pow2_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch, gss
from gss.engine import FFTGSHandle
for dims in ((32, 32, 32), (64, 64, 64), (128, 64, 64), (128, 128, 64), (128, 128, 128), (256, 128, 128), (256, 256, 128), (256, 256, 256)):
    N = int(np.prod(dims))
    R = 128 if N <= 2 ** 21 else 32
    res = {}
    for path in ("fused", "generic"):
        os.environ["GSS_FFTGS_PATH"] = path
        h = FFTGSHandle(gss.ExponentialVariogram(range=dims[0] / 10.0), dims)
        out = torch.empty((R, N), dtype=torch.float64, device="cuda")
        h.realize(1, 0, R, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.realize(1, 0, R, out=out)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        res[path] = ((t1 - t0) / R * 1e3, out[:2].clone())
        h.close()
    os.environ.pop("GSS_FFTGS_PATH", None)
    print("%-14s fused %.4f ms  generic+batch %.4f ms per realisation   max diff %.2e" %
          ("x".join(map(str, dims)), res["fused"][0], res["generic"][0], float((res["fused"][1] - res["generic"][1]).abs().max())), flush=True)
