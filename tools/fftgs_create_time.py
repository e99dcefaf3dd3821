import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/geostatssolvers.jl_amd"]
import numpy as np, torch, gss
from gss.engine import FFTGSHandle
for dims in ((1024, 1024), (1024, 1024), (2048, 1024), (100, 100), (100, 100), (128, 100), (64, 64, 50)):
    vg = gss.ExponentialVariogram(range=dims[0] / 10.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = FFTGSHandle(vg, dims)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    z = h.realize(1, 0, 4, device=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(dims, "create %.1f ms, first realize(4) %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
    h.close()
