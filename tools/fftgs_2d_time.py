"""FFTGS on grids outside the power-of-two 3-D pipeline: ms per realisation on the library's own generic passes (sizes
2^a 3^b 5^c 7^d: fftgs_generic.h) or on rocFFT (GSS_FFTGS_PATH=rocfft, and every other size).
python3 tools/fftgs_2d_time.py; GSS_FFTGS_PATH=rocfft python3 tools/fftgs_2d_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch, gss
from gss.engine import FFTGSHandle
for dims in ((100, 100), (1000, 1000), (1024, 1024), (4096, 1024), (2048, 2048), (4096, 4096), (200, 200, 200),
             (300, 300, 100), (300, 300, 300), (500, 500, 500), (256, 256, 256), (512, 512, 512)):
    vg = gss.ExponentialVariogram(range=dims[0] / 10.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = FFTGSHandle(vg, dims)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    R = 64 if np.prod(dims) <= 4e6 else 16
    z = h.realize(1, 0, R, device=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    z = None   # (the second call takes the first one's block back from torch's cache instead of a fresh hipMalloc)
    z = h.realize(1, 0, R, device=True)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    N = int(np.prod(dims))
    print("%-14s create %7.1f ms  %6.3f ms per realisation  (%.1f GB/s on 32 N bytes)  std %.3f"
          % ("x".join(map(str, dims)), 1e3 * (t1 - t0), 1e3 * (t3 - t2) / R, 32 * N * R / (t3 - t2) / 1e9, float(z.std())), flush=True)
    h.close()
