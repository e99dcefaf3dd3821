"""The rocFFT pipeline (grids with odd or non-smooth sizes) one realisation per execution against batched plans.
python3 tools/fftgs_rocfft_batch_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'geostatssolvers.jl_amd')]
import numpy as np, torch, gss
from gss.engine import FFTGSHandle
for dims in ((101, 101), (201, 201), (501, 501), (1001, 1001), (51, 51, 51), (101,), (10001,)):
    N = int(np.prod(dims)); R = 128
    res = {}
    for b in (0, -1):
        if b == 0: os.environ["GSS_FFTGS_ROCFFT_BATCH"] = "0"
        else: os.environ.pop("GSS_FFTGS_ROCFFT_BATCH", None)
        h = FFTGSHandle(gss.ExponentialVariogram(range=dims[0] / 10.0), dims)
        out = torch.empty((R, N), dtype=torch.float64, device="cuda")
        h.realize(1, 0, R, out=out); torch.cuda.synchronize(); t0 = time.perf_counter()
        h.realize(1, 0, R, out=out); torch.cuda.synchronize(); t1 = time.perf_counter()
        res[b] = ((t1 - t0) / R * 1e3, out.clone()); h.close()
    print("x".join(map(str, dims)), "single %.4f ms  batched %.4f ms per realisation  identical %s" % (res[0][0], res[-1][0], bool(torch.equal(res[0][1], res[-1][1]))))
