#!/usr/bin/env python3
"""A few FFTGS realisations on a grid of the generic pipeline (default 300^3) and nothing else: the workload of
rocprofv3 kernel traces / PMC passes of fftgs_generic.h.  tools/fftgs_gen_one.py [n1 n2 n3 | n1xn2] [realisations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import gss  # noqa: E402
from gss.engine import FFTGSHandle  # noqa: E402

if len(sys.argv) > 1 and "x" in sys.argv[1]:
    dims = tuple(int(a) for a in sys.argv[1].split("x"))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
else:
    args = [int(a) for a in sys.argv[1:]]
    dims = tuple(args[:3]) if len(args) >= 3 else (300, 300, 300)
    n = args[3] if len(args) > 3 else 6
f = FFTGSHandle(gss.ExponentialVariogram(range=dims[0] / 10.0), dims)
N = 1
for d in dims:
    N *= d
out = torch.empty((1, N), dtype=torch.float64, device="cuda")
import time  # noqa: E402
f.realize(4, 0, 1, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(n):
    f.realize(4, r, 1, out=out)
torch.cuda.synchronize()
print("%.3f ms per realisation (one per call)" % ((time.perf_counter() - t0) / n * 1e3))
big = torch.empty((n, N), dtype=torch.float64, device="cuda")
f.realize(4, 0, n, out=big)
torch.cuda.synchronize()
t0 = time.perf_counter()
f.realize(4, 0, n, out=big)
torch.cuda.synchronize()
print("%.3f ms per realisation (%d in one call)" % ((time.perf_counter() - t0) / n * 1e3, n))
print("variance", float((out[0] * out[0]).sum() / (N - 1)))
f.close()
