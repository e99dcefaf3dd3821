#!/usr/bin/env python3
"""configs[3] LUGS preprocess twice (warm-up + measured) and nothing else: workload for rocprofv3 kernel traces."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gss  # noqa: E402
from gss.engine import LUGSHandle  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 128
fact = sys.argv[2] if len(sys.argv) > 2 else "cholesky"       # "lu": the pivoted factorisation of lu.jl:70
N, nd = g * g, g * g // 4
cent = gss.CartesianGrid(g, g).centroids()
dlocs = np.sort(np.random.default_rng(5).permutation(N)[:nd])
z1 = np.random.default_rng(50).normal(size=nd)
vg = gss.SphericalVariogram(range=20.0)
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = LUGSHandle(vg, cent, dlocs, z1, factorization=fact)
    torch.cuda.synchronize()
    print("create", it, time.perf_counter() - t0)
    h.close()
