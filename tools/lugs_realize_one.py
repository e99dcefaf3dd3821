#!/usr/bin/env python3
"""Three gss_lugs_realize calls of configs[3] (100 realisations each) and nothing else after the preprocess: workload
for rocprofv3 kernel traces of the realisation step (L22 W cut along K over four streams)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gss  # noqa: E402
from gss.engine import LUGSHandle  # noqa: E402

g = 128
N, nd = g * g, g * g // 4
cent = gss.CartesianGrid(g, g).centroids()
dl = np.sort(np.random.default_rng(5).permutation(N)[:nd])
z1 = np.random.default_rng(50).normal(size=nd)
h = LUGSHandle(gss.SphericalVariogram(range=20.0), cent, dl, z1)
for i in range(3):
    h.realize(3, 0, 100, device=True)
torch.cuda.synchronize()
h.close()
