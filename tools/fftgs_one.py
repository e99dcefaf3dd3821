#!/usr/bin/env python3
"""A few 512^3 FFTGS realisations and nothing else: the workload of the rocprofv3 PMC passes
(tools/pmc_run.sh <outdir> -- tools/fftgs_one.py [edge] [realisations])."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import gss  # noqa: E402
from gss.engine import FFTGSHandle  # noqa: E402

e = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
f = FFTGSHandle(gss.ExponentialVariogram(range=50.0 * e / 512.0), (e, e, e))
out = torch.empty((1, e ** 3), dtype=torch.float64, device="cuda")
for r in range(n):
    f.realize(4, r, 1, out=out)
torch.cuda.synchronize()
print("variance", float((out[0] * out[0]).sum() / (e ** 3 - 1)))
f.close()
