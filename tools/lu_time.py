#!/usr/bin/env python3
"""Wall time of gss_dev_getrf_l (partial-pivot LU, unit lower factor) at n = 12 288: the workload of the LU panel experiments."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "geostatssolvers.jl_amd"))
import numpy as np, torch
from gss import _lib
n = 12288
rng = np.random.default_rng(0)
A = rng.normal(size=(n, n)) + 3 * np.eye(n)
dA0 = torch.as_tensor(A, device="cuda")
l = _lib.lib()
for it in range(3):
    dA = dA0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = l.gss_dev_getrf_l(_lib.ptr(dA), n, n, _lib.current_stream()); torch.cuda.synchronize()
    print("rc", rc, "getrf n=12288: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
