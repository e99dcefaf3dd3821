#!/usr/bin/env python3
"""Device time of gss_dev_potrf_inverse (factor and inverse, the kriging fit's core) for a few sizes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gss import _lib  # noqa: E402

l = _lib.lib()
for n in (512, 1000, 1024, 2048, 4096, 8192):
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, n + 8))
    A = torch.from_numpy(G @ G.T / n + np.eye(n)).cuda()
    W = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    best = 1e9
    for it in range(4):
        a = A.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(l.gss_dev_potrf_inverse(_lib.ptr(a), n, n, _lib.ptr(W), n, _lib.current_stream()))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    flop = 2.0 * n ** 3 / 3.0
    print(f"n = {n:5d}: {best * 1e3:8.3f} ms  {flop / best / 1e12:6.2f} TFLOP/s (incl. allocation and status read-back)")
