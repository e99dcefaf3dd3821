#!/usr/bin/env python3
"""Phase boundaries of the single-launch factor-and-inverse kernel (GSS_PANEL_TIMES=1 makes the library print them):
per step k the stamps are  [leaf done] [barrier] [phase 2 done] [barrier] [phase 3 done] [barrier]  in microseconds."""
import os
import sys

os.environ["GSS_PANEL_TIMES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gss import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
l = _lib.lib()
rng = np.random.default_rng(0)
G = rng.normal(size=(n, n + 8))
A = G @ G.T / n + np.eye(n)
for it in range(3):
    dA = torch.from_numpy(A.copy()).cuda()
    dW = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    _lib.check(l.gss_dev_potrf_inverse(_lib.ptr(dA), n, n, _lib.ptr(dW), n, _lib.current_stream()))
W = dW.cpu().numpy().T
print("residual", np.max(np.abs(W @ np.linalg.cholesky(A) - np.eye(n))))
