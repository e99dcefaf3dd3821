#!/usr/bin/env python3
"""One-off robustness sweep of gss_dev_potrf_inverse (single-launch panel, ragged sizes through padded copies, splits
above 1 024 rows): random sizes, random leading dimensions, ill-conditioned spectra; prints the worst residuals."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gss import _lib  # noqa: E402

l = _lib.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = (0.0, 0.0, 0)
for it in range(count):
    n = int(rng.integers(130, 2600))
    lda, ldw = n + int(rng.integers(0, 9)), n + int(rng.integers(0, 9))
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    ev = 10.0 ** rng.uniform(-4, 1, size=n)                     # condition number up to 1e5
    A = (Q * ev) @ Q.T
    A = 0.5 * (A + A.T)
    bufA = np.zeros((n, lda))
    bufA[:, :n] = A
    dA = torch.from_numpy(bufA).cuda()
    dW = torch.full((n, ldw), 3.0, dtype=torch.float64, device="cuda")
    _lib.check(l.gss_dev_potrf_inverse(_lib.ptr(dA), n, lda, _lib.ptr(dW), ldw, _lib.current_stream()))
    L = np.tril(dA.cpu().numpy()[:, :n].T)
    W = dW.cpu().numpy()[:, :n].T
    rL = np.max(np.abs(L @ L.T - A)) / np.max(np.abs(A))
    rW = np.max(np.abs(W @ L - np.eye(n)))
    assert np.array_equal(np.triu(W, 1), np.zeros((n, n)))
    if rW > worst[1]:
        worst = (rL, rW, n)
    if rL > 1e-12 or rW > 1e-7:
        print("LARGE", n, lda, ldw, rL, rW)
print("matrices", count, "worst |LL' - A| / |A|, |W L - I|, n:", worst)
