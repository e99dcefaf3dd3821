"""Products of the shapes of the late steps of the blocked Cholesky (m x 1024 x 1024 and the lower tiles of m x m x 1024):
what the tile-shape rule of gemm_f64 (csrc/dense_la.hip) was fitted on.  python3 tools/gemm_late_steps.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
from gss import _lib
from gss._lib import check, ptr, current_stream
l = _lib.lib()
def run(M, N, K, lower):
    A = torch.randn(K, M, dtype=torch.float64, device="cuda").t()
    B = torch.randn(K, N, dtype=torch.float64, device="cuda")
    D = torch.zeros(N, M, dtype=torch.float64, device="cuda")
    args = (M, N, K, 1.0, ptr(A), 1, M, ptr(B), N, 1, 0.0, ptr(D), 1, M, 1 if lower else 0, current_stream())
    check(l.gss_dev_gemm(*args)); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); check(l.gss_dev_gemm(*args)); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = min(ts)
    fl = 2.0 * M * N * K * (0.5 if lower else 1.0)
    print(f"M={M:5d} N={N:5d} K={K} lower={int(lower)}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s", flush=True)
for m in (1024, 2048, 3072, 4096, 5120, 6144, 8192, 10240):
    run(m, 1024, 1024, False)
for m in (1024, 2048, 3072, 4096, 5120, 6144, 8192):
    run(m, m, 1024, True)
