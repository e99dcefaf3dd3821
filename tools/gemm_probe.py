"""Rate of the generic FP64 GEMM (gss_dev_gemm) on square and SYRK-shaped problems."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
from gss import _lib
from gss._lib import check, ptr, current_stream
l = _lib.lib()
def run(M, N, K, lower):
    A = torch.randn(K, M, dtype=torch.float64, device="cuda").t()      # column-major M x K (sa_i = 1, sa_k = M)
    B = torch.randn(K, N, dtype=torch.float64, device="cuda")          # B(k, j) row-major: sb_k = N, sb_j = 1
    D = torch.zeros(N, M, dtype=torch.float64, device="cuda")          # column-major M x N
    args = (M, N, K, 1.0, ptr(A), 1, M, ptr(B), N, 1, 0.0, ptr(D), 1, M, 1 if lower else 0, current_stream())
    check(l.gss_dev_gemm(*args)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        check(l.gss_dev_gemm(*args))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    fl = 2.0 * M * N * K * (0.5 if lower else 1.0)
    print(f"M={M} N={N} K={K} lower={lower}: {dt*1e3:.2f} ms  {fl/dt/1e12:.1f} TFLOP/s", flush=True)
run(8192, 8192, 4096, False)
run(11264, 11264, 1024, True)
run(4096, 4096, 1024, True)
run(12288, 4096, 4096, False)
for mm in (2048, 3072, 5120, 6144, 7168, 9216):
    run(mm, mm, 1024, True)
