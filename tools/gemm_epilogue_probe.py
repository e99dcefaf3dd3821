"""How much of a generic-GEMM launch is its epilogue?  Same M x N output, K from 16 to 4096."""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/geostatssolvers.jl_amd")
import torch
from gss import _lib
l = _lib.lib()
M, N = 12288, 4096
g = torch.Generator(device="cuda").manual_seed(1)
for beta in (0.0, 1.0):
    for K in (16, 256, 1024, 4096):
        A = torch.randn((K, M), dtype=torch.float64, device="cuda", generator=g)
        B = torch.randn((K, N), dtype=torch.float64, device="cuda", generator=g)
        D = torch.zeros((N, M), dtype=torch.float64, device="cuda")
        def run():
            _lib.check(l.gss_dev_gemm(M, N, K, 1.0, _lib.ptr(A), 1, M, _lib.ptr(B), N, 1, beta, _lib.ptr(D), 1, M, 0, None))
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("beta %.0f K %5d  %.3f ms  %.1f TFLOP/s" % (beta, K, dt * 1e3, 2.0 * M * N * K / dt / 1e12), flush=True)
