"""Does skipping the k-tiles that only meet zeros of a triangular operand shorten a generic-GEMM launch?"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/geostatssolvers.jl_amd")
import torch
from gss import _lib
l = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K) in ((12288, 4096, 4096), (4096, 4096, 4096)):
    A = torch.randn((K, M), dtype=torch.float64, device="cuda", generator=g)
    B = torch.randn((K, N), dtype=torch.float64, device="cuda", generator=g)
    D = torch.zeros((N, M), dtype=torch.float64, device="cuda")
    for flag, name in ((0, "none"), (2, "B upper: kend by column"), (4, "B lower: kbeg by column"), (8, "A lower: kend by row")):
        def run():
            _lib.check(l.gss_dev_gemm(M, N, K, 1.0, _lib.ptr(A), 1, M, _lib.ptr(B), N, 1, 0.0, _lib.ptr(D), 1, M, flag, None))
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): run()
        torch.cuda.synchronize()
        print(M, N, K, "%-26s %.3f ms" % (name, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
