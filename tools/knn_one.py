"""One IDW call (K4 + estimator) for profiling: python3 tools/knn_one.py [n] [k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch
from gss.engine import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
m = 1_250_000
x0 = torch.as_tensor(np.random.default_rng(17).uniform(0, 100, (m, 3)), device="cuda")
x = torch.as_tensor(np.random.default_rng(16).uniform(0, 100, (n, 3)), device="cuda")
z = torch.zeros(n, dtype=torch.float64, device="cuda")
HipEngine.idw(x, z, x0, k)
torch.cuda.synchronize()
print("done")
