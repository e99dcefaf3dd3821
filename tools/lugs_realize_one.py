import os, sys
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")): sys.path.insert(0, p)
import numpy as np, torch, gss
from gss.engine import LUGSHandle
g=128; N=g*g; nd=N//4
cent=gss.CartesianGrid(g,g).centroids()
dl=np.sort(np.random.default_rng(5).permutation(N)[:nd]); z1=np.random.default_rng(50).normal(size=nd)
h=LUGSHandle(gss.SphericalVariogram(range=20.0), cent, dl, z1)

for i in range(3): h.realize(3, 0, 100, device=True)
torch.cuda.synchronize()
