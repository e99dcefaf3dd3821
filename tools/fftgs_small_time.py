import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]
import numpy as np, torch, gss
from gss.engine import FFTGSHandle
def t(label, fn, reps=5):
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        t0=time.perf_counter(); r=fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print("%-40s %s" % (label, " ".join("%.2f" % (1e3*x) for x in ts)), flush=True); return r
vg = gss.GaussianVariogram(range=10.0)
for dims in ((100,100),(128,128),(64,64,64),(50,50,50)):
    hs=[]
    t("create %s" % (dims,), lambda: hs.append(FFTGSHandle(vg, dims)))
    h=hs[-1]
    t("realize 3 host", lambda: h.realize(1,0,3))
    t("realize 3 device", lambda: h.realize(1,0,3,device=True))
    for x in hs: x.close()
