import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/geostatssolvers.jl_amd")
import numpy as np, torch, gss
from gss.engine import KrigHandle, OK, UK
from oracle import kriging as K
from oracle.variogram import Variogram
for n, variant, okw, kw in ((8000, OK, {}, {}), (12000, UK, dict(degree=1), dict(degree=1))):
    rng = np.random.default_rng(n)
    x = rng.uniform(0, 100, (n, 3)); z = rng.normal(size=n)
    x0 = rng.uniform(0, 100, (200_000, 3))
    vg = gss.MaternVariogram(range=30.0, order=1.5, nugget=0.01)
    t0 = time.perf_counter()
    h = KrigHandle(vg, variant, x, z, **kw)
    mu, var, st = h.predict_global(torch.as_tensor(x0, device="cuda"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sel = np.linspace(0, len(x0) - 1, 40).astype(int)
    rmu, rvar = K.exactsolve(K.OK if variant == OK else K.UK, Variogram("matern", range=30.0, nu=1.5, nugget=0.01), x, z, x0[sel], **okw)
    print(n, "time %.3f s" % dt, "err", np.abs(mu.cpu().numpy()[sel] - rmu).max(), np.abs(var.cpu().numpy()[sel] - rvar).max(), int(st.sum()), flush=True)
    h.close()
