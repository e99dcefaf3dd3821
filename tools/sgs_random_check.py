"""Random shapes: the level schedule of SGS against the walk along the path, bit for bit (two child processes)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r]
import gss
from gss.engine import SGSHandle
rng = np.random.default_rng(int(sys.argv[2]))
res = []
for it in range(int(sys.argv[3])):
    d = int(rng.integers(1, 4))
    dims = tuple(int(v) for v in rng.integers(5, {1: 3000, 2: 90, 3: 22}[d], d))
    N = int(np.prod(dims))
    g = np.meshgrid(*[np.arange(n) + 0.5 for n in dims], indexing="ij")
    cent = np.stack([a.ravel(order="F") for a in g], 1)
    k = int(rng.integers(1, min(N - 1, 90)))
    nd = int(rng.integers(0, min(N // 2, 40)))
    dl = np.sort(rng.choice(N, nd, replace=False)); zd = rng.normal(size=nd)
    mode = int(rng.integers(0, 3))
    path = None if mode == 0 else (rng.permutation(N) if mode == 1 else np.stack([rng.permutation(N) for _ in range(3)]))
    R = 3 if mode == 2 else int(rng.integers(1, 130))
    h = SGSHandle(gss.SphericalVariogram(range=float(rng.uniform(2, 15)), nugget=0.05), cent, path, dl, zd, 0.2, k,
                  int(rng.integers(1, 3)), float(rng.uniform(3, 30)) if rng.random() < 0.5 else None,
                  mask_after_search=bool(rng.integers(0, 2)))
    res.append(h.realize(11, 0, R)); h.close()
np.savez(sys.argv[1], *res)
''' % (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd"))
import numpy as np
seed = sys.argv[1] if len(sys.argv) > 1 else "0"
count = sys.argv[2] if len(sys.argv) > 2 else "40"
with tempfile.TemporaryDirectory() as d:
    outs = []
    for sw in ("1", "0"):
        out = os.path.join(d, "o%s.npz" % sw)
        subprocess.run([sys.executable, "-c", CODE, out, seed, count], check=True, env=dict(os.environ, GSS_SGS_LEVELS=sw), timeout=900)
        with np.load(out) as f:
            outs.append([f[k] for k in f.files])
bad = [i for i, (a, b) in enumerate(zip(*outs)) if not (a.shape == b.shape and np.array_equal(a, b))]
print("%d cases, %d differ" % (len(outs[0]), len(bad)), bad[:5])
sys.exit(1 if bad else 0)
