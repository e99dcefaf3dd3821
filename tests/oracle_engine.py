"""Stand-in engine for CPU-only tests of the HOST logic (parameter handling, sharding, collectives).

It has the interface of gss.engine.HipEngine but computes with the oracle.  It lives under tests/ on
purpose: the product never imports it, and GPU parity tests never use it."""
import numpy as np

from oracle import fftgs as OF, kriging as OK_, lugs as OL, philox
from oracle.variogram import Variogram, cov_pairwise


def _ovg(vg):
    if getattr(vg, "kind", None) == "nested":
        from oracle.variogram import Nested
        return Nested([(w, _ovg(m)) for w, m in vg.terms])
    return Variogram(vg.kind, sill=vg.sill, nugget=vg.nugget, range=vg.range, nu=vg.nu, radii=vg.radii,
                     regularize=getattr(vg, "regularize", True))


class _Krig:
    def __init__(self, vg, variant, xdata, z, mean=0.0, degree=0, drift_data=None, factor=True, async_fit=False):
        self.vg, self.variant = _ovg(vg), variant
        self.x = np.atleast_2d(np.asarray(xdata, dtype=np.float64))
        self.z = np.asarray(z, dtype=np.float64)
        self.mean, self.degree, self.drift_data = mean or 0.0, degree, drift_data

    def close(self):
        pass

    def set_block_support(self, cell, nsub=3):
        self.support = None if not nsub else (np.broadcast_to(np.asarray(cell, dtype=np.float64), (self.x.shape[1],)), nsub)

    def predict_global(self, xdom, drift_dom=None):
        mu, var = OK_.exactsolve(self.variant, self.vg, self.x, self.z, xdom, mean=self.mean, degree=self.degree,
                                 drift_data=self.drift_data, drift_dom=drift_dom, support=getattr(self, "support", None))
        return mu, var, np.zeros(len(mu), dtype=np.uint8)

    def predict_knn(self, xdom, k, minneighbors=1, radius=None, radii=None, drift_dom=None, return_idx=False,
                    distance=None):
        return OK_.approxsolve(self.variant, self.vg, self.x, self.z, xdom, k, minneighbors, mean=self.mean,
                               degree=self.degree, drift_data=self.drift_data, drift_dom=drift_dom, radius=radius,
                               radii=radii, return_idx=return_idx, distance=distance, support=getattr(self, "support", None))

    def predict_global_batch(self, xdom, zbatch):
        return np.stack([OK_.exactsolve(self.variant, self.vg, self.x, zb, xdom, mean=self.mean, degree=self.degree)[0]
                         for zb in zbatch])


class _FFTGS:
    def __init__(self, vg, dims, spacing=None, mean=0.0):
        self.pre = OF.preprocess(_ovg(vg), dims, spacing=spacing, mean=mean)
        self.N = int(np.prod(dims))

    def close(self):
        pass

    def spectrum(self):
        return self.pre.F.ravel()

    def realize(self, seed, first_real, nreals, noise=None, inds=None, out=None, device=False):
        if noise is not None:
            return np.stack([OF.solvesingle(self.pre, noise[r], inds) for r in range(nreals)])
        return OF.realize(self.pre, seed, first_real, nreals, inds) if nreals else np.empty((0, self.N))


class _LUGS:
    """`factor=False` + `state_tensor()` / `adopt_state()` mirror LUGSHandle so that the CPU-only multi-process test
    drives parallel.replicate_state (rank 0 factorises, the peers receive L22 and d2 by broadcast)."""
    ncomputed = 0          # class-wide count of factorisations actually carried out (the test reads it)

    def __init__(self, vg, centroids, dlocs, z1, mean=0.0, factor=True, factorization="cholesky"):
        c = np.asarray(centroids, dtype=np.float64)
        dl = np.asarray(dlocs, dtype=np.int64)
        if factor:
            self.p = OL.preprocess(_ovg(vg), c, c[dl] if dl.size else None, np.asarray(z1) if dl.size else None,
                                   mean=mean, factorization=factorization)
            _LUGS.ncomputed += 1
        else:
            mask = np.zeros(c.shape[0], dtype=bool)
            mask[dl] = True
            ns = int((~mask).sum())
            self.p = OL.LUGSParams(np.asarray(z1, dtype=np.float64), np.zeros(ns), np.zeros((ns, ns)),
                                   0.0 if mean is None else float(mean), dl, np.flatnonzero(~mask))
        self.N, self.ns = c.shape[0], self.p.slocs.size
        self._state = np.concatenate([self.p.L22.ravel(), self.p.d2])

    def state_tensor(self):
        import torch
        return torch.from_numpy(self._state)       # aliases self._state: a broadcast writes into it

    def adopt_state(self):
        ns = self.ns
        self.p.L22 = self._state[: ns * ns].reshape(ns, ns).copy()
        self.p.d2 = self._state[ns * ns:].copy()

    def close(self):
        pass

    def realize(self, seed, first_real, nreals, noise=None, rho=None, w1=None, device=False):
        ys, ws = [], []
        for r in range(nreals):
            w2 = noise[r] if noise is not None else philox.normal(seed, first_real + r, self.ns)
            y, _ = OL.lusim(self.p, w2, rho, None if w1 is None else w1[r])
            ys.append(y)
            ws.append(w2)
        return (np.stack(ys), np.stack(ws)) if nreals else (np.empty((0, self.N)), np.empty((0, self.ns)))


class _SGS:
    def __init__(self, vg, centroids, path, dlocs, zdata, mean=0.0, maxneighbors=10, minneighbors=1, radius=None,
                 radii=None, path_base=0, mask_after_search=False, distance=None):
        self.a = (_ovg(vg), mean, np.asarray(centroids, dtype=np.float64))
        self.path = None if path is None else np.asarray(path, dtype=np.int64)
        self.path_base = path_base
        self.d = (np.asarray(dlocs, dtype=np.int64), np.asarray(zdata, dtype=np.float64))
        self.kw = dict(maxneighbors=maxneighbors, minneighbors=minneighbors, radius=radius, radii=radii,
                       mask_after_search=mask_after_search, distance=distance)

    def close(self):
        pass

    def realize(self, seed, first_real, nreals, noise=None):
        from oracle import sgs
        if self.path is not None and self.path.ndim == 2:      # one visiting order per realisation
            return np.stack([sgs.realize(*self.a, self.path[first_real + r - self.path_base], *self.d, seed,
                                         first_real + r, 1, **self.kw)[0] for r in range(nreals)])
        return sgs.realize(*self.a, self.path, *self.d, seed, first_real, nreals, **self.kw)


class OracleEngine:
    name = "oracle-stand-in"
    Krig, FFTGS, LUGS, SGS = _Krig, _FFTGS, _LUGS, _SGS

    @staticmethod
    def cov_pairwise(vg, a, b=None):
        return cov_pairwise(_ovg(vg), a, b)

    @staticmethod
    def knn_search(xdata, centers, k, radius=None, radii=None, distance=None):
        return OK_.knn_search(xdata, centers, k, radius, radii, distance)

    @staticmethod
    def idw(xdata, z, xdom, k, minneighbors=1, exponent=1.0, radius=None, radii=None, distance=None):
        from oracle import idw_lwr
        z = np.asarray(z, dtype=np.float64)
        if z.ndim == 2:                                   # value columns (nz, n) that share the search: column by column
            res = [idw_lwr.idw(xdata, zc, xdom, k, minneighbors, exponent, radius, radii, distance) for zc in z]
            return np.stack([r[0] for r in res]), res[0][1], res[0][2]
        return idw_lwr.idw(xdata, z, xdom, k, minneighbors, exponent, radius, radii, distance)

    @staticmethod
    def lwr(xdata, z, xdom, k, minneighbors=1, weight=(0, 3.0, 2.0), radius=None, radii=None, distance=None):
        from oracle import idw_lwr
        kind, a, p = weight
        wf = idw_lwr.tricube if kind == 1 else idw_lwr.exp_weight(a, p)
        z = np.asarray(z, dtype=np.float64)
        if z.ndim == 2:
            res = [idw_lwr.lwr(xdata, zc, xdom, k, minneighbors, wf, radius, radii, distance) for zc in z]
            return np.stack([r[0] for r in res]), res[0][1], res[0][2]
        return idw_lwr.lwr(xdata, z, xdom, k, minneighbors, wf, radius, radii, distance)
