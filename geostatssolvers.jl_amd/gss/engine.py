"""Array-level engine over the C-ABI: the one place where Python touches libgss_hip.so.

Every method takes numpy arrays (host: the library stages them through PCIe) or CUDA torch tensors
(device: zero-copy, asynchronous on torch's current stream) and returns the same kind.
`HipEngine` is the only engine the product ships; the solver front-ends take an `engine=` argument
so that the CPU-only multi-process tests can exercise the sharding/host logic with a stand-in.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import MEM_DEVICE, MEM_HOST, check, current_stream, is_torch, make_variogram, ptr

SK, OK, UK, EDK = 0, 1, 2, 3


def _vg_struct(vg, dim, extent=None):
    if getattr(vg, "kind", None) == "power":
        # pseudo-covariance A - gamma(h): A = 2 gamma(diameter of the data box) keeps the data block positive
        # definite; constrained kriging results do not depend on A (gss.h, GSS_VG_POWER)
        if extent is None:
            raise ValueError("a power variogram is not stationary: only kriging with data can use it")
        a = 2.0 * (vg.range * max(float(extent), 1e-300) ** vg.nu) + vg.nugget + 1e-300
        return make_variogram("power", dim, a, vg.nugget, vg.range, vg.nu, None)
    if getattr(vg, "kind", None) == "nested":
        # first structure carries the total nugget; every structure contributes c_i (sill_i - nugget_i); a Gaussian
        # structure's nugget is the regularised one (variograms.py: nugget + 1e-6 unless regularize=False)
        for w, m in vg.terms:
            if m.effective_nugget > m.sill:
                # (a Gaussian structure with nugget = sill: the 1e-6 of the regularisation would make its structured part
                #  negative -- the single model is refused by the library for the same reason)
                raise ValueError(f"nested variogram: the (regularised) nugget {m.effective_nugget} of a {m.kind} structure "
                                 f"exceeds its sill {m.sill}")
        terms = [(w, m) for w, m in vg.terms if w * (m.sill - m.effective_nugget) > 0.0]
        if not terms:
            raise ValueError("nested variogram without a structured (non-nugget) component")
        w0, m0 = terms[0]
        nug = vg.effective_nugget
        extras = [(m.kind, w * (m.sill - m.effective_nugget), m.range, m.nu, m.radii) for w, m in terms[1:]]
        return make_variogram(m0.kind, dim, w0 * (m0.sill - m0.effective_nugget) + nug, nug, m0.range, m0.nu, m0.radii,
                              extras)
    return make_variogram(vg.kind, dim, vg.sill, getattr(vg, "effective_nugget", vg.nugget), vg.range, vg.nu, vg.radii)


def _extent(x):
    """Diameter of the bounding box of point-major coordinates (numpy or CUDA tensor)."""
    if is_torch(x):
        return float(((x.max(dim=0).values - x.min(dim=0).values) ** 2).sum().sqrt())
    x = np.asarray(x)
    return float(np.sqrt(((x.max(axis=0) - x.min(axis=0)) ** 2).sum()))


def _space(x):
    """MEM_DEVICE for CUDA tensors; numpy arrays and CPU tensors (pinned ones included) are host memory."""
    return MEM_DEVICE if is_torch(x) and x.is_cuda else MEM_HOST


def _empty_like_space(ref, shape, dtype):
    if is_torch(ref):
        import torch
        tdt = {np.float64: torch.float64, np.uint8: torch.uint8, np.int32: torch.int32}[dtype]
        return torch.empty(shape, dtype=tdt, device=ref.device)
    return np.empty(shape, dtype=dtype)


def _prep_in(x, dtype=np.float64):
    """Contiguous array of the right dtype in the space it already lives in."""
    if x is None:
        return None
    if is_torch(x):
        import torch
        tdt = {np.float64: torch.float64, np.int64: torch.int64}[dtype]
        if not x.is_cuda:
            return np.ascontiguousarray(x.numpy(), dtype=dtype)
        return x.to(tdt).contiguous()
    return np.ascontiguousarray(x, dtype=dtype)


def to_host(t):
    """numpy copy of a contiguous CUDA tensor through the library's transfer path (gss_dev_to_host: pinned bounce
    buffers, ~45 GB/s into pageable memory where `tensor.cpu()` manages ~10)."""
    import torch
    t = t.contiguous()
    dt = {torch.float64: np.float64, torch.int32: np.int32, torch.uint8: np.uint8, torch.int64: np.int64}[t.dtype]
    out = np.empty(tuple(t.shape), dtype=dt)
    check(_lib.lib().gss_dev_to_host(ptr(out), C.c_void_p(t.data_ptr()), out.nbytes, current_stream()))
    return out


def _alias_tensor(owner, dev_ptr, nbytes):
    """CUDA float64 tensor aliasing `nbytes` of library-owned HBM at `dev_ptr` (for torch.distributed.broadcast over
    RCCL); it keeps `owner` (the handle) alive."""
    import torch

    class _Alias:
        __cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (dev_ptr, False),
                                    "version": 3, "strides": None}
    t = torch.as_tensor(_Alias(), device=f"cuda:{torch.cuda.current_device()}")
    t._gss_keepalive = owner
    return t


class _NativeState:
    """The library's own replication routes (gss.h "multi-GPU"): RCCL broadcast inside the library and HIP IPC."""
    _STATE_KIND = None

    def bcast_state(self, root=0):
        """ncclBroadcast of the state on the communicator of `parallel.native_comm()`; the peers adopt."""
        _lib.state_bcast(self._STATE_KIND, self._h, root)

    def export_state(self) -> bytes:
        """80-byte token another process passes to `import_state` (keep this handle alive until it has)."""
        return _lib.state_ipc_export(self._STATE_KIND, self._h)

    def import_state(self, token: bytes):
        """Pull the owner's state into this handle (created without factor / spectrum) and adopt it."""
        _lib.state_ipc_import(self._STATE_KIND, self._h, token)


class KrigHandle(_NativeState):
    _STATE_KIND = _lib.STATE_KRIG

    """gss_krig_t*: fitted kriging system living in HBM."""

    def __init__(self, vg, variant, xdata, z, mean=0.0, degree=0, drift_data=None, factor=True, async_fit=False):
        """`async_fit`: GSS_KRIG_ASYNC_FIT -- the constructor returns once the fit is queued; the first global
        prediction assembles its right-hand sides beside it and reports the fit's status itself (the order of
        `solve`: fit, then predict, krig.jl:166-186)."""
        self._l = _lib.lib()
        x = np.ascontiguousarray(xdata, dtype=np.float64)
        if x.ndim == 1:
            x = x[:, None]
        self.n, self.dim = x.shape
        zz = np.ascontiguousarray(z, dtype=np.float64)
        dd = None if drift_data is None else np.ascontiguousarray(drift_data, dtype=np.float64).reshape(self.n, -1)
        self.ndrift = 0 if dd is None else dd.shape[1]
        self.variant = variant
        h = C.c_void_p()
        v = _vg_struct(vg, self.dim, extent=_extent(x))
        check(self._l.gss_krig_create(C.byref(h), C.byref(v), variant, float(mean or 0.0), int(degree or 0),
                                      self.ndrift, ptr(x), ptr(zz), ptr(dd), self.n,
                                      (0 if factor else _lib.KRIG_NO_FACTOR) |
                                      (_lib.KRIG_ASYNC_FIT if (async_fit and factor) else 0), current_stream()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._l.gss_krig_destroy(self._h)
            self._h = None

    __del__ = close

    def factor_tensor(self):
        """CUDA tensor aliasing the factor state (for torch.distributed.broadcast over RCCL)."""
        p, nb = C.c_void_p(), C.c_int64()
        check(self._l.gss_krig_factor_buffer(self._h, C.byref(p), C.byref(nb)))
        return _alias_tensor(self, p.value, nb.value)

    def adopt_factor(self):
        check(self._l.gss_krig_adopt_factor(self._h))

    state_tensor, adopt_state = factor_tensor, adopt_factor      # the names parallel.replicate_state uses

    def set_block_support(self, cell, nsub=3):
        """Regularise the right-hand sides of `predict_global` over cells of size `cell` (gss.h,
        gss_krig_set_block_support); `nsub=0` returns to point support."""
        c = None if not nsub else np.ascontiguousarray(np.broadcast_to(np.asarray(cell, dtype=np.float64), (self.dim,)))
        check(self._l.gss_krig_set_block_support(self._h, ptr(c), int(nsub or 0), current_stream()))

    def predict_global(self, xdom, drift_dom=None):
        xdom = _prep_in(xdom)
        m = xdom.shape[0]
        mem = _space(xdom)
        mean = _empty_like_space(xdom, (m,), np.float64)
        var = _empty_like_space(xdom, (m,), np.float64)
        status = _empty_like_space(xdom, (m,), np.uint8)
        dd = _prep_in(drift_dom)
        check(self._l.gss_krig_predict_global(self._h, ptr(xdom), ptr(dd), m, ptr(mean), ptr(var), ptr(status),
                                              mem, current_stream()))
        return mean, var, status

    def predict_knn(self, xdom, k, minneighbors=1, radius=None, radii=None, drift_dom=None, return_idx=False,
                    distance=None):
        xdom = _prep_in(xdom)
        m = xdom.shape[0]
        mem = _space(xdom)
        mean = _empty_like_space(xdom, (m,), np.float64)
        var = _empty_like_space(xdom, (m,), np.float64)
        status = _empty_like_space(xdom, (m,), np.uint8)
        idx = _empty_like_space(xdom, (m, k), np.int32) if return_idx else None
        cnt = _empty_like_space(xdom, (m,), np.int32) if return_idx else None
        ir = None if radii is None else np.ascontiguousarray(1.0 / np.asarray(radii, dtype=np.float64))
        r = -1.0 if radius is None and radii is None else (1.0 if radii is not None else float(radius))
        dd = _prep_in(drift_dom)
        met, mpar = _lib.metric_spec(distance)
        check(self._l.gss_krig_predict_knn(self._h, ptr(xdom), ptr(dd), m, int(k), int(minneighbors), r, ptr(ir),
                                           met, mpar, ptr(mean), ptr(var), ptr(status), ptr(idx), ptr(cnt), mem,
                                           current_stream()))
        if return_idx:
            return mean, var, status, idx, cnt
        return mean, var, status

    def predict_global_batch(self, xdom, zbatch):
        xdom = _prep_in(xdom)
        zb = _prep_in(zbatch)
        if _space(zb) != _space(xdom):
            raise ValueError("xdom and zbatch must live in the same memory space")
        nb = zb.shape[0]
        m = xdom.shape[0]
        out = _empty_like_space(xdom, (nb, m), np.float64)
        check(self._l.gss_krig_predict_global_batch(self._h, ptr(xdom), m, ptr(zb), nb, ptr(out), _space(xdom),
                                                    current_stream()))
        return out


class FFTGSHandle(_NativeState):
    _STATE_KIND = _lib.STATE_FFTGS

    """gss_fftgs_t*: spectral amplitude + rocFFT plans for one variable."""

    def __init__(self, vg, dims, spacing=None, mean=0.0, spectrum=True):
        """`spectrum=False`: allocate the state only; it arrives by `state_tensor()` broadcast + `adopt_state()`."""
        self._l = _lib.lib()
        self.dims = tuple(int(d) for d in dims)
        nd = len(self.dims)
        d = (C.c_int64 * 3)(*(list(self.dims) + [1] * (3 - nd)))
        sp = (C.c_double * 3)(*([float(s) for s in (spacing if spacing is not None else [1.0] * nd)] + [1.0] * (3 - nd)))
        v = _vg_struct(vg, nd)
        h = C.c_void_p()
        check(self._l.gss_fftgs_create(C.byref(h), C.byref(v), nd, d, sp, float(mean),
                                       0 if spectrum else _lib.FFTGS_NO_SPECTRUM, current_stream()))
        self._h = h
        self.N = int(np.prod(self.dims))

    def state_tensor(self):
        """CUDA tensor aliasing the spectral state (fft.jl:62-103 runs on one rank, the peers receive this)."""
        p, nb = C.c_void_p(), C.c_int64()
        check(self._l.gss_fftgs_state_buffer(self._h, C.byref(p), C.byref(nb)))
        return _alias_tensor(self, p.value, nb.value)

    def adopt_state(self):
        check(self._l.gss_fftgs_adopt_state(self._h, current_stream()))

    def close(self):
        if getattr(self, "_h", None):
            self._l.gss_fftgs_destroy(self._h)
            self._h = None

    __del__ = close

    def spectrum(self):
        out = np.empty(self.N)
        check(self._l.gss_fftgs_spectrum(self._h, ptr(out), MEM_HOST, current_stream()))
        return out

    def realize(self, seed, first_real, nreals, noise=None, inds=None, out=None, device=False, pinned=False):
        """nreals x npts realisations; `device=True` (or a CUDA `out`/`noise`) keeps them in HBM.  Host results leave
        the device chunk by chunk while the next realisations are computed (at most three chunks of ~256 MiB staged in
        HBM, csrc OutStream); `pinned=True` returns a numpy view of page-locked memory, which the DMA engine writes
        directly (a pageable array goes through the library's pinned bounce buffers); `out` may be a numpy array or a
        CPU / pinned / CUDA tensor of shape (nreals, npts)."""
        noise = _prep_in(noise)
        npts = self.N if inds is None else len(inds)
        if out is None:
            if device or (is_torch(noise) and noise.is_cuda):
                import torch
                out = torch.empty((nreals, npts), dtype=torch.float64, device="cuda")
            elif pinned:
                import torch
                out = torch.empty((nreals, npts), dtype=torch.float64, pin_memory=True)
            else:
                out = np.empty((nreals, npts))
        mem = _space(out)
        if noise is not None and _space(noise) != mem:
            raise ValueError("noise and out must live in the same memory space")
        ii = None
        if inds is not None:
            if mem == MEM_DEVICE:
                import torch
                ii = torch.as_tensor(np.asarray(inds, dtype=np.int64), device=out.device)
            else:
                ii = np.ascontiguousarray(inds, dtype=np.int64)
        check(self._l.gss_fftgs_realize(self._h, int(seed), int(first_real), int(nreals), ptr(noise), ptr(ii),
                                        0 if inds is None else npts, ptr(out), mem, current_stream()))
        return out.numpy() if pinned and is_torch(out) and not out.is_cuda else out


class LUGSHandle(_NativeState):
    _STATE_KIND = _lib.STATE_LUGS

    """gss_lugs_t*: d2 and L22 in HBM for one variable."""

    def __init__(self, vg, centroids, dlocs, z1, mean=0.0, factor=True, factorization="cholesky"):
        """`factor=False`: allocate the state only; it arrives by `state_tensor()` broadcast + `adopt_state()`.
        `factorization`: "cholesky" (default) or "lu" (lu.jl:70: the unit lower factor of a pivoted LU)."""
        if factorization not in ("cholesky", "lu"):
            raise ValueError(f"factorization={factorization!r}: 'cholesky' or 'lu'")
        self._l = _lib.lib()
        c = np.ascontiguousarray(centroids, dtype=np.float64)
        if c.ndim == 1:
            c = c[:, None]
        self.N, dim = c.shape
        dl = np.ascontiguousarray(dlocs, dtype=np.int64)
        zz = np.ascontiguousarray(z1, dtype=np.float64)
        v = _vg_struct(vg, dim)
        h = C.c_void_p()
        check(self._l.gss_lugs_create(C.byref(h), C.byref(v), ptr(c), self.N, ptr(dl), ptr(zz), dl.size,
                                      float(mean), (0 if factor else _lib.LUGS_NO_FACTOR)
                                      | (_lib.LUGS_FACT_LU if factorization == "lu" else 0), current_stream()))
        self._h = h
        self.nd = int(dl.size)
        self.ns = self.N - self.nd

    def state_tensor(self):
        """CUDA tensor aliasing L22 and d2 (lu.jl:76-169 runs on one rank, the peers receive this)."""
        p, nb = C.c_void_p(), C.c_int64()
        check(self._l.gss_lugs_state_buffer(self._h, C.byref(p), C.byref(nb)))
        return _alias_tensor(self, p.value, nb.value)

    def adopt_state(self):
        check(self._l.gss_lugs_adopt_state(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._l.gss_lugs_destroy(self._h)
            self._h = None

    __del__ = close

    def factor(self):
        l22 = np.empty((self.ns, self.ns))
        d2 = np.empty(self.ns)
        check(self._l.gss_lugs_factor(self._h, ptr(l22), ptr(d2), MEM_HOST, current_stream()))
        return l22.T.copy(), d2          # column-major on the wire -> numpy (row, col)

    def realize(self, seed, first_real, nreals, noise=None, rho=None, w1=None, device=False):
        noise = _prep_in(noise)
        w1 = _prep_in(w1)
        if device or is_torch(noise) or is_torch(w1):
            import torch
            out = torch.empty((nreals, self.N), dtype=torch.float64, device="cuda")
            wout = torch.empty((nreals, self.ns), dtype=torch.float64, device="cuda")
        else:
            out = np.empty((nreals, self.N))
            wout = np.empty((nreals, self.ns))
        check(self._l.gss_lugs_realize(self._h, int(seed), int(first_real), int(nreals), ptr(noise),
                                       0.0 if rho is None else float(rho), ptr(w1), ptr(out), ptr(wout), _space(out),
                                       current_stream()))
        return out, wout


class SGSHandle:
    """gss_sgs_t*: neighbour lists, simple-kriging weights and sigmas of every path node in HBM.

    `path`: None (LinearPath), a visiting order of N cells shared by every realisation, or an (npaths, N) array with
    one visiting order per realisation -- row p belongs to realisation `path_base + p` (seq.jl:99-102)."""

    def __init__(self, vg, centroids, path, dlocs, zdata, mean=0.0, maxneighbors=10, minneighbors=1, radius=None,
                 radii=None, path_base=0, mask_after_search=False, distance=None):
        """`mask_after_search`: GSS_SGS_MASK_AFTER_SEARCH (the k nearest cells of the whole domain, then the simulated
        ones) instead of the k nearest among the simulated cells.  `distance`: the search metric (euclidean,
        cityblock, chebyshev)."""
        self._l = _lib.lib()
        c = np.ascontiguousarray(centroids, dtype=np.float64)
        if c.ndim == 1:
            c = c[:, None]
        self.N, dim = c.shape
        self.k = int(maxneighbors)
        pa = None if path is None else np.ascontiguousarray(path, dtype=np.int64)
        npaths = 1
        if pa is not None and pa.ndim == 2:
            npaths = pa.shape[0]
            if pa.shape[1] != self.N:
                raise ValueError(f"paths must have {self.N} cells each")
        dl = np.ascontiguousarray(dlocs if dlocs is not None else [], dtype=np.int64)
        zd = np.ascontiguousarray(zdata if zdata is not None else [], dtype=np.float64)
        ir = None if radii is None else np.ascontiguousarray(1.0 / np.asarray(radii, dtype=np.float64))
        r = -1.0 if radius is None and radii is None else (1.0 if radii is not None else float(radius))
        v = _vg_struct(vg, dim)
        h = C.c_void_p()
        check(self._l.gss_sgs_create_paths(C.byref(h), C.byref(v), float(mean), ptr(c), self.N, dim, ptr(pa), npaths,
                                           int(path_base), ptr(dl), ptr(zd), dl.size, self.k, int(minneighbors), r,
                                           ptr(ir), (_lib.SGS_MASK_AFTER_SEARCH if mask_after_search else 0) |
                                           (_lib.metric_spec(distance)[0] << _lib.SGS_METRIC_SHIFT),
                                           current_stream()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._l.gss_sgs_destroy(self._h)
            self._h = None

    __del__ = close

    def weights(self):
        idx = np.empty((self.N, self.k), dtype=np.int32)
        nc = np.empty(self.N, dtype=np.int32)
        w = np.empty((self.N, self.k))
        sg = np.empty(self.N)
        check(self._l.gss_sgs_weights(self._h, ptr(idx), ptr(nc), ptr(w), ptr(sg), MEM_HOST, current_stream()))
        return idx, nc, w, sg

    def realize(self, seed, first_real, nreals, noise=None, device=False, out=None):
        noise = _prep_in(noise)
        if out is not None:
            if tuple(out.shape) != (nreals, self.N):
                raise ValueError(f"out must have shape ({nreals}, {self.N})")
        elif device or is_torch(noise):
            import torch
            out = torch.empty((nreals, self.N), dtype=torch.float64, device="cuda")
        else:
            out = np.empty((nreals, self.N))
        check(self._l.gss_sgs_realize(self._h, int(seed), int(first_real), int(nreals), ptr(noise), ptr(out),
                                      _space(out), current_stream()))
        return out


class HipEngine:
    """The product engine: every call lands in a gfx950 kernel."""
    name = "hip"
    device_resident = True     # handles accept / return CUDA tensors, so solvers may keep intermediates in HBM
    Krig = KrigHandle
    FFTGS = FFTGSHandle
    LUGS = LUGSHandle
    SGS = SGSHandle

    @staticmethod
    def cov_pairwise(vg, a, b=None):
        l = _lib.lib()
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.ndim == 1:
            a = a[:, None]
        bb = a if b is None else np.ascontiguousarray(b, dtype=np.float64).reshape(-1, a.shape[1])
        out = np.empty((a.shape[0], bb.shape[0]))
        v = _vg_struct(vg, a.shape[1])
        check(l.gss_cov_pairwise(C.byref(v), ptr(a), a.shape[0], None if b is None else ptr(bb), bb.shape[0],
                                 ptr(out), bb.shape[0], MEM_HOST, current_stream()))
        return out

    @staticmethod
    def knn_search(xdata, centers, k, radius=None, radii=None, distance=None):
        """Host arrays in -> host arrays out; CUDA tensors for both point sets keep the search in HBM."""
        l = _lib.lib()
        met, mpar = _lib.metric_spec(distance)
        dev = is_torch(xdata) and xdata.is_cuda
        if dev != (is_torch(centers) and centers.is_cuda):
            raise ValueError("xdata and centers must live in the same memory space")
        if dev:
            import torch
            x = _prep_in(xdata.reshape(xdata.shape[0], -1))
            c = _prep_in(centers.reshape(-1, x.shape[1]))
            m = c.shape[0]
            idx = torch.empty((m, k), dtype=torch.int32, device=x.device)
            cnt = torch.empty(m, dtype=torch.int32, device=x.device)
        else:
            x = np.ascontiguousarray(xdata, dtype=np.float64)
            if x.ndim == 1:
                x = x[:, None]
            c = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, x.shape[1])
            m = c.shape[0]
            idx = np.empty((m, k), dtype=np.int32)
            cnt = np.empty(m, dtype=np.int32)
        ir = None if radii is None else np.ascontiguousarray(1.0 / np.asarray(radii, dtype=np.float64))
        r = -1.0 if radius is None and radii is None else (1.0 if radii is not None else float(radius))
        check(l.gss_knn_search(ptr(x), x.shape[0], x.shape[1], ptr(c), m, int(k), r, ptr(ir), met, mpar, ptr(idx),
                               ptr(cnt), MEM_DEVICE if dev else MEM_HOST, current_stream()))
        return idx, cnt

    @staticmethod
    def _estimate(fn_name, extra, xdata, z, xdom, k, minneighbors, radius, radii, distance=None):
        """Host arrays in -> host arrays out; if `xdom` is a CUDA tensor everything stays in HBM.  `z` of shape (n,) is
        one value column; (nz, n) is nz columns that share the search and the weights (gss_*_predict_cols: the mean
        comes back as (nz, m), distance / variance / status per point)."""
        l = _lib.lib()
        dev = is_torch(xdom) and xdom.is_cuda
        if dev:
            import torch
            x = xdata if is_torch(xdata) else torch.as_tensor(np.asarray(xdata, dtype=np.float64), device="cuda")
            x = _prep_in(x.reshape(x.shape[0], -1))
            zz = _prep_in(z if is_torch(z) else torch.as_tensor(np.asarray(z, dtype=np.float64), device="cuda"))
            c = _prep_in(xdom.reshape(-1, x.shape[1]))
            m = c.shape[0]
            mshape = (m,) if zz.ndim == 1 else (zz.shape[0], m)
            mean = torch.empty(mshape, dtype=torch.float64, device="cuda")
            aux = torch.empty(m, dtype=torch.float64, device="cuda")
            st = torch.empty(m, dtype=torch.uint8, device="cuda")
        else:
            x = np.ascontiguousarray(xdata, dtype=np.float64)
            if x.ndim == 1:
                x = x[:, None]
            zz = np.ascontiguousarray(z, dtype=np.float64)
            c = np.ascontiguousarray(xdom, dtype=np.float64).reshape(-1, x.shape[1])
            m = c.shape[0]
            mean = np.empty((m,) if zz.ndim == 1 else (zz.shape[0], m))
            aux, st = np.empty(m), np.empty(m, dtype=np.uint8)
        if zz.ndim not in (1, 2) or zz.shape[-1] != x.shape[0]:
            raise ValueError("z must have shape (n,) or (nz, n)")
        ir = None if radii is None else np.ascontiguousarray(1.0 / np.asarray(radii, dtype=np.float64))
        r = -1.0 if radius is None and radii is None else (1.0 if radii is not None else float(radius))
        met, mpar = _lib.metric_spec(distance)
        if zz.ndim == 1:
            check(getattr(l, fn_name)(ptr(x), ptr(zz), x.shape[0], x.shape[1], ptr(c), m, int(k), int(minneighbors), r,
                                      ptr(ir), met, mpar, *extra, ptr(mean), ptr(aux), ptr(st),
                                      MEM_DEVICE if dev else MEM_HOST, current_stream()))
        else:
            check(getattr(l, fn_name + "_cols")(ptr(x), ptr(zz), x.shape[0], x.shape[1], int(zz.shape[0]), ptr(c), m, int(k),
                                                int(minneighbors), r, ptr(ir), met, mpar, *extra, ptr(mean), ptr(aux),
                                                ptr(st), MEM_DEVICE if dev else MEM_HOST, current_stream()))
        return mean, aux, st

    @staticmethod
    def idw(xdata, z, xdom, k, minneighbors=1, exponent=1.0, radius=None, radii=None, distance=None):
        """gss_idw_predict (idw.jl:111-142) -> mean, distance to the nearest sample, status."""
        return HipEngine._estimate("gss_idw_predict", (float(exponent),), xdata, z, xdom, k, minneighbors, radius,
                                   radii, distance)

    @staticmethod
    def lwr(xdata, z, xdom, k, minneighbors=1, weight=(0, 3.0, 2.0), radius=None, radii=None, distance=None):
        """gss_lwr_predict (lwr.jl:114-147); weight = (kind, a, p) -> mean, norm(r), status."""
        kind, a, p = weight
        return HipEngine._estimate("gss_lwr_predict", (int(kind), float(a), float(p)), xdata, z, xdom, k,
                                   minneighbors, radius, radii, distance)


    @staticmethod
    def lwr_callable(xdata, z, xdom, k, minneighbors, weightfun, radius=None, radii=None, distance=None):
        """LWR with an arbitrary `weightfun` callable (lwr.jl:58,136): the search runs on the device, delta = d / max d
        and w = weightfun(delta) are evaluated here on the host (the callable cannot cross the C-ABI), the normal
        equations and norm(r) on the device again (gss_lwr_predict_weights).  Host arrays."""
        x = np.ascontiguousarray(xdata, dtype=np.float64)
        if x.ndim == 1:
            x = x[:, None]
        zz = np.ascontiguousarray(z, dtype=np.float64)
        c = np.ascontiguousarray(xdom, dtype=np.float64).reshape(-1, x.shape[1])
        m = c.shape[0]
        idx, cnt = HipEngine.knn_search(x, c, k, radius, radii, distance)
        valid = np.arange(k)[None, :] < cnt[:, None]
        nb = np.where(valid, idx, 0)
        diff = x[nb] - c[:, None, :]                              # m x k x d
        name = "euclidean" if distance is None else (distance if isinstance(distance, str) else distance[0])
        if radii is not None:
            diff = diff / np.asarray(radii, dtype=np.float64)
        if name == "euclidean":
            d = np.sqrt(np.sum(diff * diff, axis=-1))
        elif name == "cityblock":
            d = np.sum(np.abs(diff), axis=-1)
        elif name == "chebyshev":
            d = np.max(np.abs(diff), axis=-1)
        else:                                                    # ("haversine", r): (longitude, latitude) in degrees
            D = np.pi / 180.0
            s1 = np.sin((c[:, None, 1] - x[nb][..., 1]) * 0.5 * D)
            s2 = np.sin((c[:, None, 0] - x[nb][..., 0]) * 0.5 * D)
            key = s1 * s1 + np.cos(x[nb][..., 1] * D) * np.cos(c[:, None, 1] * D) * (s2 * s2)
            d = 2.0 * float(distance[1]) * np.arcsin(np.minimum(np.sqrt(key), 1.0))
        d = np.where(valid, d, 0.0)
        with np.errstate(invalid="ignore", divide="ignore"):
            delta = d / d.max(axis=1, keepdims=True)             # lwr.jl:132
        # lwr.jl:136 `weightfun.(deltas)`: elementwise, on the neighbours that exist only; a scalar-only callable
        # (`lambda h: 1 - h if h < 1 else 0`) is mapped over them; a NaN weight (every distance zero: 0 / 0) stays NaN,
        # so that the point is reported singular as the reference's NaN estimate would show
        dv = delta[valid]
        try:
            wv = np.asarray(weightfun(dv), dtype=np.float64)
            if wv.shape != dv.shape:
                raise ValueError("weightfun did not map elementwise")
        except (TypeError, ValueError):
            wv = np.fromiter((float(weightfun(float(t))) for t in dv), dtype=np.float64, count=dv.size)
        w = np.zeros_like(delta)
        w[valid] = wv
        w = np.ascontiguousarray(w)
        mean, var, st = np.empty(m), np.empty(m), np.empty(m, dtype=np.uint8)
        check(_lib.lib().gss_lwr_predict_weights(ptr(x), ptr(zz), x.shape[0], x.shape[1], ptr(c), m, int(k),
                                                 int(minneighbors), ptr(np.ascontiguousarray(idx)),
                                                 ptr(np.ascontiguousarray(cnt)), ptr(w), ptr(mean), ptr(var), ptr(st),
                                                 MEM_HOST, current_stream()))
        return mean, var, st


def default_engine():
    return HipEngine
