"""ctypes binding of libgss_hip.so (the C-ABI in include/gss.h).

There is NO fallback: if the shared library is missing or no gfx950 device is visible the
calls raise.  `import torch` happens first on purpose -- PyTorch bundles its own
libamdhip64.so.7 / librocfft.so.0; loading it first makes libgss_hip.so bind to the same HIP
runtime, so device pointers and streams can be shared with torch tensors (plumbing only).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:  # plumbing: device memory, streams, torch.distributed
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for pure C-ABI use
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSS_LIB_PATH") or os.path.normpath(os.path.join(_HERE, "..", "lib", "libgss_hip.so"))

MEM_HOST, MEM_DEVICE = 0, 1
OK, ERR_INVALID, ERR_HIP, ERR_NOT_POSDEF, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_ALLOC = range(7)
KRIG_NO_FACTOR = 1
KRIG_ASYNC_FIT = 2
FFTGS_NO_SPECTRUM = 1
LUGS_NO_FACTOR = 1
LUGS_FACT_LU = 2
SGS_MASK_AFTER_SEARCH = 1
SGS_METRIC_SHIFT = 4


class GSSError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgss_hip error {code}: {msg}")
        self.code = code


class VariogramExtra(C.Structure):
    _fields_ = [("kind", C.c_int32), ("aniso", C.c_int32), ("sill", C.c_double), ("range", C.c_double),
                ("nu", C.c_double), ("inv_radii", C.c_double * 3)]


class Variogram(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim", C.c_int32), ("sill", C.c_double), ("nugget", C.c_double),
                ("range", C.c_double), ("nu", C.c_double), ("aniso", C.c_int32), ("reserved", C.c_int32),
                ("inv_radii", C.c_double * 3), ("nextra", C.c_int32), ("reserved2", C.c_int32),
                ("extra", VariogramExtra * 3)]


_p = C.c_void_p
_i32, _i64, _u64, _f64 = C.c_int32, C.c_int64, C.c_uint64, C.c_double
_VG = C.POINTER(Variogram)

# name -> argtypes; every symbol declared in include/gss.h must appear here (tests check it)
SIGNATURES = {
    "gss_version": [],
    "gss_device_count": [C.POINTER(_i32)],
    "gss_init": [_i32],
    "gss_shutdown": [],
    "gss_last_error": [C.c_char_p, _i32],
    "gss_synchronize": [_p],
    "gss_dev_to_host": [_p, _p, _i64, _p],
    "gss_trim_pool": [],
    "gss_stat": [C.c_char_p, C.POINTER(_i64)],
    "gss_comm_unique_id": [_p],
    "gss_comm_init": [_p, _i32, _i32],
    "gss_comm_info": [C.POINTER(_i32), C.POINTER(_i32)],
    "gss_comm_destroy": [],
    "gss_state_bcast": [_i32, _p, _i32, _p],
    "gss_state_ipc_export": [_i32, _p, _p],
    "gss_state_ipc_import": [_i32, _p, _p, _p],
    "gss_profile_enable": [_i32],
    "gss_profile_reset": [],
    "gss_profile_read": [C.c_char_p, C.POINTER(_f64), C.POINTER(_i64)],
    "gss_cov_pairwise": [_VG, _p, _i64, _p, _i64, _p, _i64, _i32, _p],
    "gss_knn_search": [_p, _i64, _i32, _p, _i64, _i32, _f64, _p, _i32, _f64, _p, _p, _i32, _p],
    "gss_krig_create": [C.POINTER(_p), _VG, _i32, _f64, _i32, _i32, _p, _p, _p, _i64, _i32, _p],
    "gss_krig_destroy": [_p],
    "gss_krig_info": [_p, C.POINTER(_i64), C.POINTER(_i32)],
    "gss_krig_factor_buffer": [_p, C.POINTER(_p), C.POINTER(_i64)],
    "gss_krig_adopt_factor": [_p],
    "gss_krig_predict_global": [_p, _p, _p, _i64, _p, _p, _p, _i32, _p],
    "gss_krig_set_block_support": [_p, _p, _i32, _p],
    "gss_krig_predict_knn": [_p, _p, _p, _i64, _i32, _i32, _f64, _p, _i32, _f64, _p, _p, _p, _p, _p, _i32, _p],
    "gss_krig_predict_global_batch": [_p, _p, _i64, _p, _i64, _p, _i32, _p],
    "gss_idw_predict": [_p, _p, _i64, _i32, _p, _i64, _i32, _i32, _f64, _p, _i32, _f64, _f64, _p, _p, _p, _i32, _p],
    "gss_lwr_predict": [_p, _p, _i64, _i32, _p, _i64, _i32, _i32, _f64, _p, _i32, _f64, _i32, _f64, _f64, _p, _p, _p, _i32,
                        _p],
    "gss_idw_predict_cols": [_p, _p, _i64, _i32, _i32, _p, _i64, _i32, _i32, _f64, _p, _i32, _f64, _f64, _p, _p, _p, _i32, _p],
    "gss_lwr_predict_cols": [_p, _p, _i64, _i32, _i32, _p, _i64, _i32, _i32, _f64, _p, _i32, _f64, _i32, _f64, _f64, _p, _p, _p,
                             _i32, _p],
    "gss_lwr_predict_weights": [_p, _p, _i64, _i32, _p, _i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p],
    "gss_sgs_create": [C.POINTER(_p), _VG, _f64, _p, _i64, _i32, _p, _p, _p, _i64, _i32, _i32, _f64, _p, _i32, _p],
    "gss_sgs_create_paths": [C.POINTER(_p), _VG, _f64, _p, _i64, _i32, _p, _i64, _i64, _p, _p, _i64, _i32, _i32, _f64, _p,
                             _i32, _p],
    "gss_sgs_destroy": [_p],
    "gss_sgs_weights": [_p, _p, _p, _p, _p, _i32, _p],
    "gss_sgs_realize": [_p, _u64, _i64, _i64, _p, _p, _i32, _p],
    "gss_fftgs_create": [C.POINTER(_p), _VG, _i32, _p, _p, _f64, _i32, _p],
    "gss_fftgs_destroy": [_p],
    "gss_fftgs_spectrum": [_p, _p, _i32, _p],
    "gss_fftgs_state_buffer": [_p, C.POINTER(_p), C.POINTER(_i64)],
    "gss_fftgs_adopt_state": [_p, _p],
    "gss_fftgs_realize": [_p, _u64, _i64, _i64, _p, _p, _i64, _p, _i32, _p],
    "gss_lugs_create": [C.POINTER(_p), _VG, _p, _i64, _p, _p, _i64, _f64, _i32, _p],
    "gss_lugs_destroy": [_p],
    "gss_lugs_info": [_p, C.POINTER(_i64), C.POINTER(_i64)],
    "gss_lugs_factor": [_p, _p, _p, _i32, _p],
    "gss_lugs_state_buffer": [_p, C.POINTER(_p), C.POINTER(_i64)],
    "gss_lugs_adopt_state": [_p],
    "gss_lugs_realize": [_p, _u64, _i64, _i64, _p, _f64, _p, _p, _p, _i32, _p],
    "gss_philox_uniform": [_u64, _i64, _i64, _p, _i32, _p],
    "gss_philox_normal": [_u64, _i64, _i64, _p, _i32, _p],
    "gss_dev_potrf": [_p, _i64, _i64, _p],
    "gss_dev_getrf_l": [_p, _i64, _i64, _p],
    "gss_dev_trtri": [_p, _i64, _i64, _p, _i64, _p],
    "gss_dev_potrf_inverse": [_p, _i64, _i64, _p, _i64, _p],
    "gss_dev_gemm": [_i64, _i64, _i64, _f64, _p, _i64, _i64, _p, _i64, _i64, _f64, _p, _i64, _i64, _i32, _p],
}

_lib = None
_initialised = False


def load():
    """dlopen the library (no device needed) and attach signatures."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GSSError(-1, f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               f"g.build()'` (hipcc, gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _i32
        _lib = lib
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().gss_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(code: int):
    if code != OK:
        raise GSSError(code, last_error())


def lib():
    """Library bound to the current device (one process per GPU)."""
    global _initialised
    l = load()
    if not _initialised:
        dev = 0
        if torch is not None and torch.cuda.is_available():
            dev = torch.cuda.current_device()
        check(l.gss_init(dev))
        _initialised = True
    return l


# ---- argument helpers -------------------------------------------------------------------
def is_torch(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def ptr(x):
    """void* of a numpy array / CUDA torch tensor / None."""
    if x is None:
        return None
    if is_torch(x):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)


def as_f64(x, shape=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def current_stream():
    if torch is not None and torch.cuda.is_available():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


_KINDS = {"gaussian": 0, "exponential": 1, "spherical": 2, "matern": 3, "cubic": 4, "pentaspherical": 5,
          "sinehole": 6, "power": 7}


def make_variogram(kind: str, dim: int, sill=1.0, nugget=0.0, range=1.0, nu=1.0, radii=None, extras=()) -> Variogram:
    """`extras`: further nested structures as (kind, contribution, range, nu, radii) tuples (at most 3)."""
    v = Variogram()
    v.kind = _KINDS[kind]
    v.dim = int(dim)
    v.sill, v.nugget, v.range, v.nu = float(sill), float(nugget), float(range), float(nu)
    v.aniso = 0
    for k in (0, 1, 2):
        v.inv_radii[k] = 1.0
    if radii is not None:
        if len(radii) != dim:
            raise ValueError(f"anisotropic ball has {len(radii)} radii but the domain is {dim}-D")
        v.aniso = 1
        if kind != "power":            # power: `range` carries the scaling factor
            v.range = 1.0
        for k, r in enumerate(radii):
            v.inv_radii[k] = 1.0 / float(r)
    if len(extras) > 3:
        raise ValueError("at most 4 nested structures are supported on the device")
    v.nextra = len(extras)
    for e, (ekind, esill, erange, enu, eradii) in enumerate(extras):
        x = v.extra[e]
        x.kind, x.aniso, x.sill, x.range, x.nu = _KINDS[ekind], 0, float(esill), float(erange), float(enu)
        for k in (0, 1, 2):
            x.inv_radii[k] = 1.0
        if eradii is not None:
            if len(eradii) != dim:
                raise ValueError(f"anisotropic ball has {len(eradii)} radii but the domain is {dim}-D")
            x.aniso, x.range = 1, 1.0
            for k, r in enumerate(eradii):
                x.inv_radii[k] = 1.0 / float(r)
    return v


METRICS = {"euclidean": 0, "cityblock": 1, "chebyshev": 2, "haversine": 3}


def metric_spec(distance):
    """Solver parameter `distance` -> (GSS_METRIC_*, parameter).  Accepts None / "euclidean" / "cityblock" /
    "chebyshev" / ("haversine", radius) (the Distances.jl objects Euclidean(), Cityblock(), Chebyshev(), Haversine(r))."""
    if distance is None:
        return 0, 0.0
    if isinstance(distance, str):
        name, par = distance.lower(), 0.0
    else:
        name, par = str(distance[0]).lower(), float(distance[1])
    if name not in METRICS:
        raise NotImplementedError(f"search distance {distance!r}: euclidean, cityblock, chebyshev or ('haversine', r)")
    if name == "haversine" and not par > 0.0:
        raise ValueError("('haversine', radius) needs a positive radius")
    return METRICS[name], par


STATE_KRIG, STATE_FFTGS, STATE_LUGS = 0, 1, 2
COMM_ID_BYTES, IPC_TOKEN_BYTES = 128, 96


def comm_unique_id() -> bytes:
    """Rank 0: the 128 bytes every rank passes to `comm_init` (gss.h, gss_comm_unique_id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(lib().gss_comm_unique_id(C.cast(buf, C.c_void_p)))
    return buf.raw


def comm_init(uid: bytes, rank: int, nranks: int):
    if len(uid) != COMM_ID_BYTES:
        raise ValueError("unique id must have 128 bytes")
    buf = C.create_string_buffer(uid, COMM_ID_BYTES)
    check(lib().gss_comm_init(C.cast(buf, C.c_void_p), int(rank), int(nranks)))


def comm_info():
    r, n = C.c_int32(), C.c_int32()
    check(load().gss_comm_info(C.byref(r), C.byref(n)))
    return r.value, n.value


def comm_destroy():
    check(load().gss_comm_destroy())


def state_bcast(kind: int, handle, root: int = 0):
    check(lib().gss_state_bcast(int(kind), handle, int(root), current_stream()))


def state_ipc_export(kind: int, handle) -> bytes:
    buf = C.create_string_buffer(IPC_TOKEN_BYTES)
    check(lib().gss_state_ipc_export(int(kind), handle, C.cast(buf, C.c_void_p)))
    return buf.raw


def state_ipc_import(kind: int, handle, token: bytes):
    if len(token) != IPC_TOKEN_BYTES:
        raise ValueError("IPC token must have %d bytes" % IPC_TOKEN_BYTES)
    buf = C.create_string_buffer(token, IPC_TOKEN_BYTES)
    check(lib().gss_state_ipc_import(int(kind), handle, C.cast(buf, C.c_void_p), current_stream()))


def stat(name: str) -> int:
    """A library counter: "pool_bytes", "out_ring_bytes", "out_chunks", "panel_giveups", "ipc_route" (gss.h, gss_stat)."""
    v = C.c_int64()
    check(load().gss_stat(name.encode(), C.byref(v)))
    return v.value


def trim_pool():
    """Give the library's cached device blocks back to the driver (gss.h, gss_trim_pool)."""
    check(load().gss_trim_pool())


def profile_enable(on: bool = True):
    check(load().gss_profile_enable(1 if on else 0))


def profile_reset():
    check(load().gss_profile_reset())


def profile_read(name: str):
    """(total milliseconds, launches) of the named hot kernel since the last reset."""
    ms, n = C.c_double(), C.c_int64()
    check(load().gss_profile_read(name.encode(), C.byref(ms), C.byref(n)))
    return ms.value, n.value
