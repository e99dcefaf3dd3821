"""ctypes access to oracle/libkrig_oracle.so (the C restatement; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

from .variogram import effective_nugget

_HERE = os.path.dirname(os.path.abspath(__file__))
_KINDS = {"gaussian": 0, "exponential": 1, "spherical": 2, "matern": 3}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _lib():
    path = os.path.join(_HERE, "libkrig_oracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.krig_oracle_global.restype = C.c_int
    lib.krig_oracle_global.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                       C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_int]
    return lib


def krig_global(vg, variant, x, z, x0, mean=0.0, nthreads=1):
    """variant 0 = simple, 1 = ordinary; vg is an oracle.variogram.Variogram (isotropic)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    mu = np.empty(x0.shape[0])
    var = np.empty(x0.shape[0])
    rc = _lib().krig_oracle_global(_KINDS[vg.kind], x.shape[1], vg.sill, effective_nugget(vg), vg.range, vg.nu, variant,
                                   float(mean), x.ctypes.data, z.ctypes.data, x.shape[0], x0.ctypes.data,
                                   x0.shape[0], mu.ctypes.data, var.ctypes.data, int(nthreads))
    if rc:
        raise RuntimeError(f"krig_oracle_global failed ({rc})")
    return mu, var
