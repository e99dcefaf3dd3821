"""Host sanitizer builds: `make asan` in csrc/ (host side of libgss_hip.so under ASan + UBSan,
device code untouched) and in oracle/ (the C restatement).  Each library is driven in a child process with the
sanitizer runtime preloaded; any report makes the child exit non-zero.  No GPU is used: the product library is
exercised on the paths that run before the first device call (argument validation, host-side index / path /
location checks, error unwinding with half-built handles)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "geostatssolvers.jl_amd", "csrc")
CLANG_RT = None
for _d in sorted(__import__("glob").glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux")):
    if os.path.exists(os.path.join(_d, "libclang_rt.asan-x86_64.so")):
        CLANG_RT = os.path.join(_d, "libclang_rt.asan-x86_64.so")


def _run(code, preload, extra_env=None):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")]))
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env,
                          timeout=600)


def _clean(r):
    bad = [l for l in (r.stdout + r.stderr).splitlines() if "AddressSanitizer" in l or "runtime error" in l]
    assert r.returncode == 0 and not bad, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.timeout(900)
def test_product_library_host_side_under_asan_ubsan():
    if CLANG_RT is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("ROCm clang sanitizer runtime not available")
    subprocess.check_call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", CSRC, "asan"])
    code = """
        import ctypes as C, numpy as np
        from gss import _lib
        lib = C.CDLL(_lib.LIB_PATH.replace("libgss_hip.so", "libgss_hip_asan.so"))
        for name, args in _lib.SIGNATURES.items():
            f = getattr(lib, name); f.argtypes = args; f.restype = C.c_int32
        def err():
            b = C.create_string_buffer(512); lib.gss_last_error(b, 512); return b.value.decode()
        assert lib.gss_version() == 100
        n = C.c_int32(-1); assert lib.gss_device_count(C.byref(n)) == 0
        have_gpu = n.value > 0
        h = C.c_void_p()
        P = lambda a: C.c_void_p(a.ctypes.data)
        # variogram validation (make_vgdev): every rejection path
        for kw in (dict(kind="matern", nu=80.0), dict(kind="gaussian", sill=-1.0), dict(kind="gaussian", range=0.0),
                   dict(kind="power", nu=2.5, sill=3.0)):
            v = _lib.make_variogram(kw.pop("kind"), 2, **kw)
            x = np.zeros((4, 2)); z = np.zeros(4)
            assert lib.gss_krig_create(C.byref(h), C.byref(v), 1, 0.0, 0, 0, P(x), P(z), None, 4, 0, None) == 1, err()
        v = _lib.make_variogram("spherical", 2, range=5.0)
        assert lib.gss_krig_create(C.byref(h), C.byref(v), 1, 0.0, 0, 0, C.c_void_p(8), C.c_void_p(8), None, 0, 0,
                                   None) == 1 and "missing" in err()                    # krig.jl:100-102
        assert lib.gss_krig_create(C.byref(h), C.byref(v), 1, 0.0, 0, 0, None, None, None, 4, 0, None) == 1
        # LUGS: host-side location checks run before any device call; the half-built handle unwinds
        cent = np.stack(np.meshgrid(np.arange(6.0), np.arange(5.0)), -1).reshape(-1, 2).copy()
        for dl in ([3, 3, 7], [9, 2], [0, 40], [-1, 4]):
            d = np.asarray(dl, dtype=np.int64); z1 = np.zeros(len(dl))
            assert lib.gss_lugs_create(C.byref(h), C.byref(v), P(cent), 30, P(d), P(z1), len(dl), 0.0, 0, None) == 1, dl
        # SGS: path permutation / data location checks
        for path, dl in (([0, 1, 1] + list(range(3, 30)), [2]), (list(range(30)), [5, 5]), (list(range(29)) + [77], [1])):
            pa = np.asarray(path, dtype=np.int64); d = np.asarray(dl, dtype=np.int64); zd = np.zeros(len(dl))
            assert lib.gss_sgs_create(C.byref(h), C.byref(v), 0.0, P(cent), 30, 2, P(pa), P(d), P(zd), len(dl), 5, 1,
                                      -1.0, None, 0, None) == 1, (path[:4], dl)
        # FFTGS: dimension checks
        dims = (C.c_int64 * 3)(8, 1, 1); sp = (C.c_double * 3)(1, 1, 1)
        assert lib.gss_fftgs_create(C.byref(h), C.byref(v), 1, dims, sp, 0.0, 0, None) == 1      # 2-D model, 1-D grid
        v1 = _lib.make_variogram("spherical", 1, range=5.0)
        dims1 = (C.c_int64 * 3)(1, 1, 1)
        assert lib.gss_fftgs_create(C.byref(h), C.byref(v1), 1, dims1, sp, 0.0, 0, None) == 1    # one cell
        # search arguments
        x = np.zeros((10, 2)); c = np.zeros((3, 2)); idx = np.zeros((3, 4), dtype=np.int32)
        assert lib.gss_knn_search(P(x), 10, 2, P(c), 3, 11, -1.0, None, 0, 0.0, P(idx), None, 0, None) == 1   # k > n
        assert lib.gss_knn_search(P(x), 10, 2, P(c), 3, 4, 2.0, None, 1, 0.0, P(idx), None, 0, None) == 1     # ball + metric
        assert lib.gss_knn_search(P(x), 10, 2, P(c), 3, 4, -1.0, None, 3, 0.0, P(idx), None, 0, None) == 1    # haversine r
        if not have_gpu:
            assert lib.gss_init(0) == 5 and "no CPU fallback" in err()
            # without a device the first allocation fails: creation must return an error and free what it built
            dims3 = (C.c_int64 * 3)(16, 16, 16)
            v3 = _lib.make_variogram("spherical", 3, range=5.0)
            assert lib.gss_fftgs_create(C.byref(h), C.byref(v3), 3, dims3, sp, 0.0, 0, None) != 0
            d = np.asarray([2, 9], dtype=np.int64); z1 = np.zeros(2)
            assert lib.gss_lugs_create(C.byref(h), C.byref(v), P(cent), 30, P(d), P(z1), 2, 0.0, 0, None) != 0
            x = np.random.default_rng(0).uniform(size=(50, 2)); z = np.zeros(50)
            assert lib.gss_krig_create(C.byref(h), C.byref(v), 1, 0.0, 0, 0, P(x), P(z), None, 50, 0, None) != 0
        # two host threads at once (ctypes releases the GIL inside the calls): every export serialises on the library's
        # lock (gss.h, host threads), the error text is per thread, the counters and the profile registry stay coherent
        import threading
        bad = []
        def hammer(tid):
            hh = C.c_void_p(); val = C.c_int64(); ms = C.c_double(); nl = C.c_int64()
            vb = _lib.make_variogram("matern", 2, nu=80.0 + tid)
            xx = np.zeros((4, 2)); zz = np.zeros(4)
            for it in range(300):
                if lib.gss_krig_create(C.byref(hh), C.byref(vb), 1, 0.0, 0, 0, P(xx), P(zz), None, 4, 0, None) != 1:
                    bad.append("status")
                if f"nu={80 + tid}" not in err():
                    bad.append(err())
                if lib.gss_stat(b"pool_bytes", C.byref(val)) != 0 or val.value != 0:
                    bad.append("stat")
                lib.gss_profile_enable(it & 1); lib.gss_profile_read(b"krig_rhs", C.byref(ms), C.byref(nl))
                if it % 50 == 0:
                    lib.gss_profile_reset()
                if not have_gpu and lib.gss_trim_pool() == 0:
                    bad.append("trim without a device")
        ts = [threading.Thread(target=hammer, args=(t,)) for t in range(2)]
        [t.start() for t in ts]; [t.join() for t in ts]
        assert not bad, bad[:3]
        lib.gss_profile_enable(0); lib.gss_profile_reset(); lib.gss_shutdown()
        print("asan-ok")
    """
    try:
        r = _run(code, CLANG_RT, {"LD_LIBRARY_PATH": os.path.dirname(CLANG_RT) + os.pathsep + "/opt/rocm/lib" + os.pathsep +
                                  os.environ.get("LD_LIBRARY_PATH", "")})
    finally:   # 23 MB of instrumented objects: not worth shipping to the GPU box with every gpurun snapshot
        subprocess.call(["make", "-s", "-C", CSRC, "clean-asan"])
    _clean(r)
    assert "asan-ok" in r.stdout


def test_oracle_c_restatement_under_asan_ubsan():
    gcc_asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(gcc_asan) or not os.path.exists(gcc_asan):
        pytest.skip("gcc libasan not available")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    code = """
        import ctypes as C, os, numpy as np
        from oracle import cbind, kriging as K
        from oracle.variogram import Variogram
        real = cbind._lib
        def asan_lib():
            lib = C.CDLL(os.path.join(os.path.dirname(cbind.__file__), "libkrig_oracle_asan.so"))
            ref = real()
            lib.krig_oracle_global.restype = ref.krig_oracle_global.restype
            lib.krig_oracle_global.argtypes = ref.krig_oracle_global.argtypes
            return lib
        cbind._lib = asan_lib
        rng = np.random.default_rng(3)
        for kind, kw in (("gaussian", dict(range=20.0, nugget=1e-3)), ("exponential", dict(range=15.0)),
                         ("spherical", dict(range=30.0, sill=2.0)), ("matern", dict(range=25.0, nu=1.5))):
            for dim in (1, 2, 3):
                x = rng.uniform(0, 50, (40, dim)); z = rng.normal(size=40); x0 = rng.uniform(0, 50, (33, dim))
                vg = Variogram(kind, **kw)
                for variant, mean in ((K.OK, 0.0), (K.SK, 0.4)):
                    for nt in (1, 3):
                        mu, var = cbind.krig_global(vg, variant, x, z, x0, mean=mean, nthreads=nt)
                        rmu, rvar = K.exactsolve(variant, vg, x, z, x0, mean=mean)
                        assert np.max(np.abs(mu - rmu)) < 1e-6 and np.max(np.abs(var - rvar)) < 1e-6, (kind, dim, variant)
        print("asan-ok")
    """
    try:
        r = _run(code, gcc_asan)
    finally:
        try:
            os.remove(os.path.join(ROOT, "oracle", "libkrig_oracle_asan.so"))
        except OSError:
            pass
    _clean(r)
    assert "asan-ok" in r.stdout
