"""The C-ABI library loads without a GPU and exports exactly what include/gss.h declares."""
import ctypes
import os
import re

import pytest

from gss import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gss.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint32_t\s+(gss_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gss.h but not exported"


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_buffer():
    lib = _lib.load()
    assert lib.gss_version() == 100
    buf = ctypes.create_string_buffer(64)
    assert lib.gss_last_error(buf, 64) == 0


def test_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present")
    lib = _lib.load()
    assert lib.gss_init(0) == _lib.ERR_NO_DEVICE
    assert "no CPU fallback" in _lib.last_error()


def test_invalid_arguments_do_not_need_a_device():
    lib = _lib.load()
    h = ctypes.c_void_p()
    v = _lib.make_variogram("gaussian", 2)
    # krig.jl:100-102: no non-missing sample
    code = lib.gss_krig_create(ctypes.byref(h), ctypes.byref(v), 1, 0.0, 0, 0, ctypes.c_void_p(8), ctypes.c_void_p(8),
                               None, 0, 0, None)
    assert code == _lib.ERR_INVALID and "missing" in _lib.last_error()
    bad = _lib.make_variogram("matern", 2, nu=80.0)          # orders in (0, 50] only
    code = lib.gss_krig_create(ctypes.byref(h), ctypes.byref(bad), 1, 0.0, 0, 0, ctypes.c_void_p(8),
                               ctypes.c_void_p(8), None, 4, 0, None)
    assert code == _lib.ERR_INVALID and "must lie" in _lib.last_error()


def test_no_dpp_read_follows_a_write_too_closely():
    """csrc/tile16.h issues v_fmac_f64_dpp from inline assembly, which the compiler's hazard recogniser cannot see;
    tools/check_dpp_hazards.py disassembles the gfx950 code objects of the built library and checks the two wait
    states between a VALU write and a DPP read of the same register."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_dpp_hazards.py"), _lib.LIB_PATH],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert int(r.stdout.split()[0]) > 1000, r.stdout        # the tile factorisations are in the library


def test_register_budgets_of_the_occupancy_sensitive_kernels():
    """A kernel's register count is the maximum over all its paths, so one added instantiation can halve the occupancy
    of every launch without any test noticing (round 4: a second item per thread in the radix-7 pass took the strided
    FFTGS passes from 116 to 146 registers, 300^3 0.57 -> 0.83 ms).  The budgets the measured numbers rest on, read from
    the metadata of the built library (tools/kernel_resources.py)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not available")
    budgets = {"gen_axis_kernel": 128, "gen_long_inner_kernel": 128, "gen_long_outer_kernel": 128,   # 2 x 512 threads / CU
               "gen_x_inv_kernel": 102, "gen_x_fwd_kernelILi0": 102,                                   # 5 x 256 threads / CU
               "krig_local_mfma_kernel": 170, "krig_quadform_kernel": 256}                             # 3 and 2 waves per SIMD
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), _lib.LIB_PATH],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    seen = {k: 0 for k in budgets}
    for line in r.stdout.splitlines():
        m = re.match(r"(\S+)\s+vgpr\s+(\d+)", line)
        if not m:
            continue
        for k, cap in budgets.items():
            if k in m.group(1):
                seen[k] += 1
                assert int(m.group(2)) <= cap, line
    assert all(seen.values()), seen
