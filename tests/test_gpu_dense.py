"""FP64 MFMA toolkit vs numpy/LAPACK (FP64).  Tolerances: GEMM 1e-12 relative to |A||B|,
Cholesky / inverse 1e-10 relative on well-conditioned SPD matrices."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def _gemm(M, N, K, A, sa, B, sb, D, sd, alpha=1.0, beta=0.0, lower=0):
    from gss import _lib
    l = _lib.lib()
    _lib.check(l.gss_dev_gemm(M, N, K, alpha, _lib.ptr(A), sa[0], sa[1], _lib.ptr(B), sb[0], sb[1], beta,
                              _lib.ptr(D), sd[0], sd[1], lower, _lib.current_stream()))


@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (128, 128, 16), (130, 67, 33), (257, 300, 129), (5, 1000, 1000)])
def test_gemm_all_layouts_asymmetric(M, N, K):
    import torch
    rng = np.random.default_rng(M * 7 + N)
    A = rng.normal(size=(M, K))
    B = rng.normal(size=(K, N))
    ref = A @ B
    scale = np.abs(A) @ np.abs(B)
    for a_cm in (False, True):
        for b_cm in (False, True):
            for d_cm in (False, True):
                dA = _t(A.T if a_cm else A)          # column-major image when a_cm
                dB = _t(B.T if b_cm else B)
                dD = torch.zeros((N, M) if d_cm else (M, N), dtype=torch.float64, device="cuda")
                sa = (1, M) if a_cm else (K, 1)
                sb = (1, K) if b_cm else (N, 1)
                sd = (1, M) if d_cm else (N, 1)
                _gemm(M, N, K, dA, sa, dB, sb, dD, sd)
                got = dD.cpu().numpy()
                got = got.T if d_cm else got
                assert np.max(np.abs(got - ref) / scale) < 1e-14 * K


def test_gemm_identity_catches_transposed_output():
    import torch
    n = 64
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)     # asymmetric
    dI, dB = _t(np.eye(n)), _t(B)
    dD = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    _gemm(n, n, n, dI, (n, 1), dB, (n, 1), dD, (n, 1))
    assert np.array_equal(dD.cpu().numpy(), B)


def test_gemm_alpha_beta_and_lower_only():
    import torch
    rng = np.random.default_rng(5)
    n, k = 300, 70
    A = rng.normal(size=(n, k))
    C0 = rng.normal(size=(n, n))
    dA, dC = _t(A.T), _t(C0.T.copy())                          # column-major
    _gemm(n, n, k, dA, (1, n), dA, (n, 1), dC, (1, n), alpha=-1.0, beta=1.0, lower=1)
    got = dC.cpu().numpy().T
    ref = C0 - A @ A.T
    il = np.tril_indices(n)
    assert np.allclose(got[il], ref[il], atol=1e-11)
    # tiles strictly above the diagonal blocks are untouched
    assert np.array_equal(got[:128, 128:], C0[:128, 128:])


@pytest.mark.parametrize("n", [1, 7, 64, 65, 200, 1000, 1537])
def test_potrf_and_trtri(n):
    import torch
    from gss import _lib
    l = _lib.lib()
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, n + 8))
    A = G @ G.T / n + np.eye(n)
    lda = n + 3
    buf = np.zeros((n, lda))
    buf[:, :n] = A                                            # row r of buf = column r (A symmetric)
    dA = _t(buf)
    _lib.check(l.gss_dev_potrf(_lib.ptr(dA), n, lda, _lib.current_stream()))
    L = np.tril(dA.cpu().numpy()[:, :n].T)
    Lref = np.linalg.cholesky(A)
    assert np.max(np.abs(L - Lref)) < 1e-10 * np.max(np.abs(Lref))
    dW = torch.full((n, lda), 7.0, dtype=torch.float64, device="cuda")
    _lib.check(l.gss_dev_trtri(_lib.ptr(dA), n, lda, _lib.ptr(dW), lda, _lib.current_stream()))
    W = dW.cpu().numpy()[:, :n].T
    assert np.array_equal(np.triu(W, 1), np.zeros((n, n)))
    assert np.max(np.abs(W @ Lref - np.eye(n))) < 1e-9


@pytest.mark.parametrize("n", [5, 128, 129, 256, 257, 300, 384, 385, 1000, 1024, 1025, 1600, 2048])
def test_potrf_inverse_single_launch_panel(n):
    """Factor and inverse together (the kriging fit, the LUGS blocks): blocks of 257 ... 1 024 rows run as one
    cooperative launch (potrf_inv_panel_kernel, grid barriers between its phases), larger ones split down to it,
    smaller ones are the 64 / 128 leaves.  Partial last blocks, leading dimensions that are not multiples of
    anything, and what lies outside the n x n blocks is left alone."""
    import torch
    from gss import _lib
    l = _lib.lib()
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, n + 8))
    A = G @ G.T / n + np.eye(n)
    lda, ldw = n + 3, n + 5
    buf = np.full((n + 2, lda), 9.0)
    buf[:n, :n] = A
    dA = _t(buf)
    dW = torch.full((n + 2, ldw), 7.0, dtype=torch.float64, device="cuda")
    _lib.check(l.gss_dev_potrf_inverse(_lib.ptr(dA), n, lda, _lib.ptr(dW), ldw, _lib.current_stream()))
    gotA, gotW = dA.cpu().numpy(), dW.cpu().numpy()
    L = np.tril(gotA[:n, :n].T)
    Lref = np.linalg.cholesky(A)
    assert np.max(np.abs(L - Lref)) < 1e-10 * np.max(np.abs(Lref))
    W = gotW[:n, :n].T
    assert np.array_equal(np.triu(W, 1), np.zeros((n, n)))
    assert np.max(np.abs(W @ Lref - np.eye(n))) < 1e-9
    assert np.all(gotA[n:, :] == 9.0) and np.all(gotA[:, n:] == 9.0)
    assert np.all(gotW[n:, :] == 7.0) and np.all(gotW[:, n:] == 7.0)


def test_potrf_inverse_launch_per_block_path_stays_alive():
    """GSS_PANEL_MAX=0 switches the single-launch kernel off (the path a process falls back to when a launch reports
    that its workgroups did not all arrive): same answers from the launch-per-block recursion.  The switch is read once
    per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "from gss import _lib\n"
        "l = _lib.lib()\n"
        "for n in (700, 1000, 1537):\n"
        "    rng = np.random.default_rng(n)\n"
        "    G = rng.normal(size=(n, n + 8)); A = G @ G.T / n + np.eye(n)\n"
        "    dA = torch.from_numpy(A.copy()).cuda(); dW = torch.zeros((n, n), dtype=torch.float64, device='cuda')\n"
        "    _lib.check(l.gss_dev_potrf_inverse(_lib.ptr(dA), n, n, _lib.ptr(dW), n, _lib.current_stream()))\n"
        "    L = np.linalg.cholesky(A); W = dW.cpu().numpy().T\n"
        "    assert np.max(np.abs(np.tril(dA.cpu().numpy().T) - L)) < 1e-10 * np.max(np.abs(L))\n"
        "    assert np.max(np.abs(W @ L - np.eye(n))) < 1e-9\n"
        "print('RECURSION OK')\n"
    ) % (root, os.path.join(root, "geostatssolvers.jl_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GSS_PANEL_MAX="0"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "RECURSION OK" in r.stdout, r.stderr[-2000:]


def test_recovery_when_the_single_launch_kernel_gives_up():
    """A launch whose workgroups do not all arrive within the bounded spin reports -1 (GSS_PANEL_FAIL=1 fakes exactly
    that for the first launch of the process): the kriging fit retries on the launch-per-block path by itself and the
    results are the usual ones; gss_lugs_create runs its preprocess a second time by itself."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, torch, gss\n"
        "from gss import _lib\n"
        "from gss.engine import KrigHandle, LUGSHandle, OK\n"
        "from oracle import kriging as K\n"
        "from oracle.variogram import Variogram\n"
        "rng = np.random.default_rng(0)\n"
        "x = rng.uniform(0, 100, (600, 3)); z = rng.normal(size=600); x0 = rng.uniform(0, 100, (50, 3))\n"
        "h = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5), OK, x, z)\n"
        "mu, var, st = h.predict_global(x0)\n"
        "ref = K.predict(K.fit(K.OK, Variogram('matern', range=30.0, nu=1.5), x, z), x0)\n"
        "assert np.max(np.abs(mu - ref[0])) < 1e-9 and np.max(np.abs(var - ref[1])) < 1e-9\n"
        "print('KRIG RECOVERED')\n"
    ) % (root, os.path.join(root, "geostatssolvers.jl_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GSS_PANEL_FAIL="1"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "KRIG RECOVERED" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    code2 = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, gss\n"
        "from gss import _lib\n"
        "from gss.engine import LUGSHandle\n"
        "cent = gss.CartesianGrid(40, 30).centroids()\n"
        "dl = np.arange(0, 1200, 3)[:300]; z1 = np.random.default_rng(1).normal(size=300)\n"
        "h = LUGSHandle(gss.SphericalVariogram(range=8.0, nugget=0.05), cent, dl, z1)\n"
        "r = h.realize(1, 0, 2)\n"
        "assert np.array_equal(r[0][:, dl] if isinstance(r, tuple) else r[:, dl], np.tile(z1, (2, 1)))\n"
        "print('LUGS RECOVERED')\n"
    ) % (root, os.path.join(root, "geostatssolvers.jl_amd"))
    r = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, GSS_PANEL_FAIL="1"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "LUGS RECOVERED" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_potrf_inverse_panel_reports_the_pivot():
    from gss import _lib
    l = _lib.lib()
    n = 700
    A = np.eye(n)
    A[533, 533] = -2.0
    dA, dW = _t(A), _t(np.zeros((n, n)))
    code = l.gss_dev_potrf_inverse(_lib.ptr(dA), n, n, _lib.ptr(dW), n, _lib.current_stream())
    assert code == _lib.ERR_NOT_POSDEF and "pivot at row 533" in _lib.last_error()


def test_potrf_reports_indefinite():
    from gss import _lib
    l = _lib.lib()
    A = np.eye(100)
    A[70, 70] = -1.0
    dA = _t(A)
    code = l.gss_dev_potrf(_lib.ptr(dA), 100, 100, _lib.current_stream())
    assert code == _lib.ERR_NOT_POSDEF and "pivot at row 70" in _lib.last_error()


@pytest.mark.parametrize("n", [1, 7, 32, 33, 100, 257, 1500, 2500, 4133])
def test_getrf_unit_lower_matches_lapack(n):
    """gss_dev_getrf_l: `lu(A).L` with LAPACK's pivoting rule (first row of maximal |a|), on matrices that do pivot.
    Beyond 2 048 rows the 32-column panels run on a grid of workgroups with one barrier per column (lu.hip); 4 133 is
    ragged against the 128-column outer panels, the 32-column panels and the 1 024-row workgroups."""
    import scipy.linalg as sla
    import torch
    from gss import _lib
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)) + 0.5 * np.eye(n)
    dA = _t(A.T)                                     # column-major image
    _lib.check(_lib.lib().gss_dev_getrf_l(_lib.ptr(dA), n, n, _lib.current_stream()))
    torch.cuda.synchronize()
    L = dA.cpu().numpy().T
    P, Lref, U = sla.lu(A)
    assert not np.array_equal(P, np.eye(n)) or n < 3
    assert np.max(np.abs(L - Lref)) < 1e-10 * max(1.0, np.max(np.abs(np.linalg.inv(U))) * 1e-3)


def test_lu_grid_panel_gives_up_cleanly():
    """GSS_LU_PANEL_FAIL=1 gives the first grid panel of the process no patience at its barriers, so its workgroups
    really leave through the give-up path of `getrf_panel_grid_kernel` (lu.hip): the verdict must be "gave up" (-1), never
    a spurious "exactly singular matrix" from candidate records that nobody wrote, the later panels of the call must not
    wait again, and the message must say that the matrix was overwritten.  The second call of the process runs on the
    single-workgroup panels and matches LAPACK; `gss_lugs_create(factorization = lu)` retries by itself."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, time\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, torch, scipy.linalg as sla, gss\n"
        "from gss import _lib\n"
        "n = 2500\n"
        "A = np.random.default_rng(3).normal(size=(n, n)) + 0.5 * np.eye(n)\n"
        "dA = torch.as_tensor(np.ascontiguousarray(A.T), device='cuda')\n"
        "t = time.time()\n"
        "rc = _lib.lib().gss_dev_getrf_l(_lib.ptr(dA), n, n, _lib.current_stream())\n"
        "assert rc == _lib.ERR_HIP, rc\n"
        "msg = _lib.last_error()\n"
        "assert 'gave up' in msg and 'restore the input' in msg and 'singular' not in msg, msg\n"
        "assert time.time() - t < 20.0\n"
        "dA = torch.as_tensor(np.ascontiguousarray(A.T), device='cuda')\n"
        "_lib.check(_lib.lib().gss_dev_getrf_l(_lib.ptr(dA), n, n, _lib.current_stream()))\n"
        "torch.cuda.synchronize()\n"
        "P, Lref, U = sla.lu(A)\n"
        "assert np.max(np.abs(dA.cpu().numpy().T - Lref)) < 1e-10\n"
        "print('GETRF RECOVERED')\n"
    ) % (root, os.path.join(root, "geostatssolvers.jl_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GSS_LU_PANEL_FAIL="1"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "GETRF RECOVERED" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    code2 = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, gss\n"
        "from gss.engine import LUGSHandle\n"
        "from oracle import lugs as OL\n"
        "from oracle.variogram import Variogram\n"
        "cent = gss.CartesianGrid(60, 50).centroids()\n"
        "h = LUGSHandle(gss.SphericalVariogram(range=8.0, nugget=0.05), cent, np.zeros(0, np.int64), np.zeros(0),\n"
        "               factorization='lu')\n"
        "L22, d2 = h.factor()\n"
        "pre = OL.preprocess(Variogram('spherical', range=8.0, nugget=0.05), cent, factorization='lu')\n"
        "assert np.max(np.abs(L22 - pre.L22)) < 1e-9\n"
        "print('LUGS LU RECOVERED')\n"
    ) % (root, os.path.join(root, "geostatssolvers.jl_amd"))
    r = subprocess.run([sys.executable, "-c", code2], env=dict(os.environ, GSS_LU_PANEL_FAIL="1"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "LUGS LU RECOVERED" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_single_launch_panel_beside_a_callers_own_kernels():
    """The single-launch factor-and-inverse kernel needs its 64 workgroups resident together, a whole CU each.  The
    library's own launch order guarantees that; a CALLER's kernels on another stream (here: a chain of large FP64
    matrix products from torch, LDS-heavy workgroups that keep every CU busy) can delay their arrival.  Whatever happens
    -- the panel squeezes in, or its bounded wait ends and the fit repeats on the launch-per-block path -- the
    estimates must be the usual ones and the call must return; the child prints what happened and what it cost
    (recorded in profiles/)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, time\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, torch, gss\n"
        "from gss import _lib\n"
        "from gss.engine import KrigHandle, OK\n"
        "from oracle import kriging as K\n"
        "from oracle.variogram import Variogram\n"
        "rng = np.random.default_rng(0)\n"
        "x = rng.uniform(0, 100, (1000, 3)); z = rng.normal(size=1000); x0 = rng.uniform(0, 100, (400, 3))\n"
        "ref = K.predict(K.fit(K.OK, Variogram('matern', range=30.0, nu=1.5), x, z), x0)\n"
        "vg = gss.MaternVariogram(range=30.0, order=1.5)\n"
        "def fit_predict():\n"
        "    t0 = time.perf_counter()\n"
        "    h = KrigHandle(vg, OK, x, z)\n"
        "    mu, var, st = h.predict_global(x0)\n"
        "    torch.cuda.synchronize()\n"
        "    dt = time.perf_counter() - t0\n"
        "    h.close()\n"
        "    assert np.max(np.abs(mu - ref[0])) < 1e-9 and np.max(np.abs(var - ref[1])) < 1e-9\n"
        "    return dt\n"
        "fit_predict()\n"
        "alone = min(fit_predict() for _ in range(3))\n"
        "a = torch.randn(6144, 6144, dtype=torch.float64, device='cuda')\n"
        "side = torch.cuda.Stream()\n"
        "beside = []\n"
        "for trial in range(6):\n"
        "    with torch.cuda.stream(side):\n"
        "        for _ in range(8):\n"
        "            b = a @ a\n"
        "    beside.append(fit_predict())\n"
        "    torch.cuda.synchronize()\n"
        "print('FOREIGN alone %%.2f ms, beside the products %%s ms, give-ups %%d' %% (alone * 1e3, ' '.join('%%.2f' %% (t * 1e3) for t in beside), _lib.stat('panel_giveups')))\n"
    ) % (root, os.path.join(root, "geostatssolvers.jl_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FOREIGN alone" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    print([l for l in r.stdout.splitlines() if l.startswith("FOREIGN")][0])
