"""FFTGS on the device vs the oracle (which follows fft.jl:62-198 with C2C pocketfft).

Tolerances.  Models with an algebraically decaying spectrum (exponential, spherical, Matern):
spectral amplitude F relative 1e-12 of max(F) (SURVEY.md section 8c "spectra rel 1e-12") and
realisations 1e-9 absolute -- the phase X/|X| of a noise coefficient is conditioned like 1/|X|,
|X| ~ sqrt(N/12), so R2C-vs-C2C rounding differences of 1e-16 N stay far below that.
Gaussian variogram: its true spectrum underflows below the FFT's own rounding noise
delta ~ 1e-16 N max|fft(C)|, and F = sqrt(|fft(C)|) turns an absolute error delta at a true zero
into sqrt(delta): the reference's own F is rounding noise there.  Stated tolerance for that model:
F 1e-6 relative to max(F), realisations 1e-6 absolute (same level as the Gaussian kriging
tolerance of SURVEY.md section 8c)."""
import os

import numpy as np
import pytest

from oracle import fftgs as O, philox
from oracle.variogram import Variogram

pytestmark = pytest.mark.gpu

GRIDS = [((64,), None), ((33,), (0.5,)), ((32, 24), None), ((15, 9), (2.0, 1.0)), ((16, 12, 10), None),
         ((9, 8, 7), (1.0, 2.0, 0.5)), ((100, 100), None)]


def _handle(kind, dims, spacing=None, mean=0.0, **kw):
    import gss
    from gss.engine import FFTGSHandle
    ctor = dict(gaussian=gss.GaussianVariogram, exponential=gss.ExponentialVariogram,
                spherical=gss.SphericalVariogram)[kind]
    radii = kw.pop("radii", None)
    vg = ctor(gss.MetricBall(tuple(radii)), **kw) if radii is not None else ctor(**kw)
    return FFTGSHandle(vg, dims, spacing, mean)


@pytest.mark.parametrize("dims,spacing", GRIDS)
def test_spectrum_matches_oracle(dims, spacing):
    kw = dict(range=0.3 * dims[0], sill=1.7, nugget=0.2)
    h = _handle("exponential", dims, spacing, **kw)
    F = h.spectrum()
    pre = O.preprocess(Variogram("exponential", **kw), dims, spacing=spacing)
    assert F[0] == 0.0
    assert np.max(np.abs(F - pre.F.ravel())) < 1e-12 * np.max(pre.F)
    h.close()


@pytest.mark.parametrize("dims,spacing", GRIDS)
def test_realisation_from_supplied_noise_matches_oracle(dims, spacing):
    rng = np.random.default_rng(sum(dims))
    N = int(np.prod(dims))
    kw = dict(range=0.25 * dims[0], sill=2.0)
    noise = rng.uniform(size=(2, N))
    h = _handle("gaussian", dims, spacing, mean=1.25, **kw)
    z = h.realize(0, 0, 2, noise=noise)
    pre = O.preprocess(Variogram("gaussian", **kw), dims, spacing=spacing, mean=1.25)
    for r in range(2):
        ref = O.solvesingle(pre, noise[r])
        assert np.max(np.abs(z[r] - ref)) < 1e-6                    # Gaussian model: see module docstring
        zc = z[r] - 1.25
        assert abs(np.sum(zc * zc) / (N - 1) - 2.0) < 1e-9          # fft.jl:169-170
    h.close()


@pytest.mark.parametrize("kind", ["exponential", "spherical"])
@pytest.mark.parametrize("dims,spacing", GRIDS)
def test_realisation_tight_tolerance_for_algebraic_spectra(kind, dims, spacing):
    rng = np.random.default_rng(sum(dims) + 1)
    N = int(np.prod(dims))
    kw = dict(range=0.25 * dims[0], sill=2.0, nugget=0.1)
    noise = rng.uniform(size=(1, N))
    h = _handle(kind, dims, spacing, mean=-0.5, **kw)
    z = h.realize(0, 0, 1, noise=noise)
    ref = O.solvesingle(O.preprocess(Variogram(kind, **kw), dims, spacing=spacing, mean=-0.5), noise[0])
    assert np.max(np.abs(z[0] - ref)) < 1e-9
    h.close()


def test_device_philox_noise_is_the_oracle_noise_bit_for_bit():
    import ctypes as C
    from gss import _lib
    l = _lib.lib()
    for n in (1, 2, 7, 4097):
        out = np.empty(n)
        _lib.check(l.gss_philox_uniform(12345678901234, 3, n, _lib.ptr(out), 0, None))
        assert np.array_equal(out, philox.uniform(12345678901234, 3, n))
        _lib.check(l.gss_philox_normal(99, 5, n, _lib.ptr(out), 0, None))
        assert np.max(np.abs(out - philox.normal(99, 5, n))) < 1e-13


def test_seeded_realisations_match_oracle_and_do_not_depend_on_batching():
    dims = (24, 20, 16)
    kw = dict(range=6.0)
    h = _handle("exponential", dims, **kw)
    z = h.realize(4, 0, 5)
    ref = O.realize(O.preprocess(Variogram("exponential", **kw), dims), 4, 0, 5)
    assert np.max(np.abs(z - ref)) < 1e-9
    # realisation r depends only on (seed, r): sharding over ranks cannot change results
    z2 = h.realize(4, 3, 2)
    assert np.array_equal(z2, z[3:5])
    h.close()


def test_anisotropic_and_view():                   # test/simulation/fft.jl:8-22
    import torch
    dims = (100, 100)
    h = _handle("gaussian", dims, radii=(20.0, 5.0))
    pre = O.preprocess(Variogram("gaussian", radii=(20.0, 5.0)), dims)
    assert np.max(np.abs(h.spectrum() - pre.F.ravel())) < 1e-6 * np.max(pre.F)
    inds = np.arange(5000)
    z = h.realize(2022, 0, 3, inds=inds)
    assert z.shape == (3, 5000)
    assert np.max(np.abs(z - O.realize(pre, 2022, 0, 3, inds=inds))) < 1e-6
    zd = h.realize(2022, 0, 3, inds=inds, device=True)
    torch.cuda.synchronize()
    assert np.array_equal(zd.cpu().numpy(), z)
    h.close()


def test_solve_api_unconditional_view():            # test/simulation/fft.jl:14-22
    import gss
    grid = gss.CartesianGrid(100, 100)
    vgrid = gss.view(grid, range(0, 5000))
    problem = gss.SimulationProblem(vgrid, ("z", float), 3)
    solver = gss.FFTGS(("z", dict(variogram=gss.GaussianVariogram(range=10.0))), rng=2022)
    sol = gss.solve(problem, solver)
    assert gss.domain(sol) == vgrid
    assert len(sol[0].z) == 5000 and len(sol["z"]) == 3


def test_large_grid_properties():
    """256^3 (the oracle would need minutes): size-independent properties only."""
    import torch
    from gss.engine import FFTGSHandle
    import gss
    e = 256
    N = e ** 3
    h = FFTGSHandle(gss.ExponentialVariogram(range=25.0, sill=3.0), (e, e, e), mean=-2.0)
    z = h.realize(4, 0, 2, device=True)
    torch.cuda.synchronize()
    for r in range(2):
        zc = z[r] + 2.0
        assert abs(float((zc * zc).sum()) / (N - 1) - 3.0) < 1e-8
        assert abs(float(zc.mean())) < 1e-10                       # DC killed (fft.jl:103)
    assert not torch.equal(z[0], z[1])
    # |FFT(Z - mean)| is proportional to F: compare through torch's own FFT on one realisation
    F = torch.as_tensor(h.spectrum(), device="cuda").reshape(e, e, e)
    A = torch.fft.fftn((z[0] + 2.0).reshape(e, e, e)).abs()
    sel = F > 1e-3 * F.max()
    ratio = A[sel] / F[sel]
    assert float((ratio.max() - ratio.min()) / ratio.mean()) < 1e-8
    h.close()


@pytest.mark.parametrize("maxneighbors", [None, 2])
def test_conditional_simulation_matches_oracle(maxneighbors):
    """test/simulation/fft.jl:24-32 (conditional, 3 data) at the oracle's tolerance for a spherical model;
    the conditioning path is fft.jl:105-135 + 176-192 (simple kriging with the solver's mean)."""
    import gss
    coords = np.array([(25.0, 25.0), (50.0, 75.0), (75.0, 50.0)]) * 0.4
    vals = [1.0, -1.0, 1.0]
    grid = gss.CartesianGrid(40, 40)
    problem = gss.SimulationProblem(gss.georef({"z": vals}, coords), grid, ("z", float), 4)
    params = dict(variogram=gss.SphericalVariogram(range=12.0), mean=0.2)
    if maxneighbors:
        params["maxneighbors"] = maxneighbors
    sol = gss.solve(problem, gss.FFTGS(("z", params), rng=2022))
    pre = O.preprocess(Variogram("spherical", range=12.0), (40, 40), mean=0.2, data_coords=coords, data_vals=vals,
                       maxneighbors=maxneighbors)
    for r in range(4):
        ref = O.solvesingle(pre, philox.uniform(2022, r, 1600))
        assert np.max(np.abs(sol[r].z - ref)) < 1e-8
    # data are honoured at the data cells up to the centroid offset (fft.jl:112 vs :180-184)
    assert np.max(np.abs(sol[0].z[pre.dinds] - np.array(vals))) < 0.35


POW2_GRIDS = [(32, 16, 16), (64, 128, 64), (256, 64, 128), (1024, 32, 16)]


@pytest.mark.parametrize("dims", POW2_GRIDS)
def test_fused_pipeline_matches_oracle_and_rocfft_path(dims, monkeypatch):
    """Power-of-two 3-D grids take the fused five-pass pipeline (csrc/fftgs_fused.h); it must agree with the
    oracle and with the rocFFT pipeline of the same library.  (Since round 4 the small ones go to the generic passes by
    default, where realisations share launches -- test_small_power_of_two_grids_take_the_batched_generic_passes; the
    fused kernels of every size stay covered here through GSS_FFTGS_PATH=fused.)"""
    monkeypatch.setenv("GSS_FFTGS_PATH", "fused")
    N = int(np.prod(dims))
    kw = dict(range=0.2 * dims[0], sill=1.4, nugget=0.05)
    pre = O.preprocess(Variogram("exponential", **kw), dims, mean=0.3)
    h = _handle("exponential", dims, mean=0.3, **kw)
    z = h.realize(11, 2, 3)
    ref = O.realize(pre, 11, 2, 3)
    assert np.max(np.abs(z - ref)) < 1e-9
    noise = np.random.default_rng(N).uniform(size=(2, N))
    zn = h.realize(0, 0, 2, noise=noise)
    for r in range(2):
        assert np.max(np.abs(zn[r] - O.solvesingle(pre, noise[r]))) < 1e-9
    inds = np.arange(0, N, 7)
    assert np.array_equal(h.realize(11, 2, 1, inds=inds)[0], z[0][inds])
    h.close()
    monkeypatch.setenv("GSS_FFTGS_PATH", "rocfft")
    h2 = _handle("exponential", dims, mean=0.3, **kw)
    z2 = h2.realize(11, 2, 3)
    assert np.max(np.abs(z2 - z)) < 1e-10 and not np.array_equal(z2, z)     # two different code paths
    h2.close()


def test_fused_pipeline_512_cube_properties():
    import torch
    import gss
    from gss.engine import FFTGSHandle
    e = 512
    N = e ** 3
    h = FFTGSHandle(gss.ExponentialVariogram(range=50.0), (e, e, e), mean=1.0)
    z = h.realize(4, 0, 1, device=True)
    torch.cuda.synchronize()
    zc = z[0] - 1.0
    assert abs(float((zc * zc).sum()) / (N - 1) - 1.0) < 1e-8 and abs(float(zc.mean())) < 1e-10
    F = torch.as_tensor(h.spectrum(), device="cuda").reshape(e, e, e)
    A = torch.fft.fftn(zc.reshape(e, e, e)).abs()
    sel = F > 1e-3 * F.max()
    ratio = A[sel] / F[sel]
    assert float((ratio.max() - ratio.min()) / ratio.mean()) < 1e-8
    h.close()


def test_slab_order_of_the_strided_passes_changes_nothing():
    """512-point y and z lines run their three strided passes slab by slab over the x tiles, slabs alternating over
    three streams (GSS_FFTGS_SLAB / GSS_FFTGS_SLAB_STREAMS, read once per process): the realisations must be bit for bit
    those of one launch per pass.  Two subprocesses, a grid with a ragged last slab (5 x tiles, 2 per slab)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, hashlib, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import gss\n"
        "from gss.engine import FFTGSHandle\n"
        "h = FFTGSHandle(gss.ExponentialVariogram(range=9.0), (64, 512, 512), mean=0.5)\n"
        "z = h.realize(11, 3, 2, device=True)\n"
        "torch.cuda.synchronize()\n"
        "print('DIGEST', hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest(), float(z.std()))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
         os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geostatssolvers.jl_amd"))
    outs = []
    for env_slab in ("0", "2"):
        env = dict(os.environ, GSS_FFTGS_SLAB=env_slab)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert outs[0] == outs[1]
    assert abs(float(outs[0].split()[2]) - 1.0) < 0.05


@pytest.mark.parametrize("dims", POW2_GRIDS)
def test_fused_spectrum_matches_oracle(dims, monkeypatch):
    """On power-of-two 3-D grids gss_fftgs_create builds F on the library's own passes (covariance rows generated
    inside the x pass, y and z forward passes, amplitude kernel undoing the bit reversal) -- fft.jl:96-103.
    Same 1e-12 (relative to max F) as the rocFFT build of the general grids."""
    monkeypatch.setenv("GSS_FFTGS_PATH", "fused")
    kw = dict(range=0.3 * dims[0], sill=1.7, nugget=0.2)
    h = _handle("exponential", dims, (1.0, 2.0, 0.5), **kw)
    F = h.spectrum()
    h.close()
    pre = O.preprocess(Variogram("exponential", **kw), dims, spacing=(1.0, 2.0, 0.5))
    assert F[0] == 0.0
    assert np.max(np.abs(F - pre.F.ravel())) < 1e-12 * pre.F.max()
    ha = _handle("gaussian", dims, radii=(0.2 * dims[0], 5.0, 3.0))
    Fa = ha.spectrum()
    ha.close()
    prea = O.preprocess(Variogram("gaussian", radii=(0.2 * dims[0], 5.0, 3.0)), dims)
    assert np.max(np.abs(Fa - prea.F.ravel())) < 1e-6 * prea.F.max()


@pytest.mark.parametrize("dims", [(64, 32, 16), (24, 20, 16), (100, 100)])
def test_state_broadcast_adopt(dims):
    """fft.jl:62 runs once: a handle created with GSS_FFTGS_NO_SPECTRUM refuses to realise, receives another
    handle's state (what torch.distributed.broadcast does between ranks) and then reproduces its realisations."""
    import gss
    from gss._lib import GSSError
    from gss.engine import FFTGSHandle
    vg = gss.SphericalVariogram(range=0.3 * dims[0], sill=1.3, nugget=0.1)
    a = FFTGSHandle(vg, dims, mean=0.25)
    b = FFTGSHandle(vg, dims, mean=0.25, spectrum=False)
    with pytest.raises(GSSError, match="no spectrum"):
        b.realize(5, 0, 1)
    ta, tb = a.state_tensor(), b.state_tensor()
    assert ta.shape == tb.shape and ta.is_cuda and ta.data_ptr() != tb.data_ptr()
    tb.copy_(ta)
    b.adopt_state()
    assert np.array_equal(a.realize(5, 3, 2), b.realize(5, 3, 2))
    assert np.array_equal(a.spectrum(), b.spectrum())
    a.close()
    b.close()


def test_pipelined_realisations_equal_single_calls():
    """One gss_fftgs_realize call with several realisations runs them as a two-stream pipeline (noise / x pass of
    r + 1 on a helper stream beside the strided passes of r, two half-spectrum buffers).  The fields must be
    bit-identical to one call per realisation, also on the second call (buffers and events re-used) and with an
    index view."""
    import gss
    from gss.engine import FFTGSHandle
    dims = (64, 32, 32)
    h = FFTGSHandle(gss.ExponentialVariogram(range=9.0, sill=2.0, nugget=0.1), dims, mean=0.5)
    single = np.stack([h.realize(21, r, 1)[0] for r in range(7)])
    for _ in range(2):
        assert np.array_equal(h.realize(21, 0, 7), single)
        assert np.array_equal(h.realize(21, 2, 5), single[2:])
    inds = np.arange(5, 64 * 32 * 32, 11)
    assert np.array_equal(h.realize(21, 1, 4, inds=inds), single[1:5][:, inds])
    h.close()


def test_rocfft_kernels_are_kept_for_later_processes():
    """Grids outside the fused passes go through rocFFT, which compiles its kernels at run time; unless the user chose
    a place, the library points ROCFFT_RTC_CACHE_PATH at ~/.cache/gss_hip (fftgs.hip) so that a later process finds
    them (0.1 s instead of 1.7 s for its first plan)."""
    import ctypes
    import os
    import gss
    from gss.engine import FFTGSHandle
    h = FFTGSHandle(gss.SphericalVariogram(range=8.0), (60, 50))
    z = h.realize(3, 0, 2)
    assert z.shape == (2, 3000) and np.isfinite(z).all()
    h.close()
    libc = ctypes.CDLL(None)
    libc.getenv.restype = ctypes.c_char_p
    path = libc.getenv(b"ROCFFT_RTC_CACHE_PATH")
    if not path:
        home = os.environ.get("XDG_CACHE_HOME") or os.path.expanduser("~")
        if not os.access(home, os.W_OK):
            pytest.skip("no writable cache directory on this machine: rocFFT keeps its kernels for the process only")
    assert path, "no kernel cache path in the process environment"
    if os.environ.get("ROCFFT_RTC_CACHE_PATH") is None:          # not chosen by the user: the library's default
        assert path.decode().endswith(os.path.join("gss_hip", "rocfft_kernels.db"))
        assert os.path.exists(path.decode())


# ---- 512-point y / z lines: ff_axis2_fast_kernel<MODE,3,512,9> and the slab order (the dominant kernels of configs[2]) ----

LINE512_GRIDS = [(32, 512, 512), (64, 512, 512)]
_LINE512_KW = dict(range=40.0, sill=1.4, nugget=0.05)
_line512_cache = {}


def _line512_oracle(dims):
    """Oracle fields of one grid (fft.jl:62-198 through pocketfft), computed once per session: the spectrum, two
    Philox realisations (seed 11, realisations 2 and 3) and one realisation from supplied noise."""
    if dims not in _line512_cache:
        N = int(np.prod(dims))
        pre = O.preprocess(Variogram("exponential", **_LINE512_KW), dims, mean=0.3)
        noise = np.random.default_rng(N + 7).uniform(size=N)
        _line512_cache[dims] = dict(F=pre.F.ravel().copy(), z=O.realize(pre, 11, 2, 2), zn=O.solvesingle(pre, noise))
    return _line512_cache[dims]


_LINE512_CHILD = """
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gss
from gss.engine import FFTGSHandle
dims = %r
N = int(np.prod(dims))
h = FFTGSHandle(gss.ExponentialVariogram(**%r), dims, mean=0.3)
noise = np.random.default_rng(N + 7).uniform(size=(1, N))
np.savez(%r, F=h.spectrum(), z=h.realize(11, 2, 2), zn=h.realize(0, 0, 1, noise=noise)[0])
h.close()
"""


@pytest.mark.parametrize("slab", ["0", "2"])
@pytest.mark.parametrize("dims", LINE512_GRIDS)
def test_fused_pipeline_on_512_point_lines_matches_oracle(dims, slab, tmp_path):
    """fft.jl:163-170 at the line length the headline number is quoted on.  y and z lines of 512 points run the
    pass-by-pass kernel `ff_axis2_fast_kernel<MODE,3,512,9>` (csrc/fftgs_fused.h) and, with GSS_FFTGS_SLAB != 0
    (the default), the slab order over three streams; both are compared here with the oracle -- spectrum 1e-12 of
    max F, Philox and supplied-noise realisations 1e-9 -- in a child process per setting because the switch is read
    once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dev.npz")
    code = _LINE512_CHILD % (root, os.path.join(root, "geostatssolvers.jl_amd"), dims, _LINE512_KW, out)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GSS_FFTGS_SLAB=slab), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    dev = np.load(out)
    ref = _line512_oracle(dims)
    assert dev["F"][0] == 0.0
    assert np.max(np.abs(dev["F"] - ref["F"])) < 1e-12 * ref["F"].max()
    assert np.max(np.abs(dev["z"] - ref["z"])) < 1e-9
    assert np.max(np.abs(dev["zn"] - ref["zn"])) < 1e-9


def _torch_restatement_512(e, rng_range, mean, U):
    """fft.jl:96-103 and :163-170 written with torch's complex-to-complex FFT on the device: an implementation that
    shares nothing with the library (its pipelines are real-to-complex; this one is the reference's literal C2C form)."""
    import torch
    ax = torch.arange(e, device="cuda", dtype=torch.float64) + 0.5
    c = ax[e // 2 - 1]                                              # centre cell, fft.jl:69-70 (1-based e // 2)
    d2 = ((ax - c) ** 2)
    h = torch.sqrt(d2[:, None, None] + d2[None, :, None] + d2[None, None, :])
    C = torch.exp(-3.0 * h / rng_range)                             # exponential model, sill 1, no nugget
    del h
    F = torch.sqrt(torch.abs(torch.fft.fftn(torch.fft.fftshift(C))))    # fft.jl:102
    del C
    F.view(-1)[0] = 0.0                                             # fft.jl:103
    X = torch.fft.fftn(U.reshape(e, e, e))
    P = F * torch.exp(1j * torch.angle(X))                          # fft.jl:163
    del X
    Z = torch.fft.ifftn(P).real                                     # fft.jl:166
    del P
    s2 = float((Z * Z).sum()) / (e ** 3 - 1)                        # fft.jl:169
    return (Z * (1.0 / s2) ** 0.5 + mean).reshape(-1), F.reshape(-1)   # fft.jl:170


def test_config3_full_size_realisation_against_two_other_implementations(monkeypatch):
    """configs[2] at its full 512^3: one realisation of the fused pipeline against (i) a literal complex-to-complex
    restatement of fft.jl:96-103,163-170 in torch on the same noise (1e-9, the tolerance of the oracle comparisons;
    spectrum 1e-12 of max F) and (ii) the library's rocFFT pipeline (a different code path that is oracle-tested at
    the smaller sizes; 1e-10)."""
    import torch
    import gss
    from gss.engine import FFTGSHandle
    e = 512
    N = e ** 3
    vg = gss.ExponentialVariogram(range=50.0)
    h = FFTGSHandle(vg, (e, e, e), mean=1.0)
    U = torch.rand(N, device="cuda", dtype=torch.float64, generator=torch.Generator("cuda").manual_seed(5))
    zn = h.realize(0, 0, 1, noise=U.reshape(1, N), device=True)[0].clone()
    zp = h.realize(4, 1, 1, device=True)[0].clone()
    Fdev = torch.as_tensor(h.spectrum(), device="cuda")
    h.close()
    torch.cuda.synchronize()
    ref, F = _torch_restatement_512(e, 50.0, 1.0, U)
    assert float((Fdev - F).abs().max()) < 1e-12 * float(F.max())
    assert float((zn - ref).abs().max()) < 1e-9
    del ref, F, Fdev
    monkeypatch.setenv("GSS_FFTGS_PATH", "rocfft")
    h2 = FFTGSHandle(vg, (e, e, e), mean=1.0)
    zn2 = h2.realize(0, 0, 1, noise=U.reshape(1, N), device=True)[0]
    torch.cuda.synchronize()
    assert float((zn2 - zn).abs().max()) < 1e-10 and not torch.equal(zn2, zn)
    zp2 = h2.realize(4, 1, 1, device=True)[0]                       # Philox noise: the in-register generator of the
    torch.cuda.synchronize()                                        # x pass against the stand-alone noise kernel
    assert float((zp2 - zp).abs().max()) < 1e-10 and not torch.equal(zp2, zp)
    h2.close()


def _host_can_do_512():
    import os
    if os.environ.get("GSS_TEST_HOST_512") == "0":
        return False
    try:
        import psutil
        return psutil.virtual_memory().available > 48 * 2 ** 30 and (os.cpu_count() or 1) >= 8
    except Exception:                                   # noqa: BLE001
        return os.environ.get("GSS_TEST_HOST_512") == "1"


@pytest.mark.skipif(not _host_can_do_512(), reason="needs ~20 GiB of free host memory and a few cores for 512^3 host FFTs "
                                                   "(GSS_TEST_HOST_512=0 switches it off)")
def test_config3_full_size_realisation_against_host_fft():
    """configs[2] at full size against the host: fft.jl:163-170 through scipy's threaded pocketfft (complex-to-complex,
    as the reference), the spectrum taken from the device handle (oracle-checked at 16 M cells above).  A few seconds on
    the GPU box's host cores (8.3e-14 in round 4, profiles/r04_fftgs_512_host_parity.txt)."""
    import os
    import scipy.fft
    import gss
    from gss.engine import FFTGSHandle
    e = 512
    N = e ** 3
    w = os.cpu_count()
    h = FFTGSHandle(gss.ExponentialVariogram(range=50.0), (e, e, e), mean=1.0)
    U = np.random.default_rng(5).uniform(size=N)
    z = h.realize(0, 0, 1, noise=U.reshape(1, N))[0]
    F = h.spectrum().reshape(e, e, e)
    h.close()
    X = scipy.fft.fftn(U.reshape(e, e, e), workers=w)
    X /= np.abs(X)                                                  # exp(i angle(X)), fft.jl:163
    X *= F
    Z = scipy.fft.ifftn(X, workers=w, overwrite_x=True).real        # fft.jl:166
    Z *= np.sqrt(1.0 / (np.sum(Z * Z) / (N - 1)))                   # fft.jl:169-170
    Z += 1.0
    err = float(np.max(np.abs(z - Z.ravel())))
    print("512^3 fused pipeline vs scipy.fft (complex-to-complex): max abs difference %.3e" % err)
    assert err < 1e-9


# ---- generic pipeline (fftgs_generic.h): 2-D grids and sizes 2^a 3^b 5^c 7^d on the library's own Stockham passes ----------

GENERIC_GRIDS = [(100, 100), (60, 50), (50, 64), (64, 64), (1000, 36), (36, 1000), (4096, 16), (250, 250),
                 (48, 36, 30), (100, 100, 100), (20, 18, 10), (64, 48, 40), (30, 625, 8),
                 (4, 2), (6, 3), (8, 8, 8), (16, 16, 16), (4, 1024),
                 # long y lines (1 024 < n2 <= 4 096): the split n2 = L1 L2 of fftgs_generic.h
                 (64, 2048), (2048, 2048), (16, 4096), (100, 3000), (36, 1500), (250, 1280), (1024, 4096),
                 # radix 7 (140, 210, 280, 350, 420, 700 ...)
                 (70, 98), (140, 140), (28, 49, 14), (686, 20), (98, 2744), (14, 7, 7), (210, 210, 70), (350, 700),
                 # 1-D grids (x passes and an elementwise phase step)
                 (100,), (1000,), (4096,), (6,), (14,), (3000,), (4,)]


@pytest.mark.parametrize("dims", GENERIC_GRIDS)
def test_generic_pipeline_matches_oracle_and_rocfft_path(dims, monkeypatch):
    """Grids the power-of-two pipeline does not take but whose sizes factor into 2, 3 and 5 -- the reference's own test
    grids are 100 x 100 (test/simulation/fft.jl:4,11,26) -- run on the library's own Stockham passes of radix 2 / 3 / 4 /
    5 / 8 (three passes in 2-D, five in 3-D) instead of noise kernel + rocFFT R2C + phase kernel + rocFFT C2R.  Spectrum
    1e-12 of max F, realisations from Philox and from supplied noise 1e-9 against the oracle; 1e-10 against the rocFFT
    pipeline of the same library, which is a different code path (and must differ in the last bits)."""
    from gss import _lib
    N = int(np.prod(dims))
    kw = dict(range=0.2 * dims[0], sill=1.4, nugget=0.05)
    pre = O.preprocess(Variogram("exponential", **kw), dims, mean=0.3)
    h = _handle("exponential", dims, mean=0.3, **kw)
    F = h.spectrum()
    assert F[0] == 0.0 and np.max(np.abs(F - pre.F.ravel())) < 1e-12 * pre.F.max()
    _lib.profile_reset(); _lib.profile_enable(True)
    z = h.realize(11, 2, 3)
    _lib.profile_enable(False)
    # (one profile scope per batch of realisations that share the launches: 1 .. 3 scopes for the three)
    assert 1 <= _lib.profile_read("fftgs_generic")[1] <= 3, "the realisations did not run on the generic passes"
    ref = O.realize(pre, 11, 2, 3)
    assert np.max(np.abs(z - ref)) < 1e-9
    noise = np.random.default_rng(N).uniform(size=(2, N))
    zn = h.realize(0, 0, 2, noise=noise)
    for r in range(2):
        assert np.max(np.abs(zn[r] - O.solvesingle(pre, noise[r]))) < 1e-9
    inds = np.arange(0, N, 7)
    assert np.array_equal(h.realize(11, 2, 1, inds=inds)[0], z[0][inds])
    h.close()
    monkeypatch.setenv("GSS_FFTGS_PATH", "rocfft")
    h2 = _handle("exponential", dims, mean=0.3, **kw)
    z2 = h2.realize(11, 2, 3)
    assert np.max(np.abs(z2 - z)) < 1e-10 and not np.array_equal(z2, z)
    h2.close()


def _random_shapes():
    """Forty grid shapes drawn once (fixed seed): 1-D, 2-D and 3-D, axis lengths from smooth numbers (2^a 3^b 5^c 7^d, the
    generic passes: every radix order, odd half lengths, the long-line split), powers of two (fused pipeline) and
    arbitrary integers (rocFFT), at most 3 million cells."""
    # (GSS_TEST_SHAPES="count,seed": a longer hunt over other shapes)
    count, seed = (int(v) for v in os.environ.get("GSS_TEST_SHAPES", "40,20260405").split(","))
    rng = np.random.default_rng(seed)
    smooth = sorted({2 ** a * 3 ** b * 5 ** c * 7 ** d for a in range(13) for b in range(8) for c in range(6) for d in range(5)
                     if 2 <= 2 ** a * 3 ** b * 5 ** c * 7 ** d <= 4096})
    shapes = []
    while len(shapes) < count:
        nd = int(rng.choice([1, 2, 2, 2, 3, 3]))
        cap = {1: 4096, 2: 4096, 3: 200}[nd]
        dims = []
        for _ in range(nd):
            kind = rng.integers(0, 10)
            if kind < 7:
                n = int(rng.choice([v for v in smooth if v <= cap]))
            elif kind < 8:
                n = int(2 ** rng.integers(1, int(np.log2(cap)) + 1))
            else:
                n = int(rng.integers(2, cap + 1))
            dims.append(n)
        if 2 <= int(np.prod(dims)) <= 3_000_000 and tuple(dims) not in shapes:
            shapes.append(tuple(dims))
    return shapes


@pytest.mark.parametrize("dims", _random_shapes(), ids=lambda d: "x".join(map(str, d)))
def test_any_grid_shape_matches_oracle(dims):
    """Whatever pipeline the library picks for a shape (fused power-of-two passes, generic Stockham passes with or without
    the long-line split, rocFFT), spectrum and realisations are the oracle's: 1e-12 of max F / 1e-9."""
    N = int(np.prod(dims))
    kw = dict(range=max(2.0, 0.15 * dims[0]), sill=0.9, nugget=0.1)
    pre = O.preprocess(Variogram("exponential", **kw), dims, mean=-0.2)
    h = _handle("exponential", dims, mean=-0.2, **kw)
    F = h.spectrum()
    assert np.max(np.abs(F - pre.F.ravel())) < 1e-12 * pre.F.max()
    z = h.realize(3, 1, 2)
    assert np.max(np.abs(z - O.realize(pre, 3, 1, 2))) < 1e-9
    noise = np.random.default_rng(N).uniform(size=(1, N))
    assert np.max(np.abs(h.realize(0, 0, 1, noise=noise)[0] - O.solvesingle(pre, noise[0]))) < 1e-9
    h.close()


@pytest.mark.parametrize("dims", [(60, 50), (100, 100), (24, 20, 18), (64, 2048), (500,), (101, 101), (51, 47, 23), (101,)])
def test_batched_realisations_equal_single_ones(dims, monkeypatch):
    """Small grids on the generic passes: up to 64 realisations share every launch (grid y) -- and on the rocFFT pipeline
    (odd and non-smooth sizes: the last three shapes) 64 or 16 realisations share every plan execution.  Seventy realisations in one
    call (batches of 64 + 6, or fewer per batch on the larger grid) are bit-identical to seventy calls of one, from Philox
    and from supplied noise, into device memory, into host memory through a ring of small chunks (batches cut at the
    chunk ends: GSS_OUT_CHUNK_MB=1), and with an index subset; a few of them against the oracle."""
    import torch
    N, R = int(np.prod(dims)), 70
    kw = dict(range=0.2 * dims[0], sill=1.1, nugget=0.02)
    h = _handle("exponential", dims, mean=0.1, **kw)
    singles = np.stack([h.realize(9, 5 + r, 1)[0] for r in range(R)])
    pre = O.preprocess(Variogram("exponential", **kw), dims, mean=0.1)
    for r in (0, 1, 63, 64, 69):
        assert np.max(np.abs(singles[r] - O.realize(pre, 9, 5 + r, 1)[0])) < 1e-9
    zd = h.realize(9, 5, R, device=True)
    assert np.array_equal(zd.cpu().numpy(), singles)
    monkeypatch.setenv("GSS_OUT_CHUNK_MB", "1")
    assert np.array_equal(h.realize(9, 5, R), singles)
    inds = np.arange(3, N, 5)
    assert np.array_equal(h.realize(9, 5, R, inds=inds), singles[:, inds])
    assert np.array_equal(h.realize(9, 5, R, inds=inds, device=True).cpu().numpy(), singles[:, inds])
    monkeypatch.delenv("GSS_OUT_CHUNK_MB")
    noise = np.random.default_rng(4).uniform(size=(9, N))
    zn1 = np.stack([h.realize(0, 0, 1, noise=noise[r:r + 1])[0] for r in range(9)])
    znd = h.realize(0, 0, 9, noise=torch.as_tensor(noise, device="cuda"))
    assert np.array_equal(znd.cpu().numpy(), zn1)
    assert np.array_equal(h.realize(0, 0, 9, noise=noise), zn1)
    assert np.max(np.abs(zn1[8] - O.solvesingle(pre, noise[8]))) < 1e-9
    h.close()


@pytest.mark.parametrize("dims", [(32, 32, 32), (64, 128, 64), (128, 128, 64), (1024, 32, 16)])
def test_small_power_of_two_grids_take_the_batched_generic_passes(dims, monkeypatch):
    """Power-of-two 3-D grids whose half spectrum is at most 12 MiB run on the generic passes (realisations share
    launches there: 32^3 six times, 64^3 twice as fast as one realisation per launch on the fused kernels); the two
    pipelines agree to 1e-10 and the default one with the oracle to 1e-9."""
    from gss import _lib
    kw = dict(range=0.2 * dims[0], sill=1.4, nugget=0.05)
    h = _handle("exponential", dims, mean=0.3, **kw)
    _lib.profile_reset(); _lib.profile_enable(True)
    z = h.realize(11, 2, 12)
    _lib.profile_enable(False)
    assert 1 <= _lib.profile_read("fftgs_generic")[1] <= 2
    h.close()
    pre = O.preprocess(Variogram("exponential", **kw), dims, mean=0.3)
    assert np.max(np.abs(z - O.realize(pre, 11, 2, 12))) < 1e-9
    monkeypatch.setenv("GSS_FFTGS_PATH", "fused")
    h2 = _handle("exponential", dims, mean=0.3, **kw)
    _lib.profile_reset(); _lib.profile_enable(True)
    z2 = h2.realize(11, 2, 12)
    _lib.profile_enable(False)
    assert _lib.profile_read("fftgs_generic")[1] == 0
    h2.close()
    assert np.max(np.abs(z2 - z)) < 1e-10 and not np.array_equal(z2, z)


def test_generic_pipeline_with_anisotropy_spacing_and_state_adoption():
    """The reference's anisotropic case (`GaussianVariogram(MetricBall((20., 5.)))` on 100 x 100, test/simulation/fft.jl:
    8-12) and a grid with unequal spacing on the generic passes (Gaussian model: 1e-6, module docstring), and a handle
    that adopts another handle's state."""
    import gss
    from gss.engine import FFTGSHandle
    dims = (100, 100)
    h = _handle("gaussian", dims, radii=(20.0, 5.0))
    pre = O.preprocess(Variogram("gaussian", radii=(20.0, 5.0)), dims)
    assert np.max(np.abs(h.spectrum() - pre.F.ravel())) < 1e-6 * np.max(pre.F)
    assert np.max(np.abs(h.realize(2022, 0, 3) - O.realize(pre, 2022, 0, 3))) < 1e-6
    h.close()
    dims = (60, 45, 20)
    kw = dict(range=9.0, sill=2.0, nugget=0.1)
    a = _handle("spherical", dims, (1.0, 2.0, 0.5), mean=-0.5, **kw)
    pre = O.preprocess(Variogram("spherical", **kw), dims, spacing=(1.0, 2.0, 0.5), mean=-0.5)
    za = a.realize(5, 0, 2)
    assert np.max(np.abs(za - O.realize(pre, 5, 0, 2))) < 1e-9
    b = FFTGSHandle(gss.SphericalVariogram(**kw), dims, (1.0, 2.0, 0.5), -0.5, spectrum=False)
    b.state_tensor().copy_(a.state_tensor())
    b.adopt_state()
    assert np.array_equal(b.realize(5, 0, 2), za)
    a.close()
    b.close()
