#!/usr/bin/env python3
"""Measures the remaining BASELINE.json configs on ONE MI355X (per-GPU share of each config) and prints one JSON
line per config with its roofline and a bounded CPU baseline (oracle).  `bench.py` stays the headline
(configs[1]); this script feeds DESIGN.md section 5 and profiles/.

    python bench_configs.py [--configs 1h,2,2h,3,4,bigk,lu,idw,lwr,sgs,sgs_bigk,est_all,cond_fftgs] [--quick]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP64_PEAK = 78.6
HBM_PEAK = 8000.0


def sync():
    torch.cuda.synchronize()


def cfg2_fftgs(a, gss, _lib):
    """configs[2]: FFTGS 512^3, exponential range 50, 256 realisations over 8 GPUs -> 32 per GPU."""
    from gss.engine import FFTGSHandle
    e = 256 if a.quick else 512
    R = 8 if a.quick else 32
    N = e ** 3
    t0 = time.perf_counter()
    f = FFTGSHandle(gss.ExponentialVariogram(range=50.0 * e / 512), (e, e, e))
    sync()
    t_cold = time.perf_counter() - t0      # first device call of the process: code objects, first allocations
    f.close()
    t0 = time.perf_counter()
    f = FFTGSHandle(gss.ExponentialVariogram(range=50.0 * e / 512), (e, e, e))
    sync()
    t_pre = time.perf_counter() - t0
    out = torch.empty((1, N), dtype=torch.float64, device="cuda")
    f.realize(4, 0, 1, out=out)
    sync()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    for r in range(R):
        f.realize(4, r, 1, out=out)
    sync()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    parts = {k: _lib.profile_read(k) for k in ("fftgs_noise", "fftgs_fwd", "fftgs_phase", "fftgs_inv", "fftgs_p1", "fftgs_p2", "fftgs_p3", "fftgs_p4",
                                               "fftgs_p5")}
    var = float((out[0] * out[0]).sum().item() / (N - 1))
    # CPU baseline: oracle solvesingle (two C2C FFTs + temporaries like fft.jl:163-170) on a 256^3 grid
    from oracle import fftgs as O, philox
    from oracle.variogram import Variogram
    ce = 128 if a.quick else 256
    pre = O.preprocess(Variogram("exponential", range=50.0 * ce / 512), (ce, ce, ce))
    u = philox.uniform(4, 0, ce ** 3)
    t1 = time.perf_counter()
    O.solvesingle(pre, u)
    cdt = time.perf_counter() - t1
    return {"config": "configs[2] FFTGS %d^3 exponential, %d realisations on this GPU" % (e, R),
            "metric": "realisations/s", "value": round(R / dt, 2), "preprocess_s": round(t_pre, 4),
            "preprocess_first_call_of_process_s": round(t_cold, 3),
            "roofline": {"bound": "hbm", "achieved": round(32.0 * N * R / dt / 1e9, 1), "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": round(32.0 * N * R / dt / 1e9 / HBM_PEAK, 4)},
            "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in parts.items() if v[1]}, "sample_variance": var,
            "cpu_baseline": {"value": round(1.0 / cdt * (ce / e) ** 3, 4), "unit": "realisations/s", "cores": 1,
                             "kind": "port", "sample": "oracle.fftgs.solvesingle (numpy pocketfft C2C as fft.jl:163-166) on "
                             "%d^3 in %.1f s, scaled by cell count to %d^3" % (ce, cdt, e)}}


def cfg3_lugs(a, gss, _lib):
    """configs[3]: LUGS, 128x128 grid, 4 096 conditioning cells, spherical range 20, 100 realisations."""
    from gss.engine import LUGSHandle
    from oracle import fftgs as O
    g = 64 if a.quick else 128
    nd = g * g // 4
    cent = O.grid_centroids((g, g))
    N = g * g
    dlocs = np.sort(np.random.default_rng(5).permutation(N)[:nd])
    z1 = np.random.default_rng(50).normal(size=nd)
    vg = gss.SphericalVariogram(range=20.0)
    t0 = time.perf_counter()
    h = LUGSHandle(vg, cent, dlocs, z1)
    sync()
    t_pre = time.perf_counter() - t0
    t0 = time.perf_counter()
    h2 = LUGSHandle(vg, cent, dlocs, z1)
    sync()
    t_pre2 = time.perf_counter() - t0
    ns = N - nd
    flops = nd ** 3 / 3 + nd * nd * ns + nd * ns * ns + ns ** 3 / 3
    R = 100
    h.realize(5, 0, R, device=True)
    sync()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    y, _ = h.realize(5, 0, R, device=True)
    sync()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    gm = _lib.profile_read("lugs_gemm")
    ok = bool(torch.equal(y[:, torch.as_tensor(dlocs, device="cuda")][0], torch.as_tensor(z1, device="cuda")))
    # CPU baseline: oracle preprocess (LAPACK, threaded) at the same size
    from oracle import lugs as OL
    from oracle.variogram import Variogram
    t1 = time.perf_counter()
    OL.preprocess(Variogram("spherical", range=20.0), cent, cent[dlocs], z1)
    cdt = time.perf_counter() - t1
    return {"config": "configs[3] LUGS %dx%d grid, %d conditioning cells, ns=%d, spherical range 20, %d realisations" % (g, g, nd, ns, R),
            "metric": "preprocess seconds / realisations per s", "preprocess_s": round(min(t_pre, t_pre2), 4),
            "value": round(R / dt, 1), "unit": "realisations/s",
            "roofline": {"bound": "mfma", "stage": "preprocess (potrf+trsm+syrk+potrf)", "achieved": round(flops / min(t_pre, t_pre2) / 1e12, 3),
                         "peak": FP64_PEAK, "unit": "TFLOP/s", "frac": round(flops / min(t_pre, t_pre2) / 1e12 / FP64_PEAK, 4)},
            "realize_gemm_ms": round(gm[0] / max(gm[1], 1), 3), "hard_data_honoured": ok,
            "cpu_baseline": {"value": round(cdt, 3), "unit": "preprocess seconds", "cores": os.cpu_count(), "kind": "port",
                             "sample": "oracle.lugs.preprocess (numpy/scipy LAPACK, threaded BLAS) same size"}}


def cfg4_local(a, gss, _lib):
    """configs[4]: UK degree 1, 5 000 3-D data, k = 64, Matern-3/2; 10^7 points over 8 GPUs -> 1.25e6 per GPU."""
    from gss.engine import KrigHandle, UK
    n = 5000
    m = 100_000 if a.quick else 1_250_000
    x = np.random.default_rng(6).uniform(0, 100, (n, 3))
    z = 1.0 + 0.03 * x[:, 0] - 0.02 * x[:, 1] + 0.01 * x[:, 2] + np.random.default_rng(60).normal(size=n)
    x0 = np.random.default_rng(7).uniform(0, 100, (m, 3))
    x0d = torch.as_tensor(x0, device="cuda")
    h = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5), UK, x, z, degree=1, factor=False)
    h.predict_knn(x0d[:50000], 64)
    sync()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    mu, var, st = h.predict_knn(x0d, 64)
    sync()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    knn = _lib.profile_read("knn")
    loc = _lib.profile_read("krig_local")
    flop_pt = 68 ** 3 / 3 + 2 * 68 ** 2 + 8 * n
    from oracle import kriging as K
    from oracle.variogram import Variogram
    ns = 300
    t1 = time.perf_counter()
    rmu, rvar, _ = K.approxsolve(K.UK, Variogram("matern", range=30.0, nu=1.5), x, z, x0[:ns], 64, degree=1)
    cdt = time.perf_counter() - t1
    err = float(np.max(np.abs(mu[:ns].cpu().numpy() - rmu)))
    return {"config": "configs[4] UK degree 1, 5000 3-D data, k=64, Matern-3/2, %d points on this GPU" % m,
            "metric": "kriged points/s", "value": round(m / dt, 1), "unit": "points/s",
            "roofline": {"bound": "fp64-valu", "achieved": round(flop_pt * m / dt / 1e12, 3), "peak": FP64_PEAK, "unit": "TFLOP/s",
                         "frac": round(flop_pt * m / dt / 1e12 / FP64_PEAK, 4)},
            "kernel_ms": {"knn": round(knn[0], 2), "krig_local": round(loc[0], 2)}, "missing": int(st.sum().item()),
            "parity_max_abs_err_first_%d" % ns: err,
            "cpu_baseline": {"value": round(ns / cdt, 1), "unit": "points/s", "cores": 1, "kind": "port",
                             "sample": "oracle.kriging.approxsolve (search + fit + predict per point as krig.jl:205-228), %d points in %.1f s" % (ns, cdt)}}


def cfg4_full(a, gss, _lib):
    """configs[4] at its FULL size on one GPU: UK degree 1, 5 000 3-D data, k = 64, Matern-3/2, 10^7 domain points in ONE
    gss_krig_predict_knn call (the library walks them in chunks of 2^20: search, then systems, per chunk)."""
    from gss.engine import KrigHandle, UK
    from oracle import kriging as K
    from oracle.variogram import Variogram
    n = 5000
    m = 1_000_000 if a.quick else 10_000_000
    x = np.random.default_rng(6).uniform(0, 100, (n, 3))
    z = 1.0 + 0.03 * x[:, 0] - 0.02 * x[:, 1] + 0.01 * x[:, 2] + np.random.default_rng(60).normal(size=n)
    x0d = torch.rand((m, 3), dtype=torch.float64, device="cuda", generator=torch.Generator("cuda").manual_seed(7)) * 100.0
    h = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5), UK, x, z, degree=1, factor=False)
    h.predict_knn(x0d[:50000], 64)
    sync()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    mu, var, st = h.predict_knn(x0d, 64)
    sync()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    knn = _lib.profile_read("knn")
    loc = _lib.profile_read("krig_local")
    flop_pt = 68 ** 3 / 3 + 2 * 68 ** 2 + 8 * n
    ns = 200
    sel = torch.linspace(0, m - 1, ns, device="cuda").long()          # a sample across the whole call, last chunk included
    rmu, rvar, _ = K.approxsolve(K.UK, Variogram("matern", range=30.0, nu=1.5), x, z, x0d[sel].cpu().numpy(), 64, degree=1)
    err = float(np.max(np.abs(mu[sel].cpu().numpy() - rmu)))
    return {"config": "configs[4] at full size: UK degree 1, 5000 3-D data, k=64, Matern-3/2, %d points in one call on one GPU" % m,
            "metric": "kriged points/s", "value": round(m / dt, 1), "unit": "points/s", "seconds": round(dt, 4),
            "roofline": {"bound": "fp64-valu", "achieved": round(flop_pt * m / dt / 1e12, 3), "peak": FP64_PEAK, "unit": "TFLOP/s",
                         "frac": round(flop_pt * m / dt / 1e12 / FP64_PEAK, 4)},
            "kernel_ms": {"knn": round(knn[0], 2), "krig_local": round(loc[0], 2), "launches": [knn[1], loc[1]]},
            "missing": int(st.sum().item()), "parity_max_abs_err_%d_points_across_the_call" % ns: err}


def cfg_api(a, gss, _lib):
    """The two headline workloads through the FRONT-END the reference's users call -- `solve(problem, solver)`
    (krig.jl:130, fft.jl:62/145 under GeoStatsBase's loop) -- beside the handle-level numbers of bench.py, so that the
    cost of the API layer (problem objects, missing-value handling, host tables in and out) is a number:
    configs[1] `solve(EstimationProblem(1 000 data, PointSet of 10^6), KrigingSolver(z => (variogram = Matern-3/2)))`
    and configs[2] `solve(SimulationProblem(CartesianGrid(512^3), z, 8), FFTGS(...))`, results in host memory as the
    reference returns them."""
    from gss.engine import KrigHandle, FFTGSHandle, OK
    n, m = 1000, (100_000 if a.quick else 1_000_000)
    x = np.random.default_rng(2).uniform(0.0, 100.0, (n, 3))
    z = np.random.default_rng(1002).normal(size=n)
    x0 = np.random.default_rng(3).uniform(0.0, 100.0, (m, 3))
    vg = gss.MaternVariogram(range=30.0, order=1.5)
    prob = gss.EstimationProblem(gss.georef({"z": z}, x), gss.PointSet(x0), "z")
    solver = gss.KrigingSolver(("z", dict(variogram=vg)))
    sol = gss.solve(prob, solver)
    t0 = time.perf_counter()
    for _ in range(3):
        sol = gss.solve(prob, solver)
    dt_api = (time.perf_counter() - t0) / 3

    def step():
        h = KrigHandle(vg, OK, x, z)
        out = h.predict_global(x0)
        sync()
        h.close()
        return out

    step()
    t0 = time.perf_counter()
    for _ in range(3):
        mu, var, st = step()
    dt_handle = (time.perf_counter() - t0) / 3
    same = bool(np.array_equal(np.asarray(sol["z"]), mu))
    krig = {"workload": "configs[1]: solve(EstimationProblem(1000 3-D data, PointSet(%d)), KrigingSolver(Matern-3/2))" % m,
            "solve_ms": round(dt_api * 1e3, 2), "handle_level_ms_same_host_arrays": round(dt_handle * 1e3, 2),
            "points_per_s_through_solve": round(m / dt_api, 1), "api_overhead_ms": round((dt_api - dt_handle) * 1e3, 2),
            "same_estimates": same}
    e = 128 if a.quick else 512
    R = 8
    N = e ** 3
    grid = gss.CartesianGrid((e, e, e))
    fsolver = gss.FFTGS(("z", dict(variogram=gss.ExponentialVariogram(range=50.0 * e / 512))), rng=4)
    fprob = gss.SimulationProblem(grid, ("z", float), R)
    ens = gss.solve(fprob, fsolver)
    del ens
    t0 = time.perf_counter()
    ens = gss.solve(fprob, fsolver)
    dt_fapi = time.perf_counter() - t0
    r0 = np.array(ens["z"][0][:1000])
    del ens
    f = FFTGSHandle(gss.ExponentialVariogram(range=50.0 * e / 512), (e, e, e))
    out = np.empty((R, N))
    out.fill(0.0)
    f.realize(4, 0, R, out=out)
    t0 = time.perf_counter()
    f.realize(4, 0, R, out=out)
    dt_fh = time.perf_counter() - t0
    f.close()
    fft = {"workload": "configs[2]: solve(SimulationProblem(CartesianGrid(%d^3), z, %d), FFTGS(exponential)), host vectors out" % (e, R),
           "solve_s": round(dt_fapi, 3), "realisations_per_s_through_solve": round(R / dt_fapi, 2),
           "handle_level_s_same_destination_kind": round(dt_fh, 3), "handle_level_realisations_per_s": round(R / dt_fh, 2),
           "same_first_values": bool(np.array_equal(r0, out[0][:1000]))}
    # estimation on a large Cartesian grid through solve: the domain points are formed in HBM, the table comes back through
    # page-locked memory (configs[4]-like problem at 10^7 cells: 5 000 3-D data, 16 neighbours)
    ge = 64 if a.quick else 216
    rngg = np.random.default_rng(3)
    gdata = gss.georef({"z": rngg.normal(size=5000)}, rngg.uniform(0, ge, (5000, 3)))
    ggrid = gss.CartesianGrid(ge, ge, ge)
    grows = {}
    for name, solver in (("kriging_k16", gss.KrigingSolver(("z", dict(variogram=gss.MaternVariogram(range=30.0, order=1.5),
                                                                       maxneighbors=16)))),
                         ("idw_k16", gss.IDWSolver(("z", dict(maxneighbors=16))))):
        gprob = gss.EstimationProblem(gdata, ggrid, "z")
        gss.solve(gprob, solver)
        sync(); t0 = time.perf_counter()
        gsol = gss.solve(gprob, solver)
        sync(); dtg = time.perf_counter() - t0
        grows[name] = {"solve_s": round(dtg, 4), "cells_per_s": round(ge ** 3 / dtg, 1), "finite": bool(np.isfinite(gsol["z"]).all())}
    grid_row = {"workload": "solve(EstimationProblem(5000 3-D data, CartesianGrid(%d^3)), .) with 16 neighbours, host table out" % ge,
                **grows}
    return {"config": "front-end rows: the headline workloads through gss.solve(problem, solver)", "metric": "kriged points/s through solve",
            "value": krig["points_per_s_through_solve"], "unit": "points/s", "kriging": krig, "fftgs": fft, "grid_domain": grid_row}


def cfg_fftgs_generic(a, gss, _lib):
    """FFTGS on grids outside the power-of-two 3-D pipeline -- the reference's own test grids are 100 x 100
    (test/simulation/fft.jl:4,11,26) -- on the library's generic Stockham passes (sizes 2^a 3^b 5^c 7^d, fftgs_generic.h)
    and, beside it, on the rocFFT pipeline of the same library (GSS_FFTGS_PATH=rocfft, read at handle creation)."""
    from gss.engine import FFTGSHandle
    rows = []
    shapes = ((100, 100), (1000, 1000), (200, 200, 200), (300, 300, 300)) if a.quick else \
        ((1000,), (100, 100), (500, 500), (1000, 1000), (4096, 1024), (2048, 2048), (4096, 4096), (3000, 3000), (64, 64, 64),
         (100, 100, 100), (200, 200, 200), (300, 300, 300), (500, 500, 500))
    for dims in shapes:
        N = int(np.prod(dims))
        R = 64 if N <= 4e6 else (16 if N < 5e7 else 4)      # (small grids: realisations share launches, up to 64)
        res = {}
        for path in ("generic", "rocfft"):
            if path == "rocfft":
                os.environ["GSS_FFTGS_PATH"] = "rocfft"
            else:
                os.environ.pop("GSS_FFTGS_PATH", None)
            vg = gss.ExponentialVariogram(range=dims[0] / 10.0)
            FFTGSHandle(vg, dims).close()                      # plans / kernels of this size exist
            sync(); t0 = time.perf_counter()
            h = FFTGSHandle(vg, dims)
            sync(); t1 = time.perf_counter()
            out = torch.empty((R, N), dtype=torch.float64, device="cuda")
            h.realize(1, 0, R, out=out)
            sync(); t2 = time.perf_counter()
            h.realize(1, 0, R, out=out)
            sync(); t3 = time.perf_counter()
            res[path] = dict(ms=(t3 - t2) / R * 1e3, create_ms=(t1 - t0) * 1e3, first=out[0].clone())
            h.close()
            del out
        os.environ.pop("GSS_FFTGS_PATH", None)
        g, r = res["generic"], res["rocfft"]
        rows.append({"grid": "x".join(map(str, dims)), "ms_per_realisation": round(g["ms"], 4),
                     "rocfft_pipeline_ms": round(r["ms"], 4), "speedup": round(r["ms"] / g["ms"], 2),
                     "create_ms_warm": round(g["create_ms"], 2), "rocfft_create_ms_warm": round(r["create_ms"], 2),
                     "max_abs_difference_between_the_two": float((g["first"] - r["first"]).abs().max()),
                     "roofline": {"bound": "hbm", "achieved": round(32.0 * N / (g["ms"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK,
                                  "unit": "GB/s", "frac": round(32.0 * N / (g["ms"] * 1e-3) / 1e9 / HBM_PEAK, 4),
                                  "algorithmic_bytes": 32.0 * N}})
    return {"config": "FFTGS on 2-D grids and on sizes 2^a 3^b 5^c 7^d: the library's generic passes against its rocFFT pipeline",
            "metric": "ms per realisation", "rows": rows}


def cfg_hugek(a, gss, _lib):
    """Beyond 256 neighbours: 257 .. 768 on `krig_local_slab_kernel` (round 4: the tile algorithm with the triangle in a
    per-workgroup slab of global memory), 769 .. 4 096 on the functional `krig_local_big_kernel` (one workgroup per point,
    unblocked root-free Cholesky through an HBM slab), each with its roofline: k = 300, 512, 768, 1 000."""
    return cfg_bigk(a, gss, _lib, ks=(300, 512, 768, 1000))


def cfg_bigk(a, gss, _lib, ks=(96, 128, 256)):
    """Moving neighbourhoods with more than 64 neighbours (krig.jl:201-210, ui.jl:16-23 accept any count): UK degree 1,
    5 000 3-D data, Matern-3/2, k = 96 / 128 / 256.  One JSON object with a row per k."""
    from gss.engine import KrigHandle, UK
    from oracle import kriging as K
    from oracle.variogram import Variogram
    n = 5000
    x = np.random.default_rng(6).uniform(0, 100, (n, 3))
    z = 1.0 + 0.03 * x[:, 0] - 0.02 * x[:, 1] + 0.01 * x[:, 2] + np.random.default_rng(60).normal(size=n)
    h = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5), UK, x, z, degree=1, factor=False)
    rows = []
    for k in ks:
        m = (20_000 if a.quick else 200_000) if k <= 128 else ((10_000 if a.quick else 100_000) if k <= 256 else
                                                               ((2_000 if a.quick else 20_000) if k <= 768 else 2_000))
        x0 = np.random.default_rng(7).uniform(0, 100, (m, 3))
        x0d = torch.as_tensor(x0, device="cuda")
        h.predict_knn(x0d[:2000], k)
        sync()
        _lib.profile_reset(); _lib.profile_enable(True)
        t0 = time.perf_counter()
        mu, var, st = h.predict_knn(x0d, k)
        sync()
        dt = time.perf_counter() - t0
        _lib.profile_enable(False)
        knn, loc = _lib.profile_read("knn"), _lib.profile_read("krig_local")
        N1 = k + 4
        flop_pt = N1 ** 3 / 3 + 2 * N1 ** 2
        ns = 40
        t1 = time.perf_counter()
        rmu, rvar, _ = K.approxsolve(K.UK, Variogram("matern", range=30.0, nu=1.5), x, z, x0[:ns], k, degree=1)
        cdt = time.perf_counter() - t1
        err = float(np.max(np.abs(mu[:ns].cpu().numpy() - rmu)))
        rows.append({"k": k, "points": m, "value": round(m / dt, 1), "unit": "points/s",
                     "kernel_ms": {"knn": round(knn[0], 2), "krig_local": round(loc[0], 2)},
                     "roofline": {"bound": "mfma", "kernel": "krig_local (per-point systems)",
                                  "achieved": round(flop_pt * m / (loc[0] * 1e-3) / 1e12, 3) if loc[0] else None,
                                  "peak": FP64_PEAK, "unit": "TFLOP/s",
                                  "frac": round(flop_pt * m / (loc[0] * 1e-3) / 1e12 / FP64_PEAK, 4) if loc[0] else None,
                                  "flop_per_point": flop_pt},
                     "parity_max_abs_err_first_%d" % ns: err,
                     "cpu_baseline": {"value": round(ns / cdt, 1), "unit": "points/s", "cores": 1, "kind": "port",
                                      "sample": "oracle.kriging.approxsolve, %d points in %.2f s" % (ns, cdt)}})
    return {"config": "moving neighbourhood beyond 64 neighbours: UK degree 1, 5000 3-D data, Matern-3/2",
            "metric": "kriged points/s", "rows": rows}


def cfg_lu(a, gss, _lib):
    """LUGS with `factorization = lu` (lu.jl:70,107) at configs[3] size beside the Cholesky preprocess."""
    from gss.engine import LUGSHandle
    from oracle import fftgs as O
    g = 64 if a.quick else 128
    nd = g * g // 4
    cent = O.grid_centroids((g, g))
    N = g * g
    dlocs = np.sort(np.random.default_rng(5).permutation(N)[:nd])
    z1 = np.random.default_rng(50).normal(size=nd)
    vg = gss.SphericalVariogram(range=20.0)
    ns = N - nd
    out = {}
    for fact in ("cholesky", "lu"):
        LUGSHandle(vg, cent, dlocs, z1, factorization=fact).close()
        sync()
        t0 = time.perf_counter()
        h = LUGSHandle(vg, cent, dlocs, z1, factorization=fact)
        sync()
        out[fact] = time.perf_counter() - t0
        h.close()
    # LU: 2/3 n^3 per factorisation (nd and ns), the two triangular solves and the Schur product as in the Cholesky path
    flops = 2 * nd ** 3 / 3 + 2 * nd * nd * ns + 2 * nd * ns * ns + 2 * ns ** 3 / 3
    import scipy.linalg as sla
    from oracle.variogram import Variogram, cov_pairwise
    nb = 2048
    C = cov_pairwise(Variogram("spherical", range=20.0), cent[:nb])
    t1 = time.perf_counter()
    sla.lu(C)
    cdt = time.perf_counter() - t1
    return {"config": "LUGS factorization = lu, %dx%d grid, %d conditioning cells, ns=%d" % (g, g, nd, ns),
            "metric": "preprocess seconds", "value": round(out["lu"], 4), "unit": "s", "cholesky_s": round(out["cholesky"], 4),
            "ratio_to_cholesky": round(out["lu"] / out["cholesky"], 2),
            "roofline": {"bound": "mfma", "achieved": round(flops / out["lu"] / 1e12, 3), "peak": FP64_PEAK, "unit": "TFLOP/s",
                         "frac": round(flops / out["lu"] / 1e12 / FP64_PEAK, 4)},
            "cpu_baseline": {"value": round(cdt * (ns / nb) ** 3, 2), "unit": "s (getrf of the ns x ns block alone)", "cores": os.cpu_count(),
                             "kind": "port", "sample": "scipy.linalg.lu on %d x %d in %.2f s, scaled by n^3 to ns" % (nb, nb, cdt)}}


def _est(a, gss, _lib, which):
    """Section 8f.3 rows: IDW / LWR with k = 16 neighbours, 50 000 3-D samples, 1.25e6 estimation points."""
    from gss.engine import HipEngine
    n, k = 50_000, 16
    m = 100_000 if a.quick else 1_250_000
    x = np.random.default_rng(16).uniform(0, 100, (n, 3))
    z = np.sin(x[:, 0] / 9.0) + 0.01 * x[:, 1] + np.random.default_rng(61).normal(size=n) * 0.1
    x0 = np.random.default_rng(17).uniform(0, 100, (m, 3))
    xd, zd, x0d = (torch.as_tensor(v, device="cuda") for v in (x, z, x0))
    run = (lambda q: HipEngine.idw(xd, zd, q, k, 1, 2.0)) if which == "idw" else (lambda q: HipEngine.lwr(xd, zd, q, k))
    run(x0d[:50000])
    sync()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    mu, aux, st = run(x0d)
    sync()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    knn = _lib.profile_read("knn")
    est = _lib.profile_read(which)
    from oracle import idw_lwr as E
    ns = 400
    t1 = time.perf_counter()
    r = E.idw(x, z, x0[:ns], k, exponent=2.0) if which == "idw" else E.lwr(x, z, x0[:ns], k)
    cdt = time.perf_counter() - t1
    err = float(np.max(np.abs(mu[:ns].cpu().numpy() - r[0])))
    # algorithmic HBM bytes per point of the estimator kernel: k neighbour indices (4 B) + k gathered samples
    # (3 coordinates + value, 32 B) + the point (24 B) + 17 B of outputs
    bytes_pt = k * 36 + 41
    return {"config": "8f.3 %s k=%d, %d 3-D samples, %d points on this GPU" % (which.upper(), k, n, m),
            "metric": "estimated points/s", "value": round(m / dt, 1), "unit": "points/s",
            "roofline": {"bound": "hbm", "kernel": which, "achieved": round(bytes_pt * m / (est[0] * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK, "unit": "GB/s", "frac": round(bytes_pt * m / (est[0] * 1e-3) / 1e9 / HBM_PEAK, 4)},
            "kernel_ms": {"knn": round(knn[0], 2), which: round(est[0], 2)}, "missing": int(st.sum().item()),
            "parity_max_abs_err_first_%d" % ns: err,
            "cpu_baseline": {"value": round(ns / cdt, 1), "unit": "points/s", "cores": 1, "kind": "port",
                             "sample": "oracle.idw_lwr.%s, %d points in %.1f s" % (which, ns, cdt)}}


def cfg_est_all(a, gss, _lib):
    """Section 8f.3, the reference's DEFAULT configuration (maxneighbors = nothing, idw.jl:93): every sample is a
    neighbour of every point -- no search, est_all_kernel sweeps the samples from LDS.  5 000 samples, 1.25e6 points."""
    from gss.engine import HipEngine
    n = 5000
    m = 100_000 if a.quick else 1_250_000
    x = np.random.default_rng(16).uniform(0, 100, (n, 3))
    z = np.sin(x[:, 0] / 9.0) + 0.01 * x[:, 1]
    x0 = np.random.default_rng(17).uniform(0, 100, (m, 3))
    xd, zd, x0d = (torch.as_tensor(v, device="cuda") for v in (x, z, x0))
    out = {}
    for which, run in (("idw", lambda q: HipEngine.idw(xd, zd, q, n, 1, 1.0)), ("lwr", lambda q: HipEngine.lwr(xd, zd, q, n))):
        run(x0d[:20000])
        sync()
        _lib.profile_reset(); _lib.profile_enable(True)
        t0 = time.perf_counter()
        mu, aux, st = run(x0d)
        sync()
        dt = time.perf_counter() - t0
        _lib.profile_enable(False)
        out[which] = {"points_per_s": round(m / dt, 1), "kernel_ms": round(_lib.profile_read(which)[0], 2),
                      "pairs_per_s": round(n * m / dt, 1)}
    from oracle import idw_lwr as E
    ns = 200
    r = E.idw(x, z, x0[:ns])
    err = float(np.max(np.abs(mu[:ns].cpu().numpy() - E.lwr(x, z, x0[:ns])[0])))
    return {"config": "8f.3 IDW / LWR with all %d samples per point (reference default), %d points" % (n, m),
            "metric": "estimated points/s", "value": out["idw"]["points_per_s"], "unit": "points/s", "idw": out["idw"],
            "lwr": out["lwr"], "lwr_parity_max_abs_err_first_%d" % ns: err}


def cfg_idw(a, gss, _lib):
    return _est(a, gss, _lib, "idw")


def cfg_lwr(a, gss, _lib):
    return _est(a, gss, _lib, "lwr")


def cfg_sgs(a, gss, _lib):
    """Section 8f.4 row: SGS on a 512 x 512 grid, spherical range 35, k = 16, ball 30, 200 conditioning cells,
    1024 realisations per sweep (the recursion runs level by level over its dependency graph; the row-by-row order
    of this row has ~3 100 levels of ~85 nodes, which `sgs_level_team_kernel` sweeps in one launch: sgs.hip)."""
    from gss.engine import SGSHandle
    from oracle import fftgs as offt
    e = 128 if a.quick else 512
    R = a.sgs_reals   # one wave carries 64 realisations of one node of a level
    cent = offt.grid_centroids((e, e))
    N = cent.shape[0]
    rng = np.random.default_rng(5)
    dl = np.sort(rng.choice(N, 200, replace=False))
    zd = rng.normal(size=200)
    vg = gss.SphericalVariogram(range=35.0)
    SGSHandle(vg, cent[:20000], None, dl[dl < 20000], zd[dl < 20000], 0.0, 16, 1, 30.0).close()   # code objects loaded
    sync()
    _lib.profile_reset(); _lib.profile_enable(True)
    t0 = time.perf_counter()
    h = SGSHandle(vg, cent, None, dl, zd, 0.0, 16, 1, 30.0)
    sync()
    t_pre = time.perf_counter() - t0
    z = h.realize(1, 0, R, device=True)
    sync()
    _lib.profile_reset()
    t0 = time.perf_counter()
    h.realize(1, 0, R, out=z)              # same buffer: the 2 GiB allocation is not part of the sweep
    sync()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    sweep = _lib.profile_read("sgs_sweep")
    zc = z.cpu().numpy()
    honoured = bool(np.array_equal(zc[:, dl], np.tile(zd, (R, 1))))
    h.close()
    # CPU baseline: the oracle's per-realisation path loop on a 64 x 64 grid (cost per node grows with N)
    from oracle import sgs as S
    from oracle.variogram import Variogram
    ce = 64
    cc = offt.grid_centroids((ce, ce))
    t1 = time.perf_counter()
    S.realize(Variogram("spherical", range=35.0), 0.0, cc, None, np.array([100, 2000]), np.array([1.0, 0.0]), 1, 0, 1,
              maxneighbors=16, radius=30.0)
    cdt = time.perf_counter() - t1
    # stage B bytes that have to cross HBM per node and realisation: the normal read and the value written (the k
    # gathered values were written a few levels earlier and are served on chip); the sweep is bound by the latency of
    # its ~3 100 dependent levels, which is what the small fraction says
    bytes_nr = 2 * 8
    return {"config": "8f.4 SGS %dx%d grid, spherical range 35, k=16, ball 30, 200 data, %d realisations" % (e, e, R),
            "metric": "simulated cells/s (all realisations)", "value": round(N * R / dt, 1), "unit": "cells/s",
            "preprocess_s": round(t_pre, 3), "realize_s": round(dt, 4),
            "roofline": {"bound": "hbm (normals in, values out); in fact the latency of ~3 100 dependent levels",
                         "kernel": "sgs_level_team_kernel (one launch; GSS_SGS_TEAM=0: sgs_level_sweep_kernel, one launch per level)",
                         "achieved": round(bytes_nr * N * R / (sweep[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": round(bytes_nr * N * R / (sweep[0] * 1e-3) / 1e9 / HBM_PEAK, 5)},
            "kernel_ms": {"sgs_sweep": round(sweep[0], 2)}, "us_per_path_node": round(sweep[0] * 1e3 / N, 3),
            "hard_data_honoured": honoured,
            "field_std": round(float(zc.std()), 4),
            "cpu_baseline": {"value": round(ce * ce / cdt, 1), "unit": "cells/s", "cores": 1, "kind": "port",
                             "sample": "oracle.sgs.realize, one realisation of a %dx%d grid in %.1f s" % (ce, ce, cdt)}}


def cfg1_host(a, gss, _lib):
    """configs[1] with host buffers on both sides of the call (the second number, PCIe included):
    coordinates staged host -> device, mean / variance / status copied back, per step as in bench.py."""
    from gss.engine import KrigHandle, OK
    n, m = 1000, (100_000 if a.quick else 1_000_000)
    x = np.random.default_rng(2).uniform(0.0, 100.0, (n, 3))
    z = np.random.default_rng(1002).normal(size=n)
    x0 = np.random.default_rng(3).uniform(0.0, 100.0, (m, 3))
    vg = gss.MaternVariogram(range=30.0, order=1.5)

    def step(dom):
        h = KrigHandle(vg, OK, x, z)
        out = h.predict_global(dom)
        sync()
        h.close()
        return out

    step(x0)
    t0 = time.perf_counter()
    for _ in range(3):
        mu, var, st = step(x0)
    dt_host = (time.perf_counter() - t0) / 3
    x0d = torch.as_tensor(x0, device="cuda")
    step(x0d)
    t0 = time.perf_counter()
    for _ in range(3):
        step(x0d)
    dt_dev = (time.perf_counter() - t0) / 3
    return {"config": "configs[1] OK, 1000 3-D data -> %d points, host arrays in / out (pageable memory)" % m,
            "metric": "kriged points/s including PCIe", "value": round(m / dt_host, 1), "unit": "points/s",
            "ms_per_step_host_buffers": round(dt_host * 1e3, 2), "ms_per_step_device_buffers": round(dt_dev * 1e3, 2),
            "bytes_over_pcie_per_point": 24 + 17, "finite": bool(np.isfinite(mu).all())}


def cfg2_host(a, gss, _lib):
    """configs[2] with the realisations delivered to host memory (the second number; the reference returns
    host vectors, fft.jl:173,197): one gss_fftgs_realize call for `--host-reals` realisations of 512^3 cells into a
    page-locked and into a pageable destination.  The library stages three chunks (3 GiB) whatever the count."""
    from gss.engine import FFTGSHandle
    e = 256 if a.quick else 512
    R = a.host_reals
    N = e ** 3
    f = FFTGSHandle(gss.ExponentialVariogram(range=50.0 * e / 512), (e, e, e))
    dev = torch.empty((2, N), dtype=torch.float64, device="cuda")
    f.realize(4, 0, 2, out=dev)
    sync()
    host = torch.empty((R, N), dtype=torch.float64, pin_memory=True)
    host.zero_()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from copy_rate import d2h_copy_rate_gbs, copy_rate_fraction
    copy_gbs = d2h_copy_rate_gbs(torch, host, dev)                    # warm, best of 4 copies of 2 GiB
    f.realize(4, 0, 1, out=host[:1])
    t0 = time.perf_counter()
    f.realize(4, 0, R, out=host)
    dt = time.perf_counter() - t0
    ring, chunks = _lib.stat("out_ring_bytes"), _lib.stat("out_chunks")
    same = bool(torch.equal(host[1].cuda(), dev[1])) and bool(torch.equal(host[R - 1].cuda(), f.realize(4, R - 1, 1, out=dev[:1])[0]))
    Rp = min(R, 4)
    pg = np.empty((Rp, N))
    pg.fill(0.0)
    t0 = time.perf_counter()
    f.realize(4, 0, Rp, out=pg)
    dtp = time.perf_counter() - t0
    samep = bool(np.array_equal(pg[1], host[1].numpy()))
    f.realize(4, 0, 2, out=dev)
    sync()
    copy_gbs, frac, fnote = copy_rate_fraction(torch, R * N * 8 / dt / 1e9, copy_gbs, host, dev)
    f.close()
    return {"config": "configs[2] FFTGS %d^3, %d realisations in ONE call, results in host memory" % (e, R),
            "metric": "FFTGS realisations/s including D2H", "value": round(R / dt, 2), "unit": "realisations/s",
            "achieved_GBs": round(R * N * 8 / dt / 1e9, 2), "pcie_copy_rate_GBs": round(copy_gbs, 2),
            "frac_of_copy_rate": frac, "copy_rate_note": fnote, "hbm_staged_bytes": ring, "chunks": chunks,
            "bit_identical_to_device_path": same,
            "pageable": {"realisations": Rp, "value": round(Rp / dtp, 2), "GBs": round(Rp * N * 8 / dtp / 1e9, 2),
                         "bit_identical": samep}}


def cfg_cond_fftgs(a, gss, _lib):
    """Section 8f.1 row: conditional FFTGS through the solver API (fft.jl:106-135,176-192): e^3 grid, exponential
    range 20, 1 000 conditioning points, global kriging of the data and of every unconditional realisation."""
    e = 64 if a.quick else 128
    R = 16
    rng = np.random.default_rng(9)
    grid = gss.CartesianGrid((e, e, e))
    N = e ** 3
    xd = rng.uniform(0.0, float(e), (1000, 3))
    zd = rng.normal(size=1000)
    prob = gss.SimulationProblem(gss.georef({"z": zd}, xd), grid, "z", R)
    solver = gss.FFTGS(("z", dict(variogram=gss.ExponentialVariogram(range=20.0))), rng=3)
    solver.solve(prob)                     # warm-up (rocFFT-free plans, kernels, allocator)
    sync()
    t0 = time.perf_counter()
    ens = solver.solve(prob)
    sync()
    dt = time.perf_counter() - t0
    z0 = ens["z"][0]
    # honouring: cells that hold a datum reproduce the value kriged there (exact at a coincident centroid only)
    return {"config": "8f.1 conditional FFTGS %d^3 grid, exponential range 20, 1000 data, %d realisations, solver API "
                      "(host arrays in / out)" % (e, R),
            "metric": "conditional realisations/s", "value": round(R / dt, 2), "unit": "realisations/s",
            "cells_per_s": round(N * R / dt, 1), "solve_s": round(dt, 3), "field_std": round(float(np.std(z0)), 4)}


def cfg_sgs_bigk(a, gss, _lib):
    """SGS with more than 64 neighbours (seq.jl:91-98 accepts any maxneighbors): the functional path of the library --
    search in passes of 64, `sgs_weights_big_kernel` (one workgroup per node), `sgs_level_sweep_kernel` / the walk along
    the path.  128 x 128 grid, spherical range 35, 100 neighbours, 64 realisations."""
    from gss.engine import SGSHandle
    from oracle import fftgs as offt
    e, k, R = (64 if a.quick else 128), 100, 64
    cent = offt.grid_centroids((e, e))
    N = cent.shape[0]
    rng = np.random.default_rng(5)
    dl = np.sort(rng.choice(N, 50, replace=False))
    zd = rng.normal(size=50)
    vg = gss.SphericalVariogram(range=35.0)
    SGSHandle(vg, cent[:2000], None, dl[dl < 2000], zd[dl < 2000], 0.0, k, 1, 30.0).close()
    sync()
    t0 = time.perf_counter()
    h = SGSHandle(vg, cent, None, dl, zd, 0.0, k, 1, 30.0)
    sync()
    t_pre = time.perf_counter() - t0
    z = h.realize(1, 0, R, device=True)
    sync()
    t0 = time.perf_counter()
    h.realize(1, 0, R, out=z)
    sync()
    dt = time.perf_counter() - t0
    zc = z.cpu().numpy()
    honoured = bool(np.array_equal(zc[:, dl], np.tile(zd, (R, 1))))
    h.close()
    from oracle import sgs as S
    from oracle.variogram import Variogram
    ce = 32
    cc = offt.grid_centroids((ce, ce))
    t1 = time.perf_counter()
    S.realize(Variogram("spherical", range=35.0), 0.0, cc, None, np.array([100, 700]), np.array([1.0, 0.0]), 1, 0, 1,
              maxneighbors=k, radius=30.0)
    cdt = time.perf_counter() - t1
    return {"config": "8f.4 SGS %dx%d grid, spherical range 35, k=%d (functional path beyond 64 neighbours), ball 30, 50 data, "
                      "%d realisations" % (e, e, k, R),
            "metric": "simulated cells/s (all realisations)", "value": round(N * R / dt, 1), "unit": "cells/s",
            "preprocess_s": round(t_pre, 3), "realize_s": round(dt, 4), "hard_data_honoured": honoured,
            "field_std": round(float(zc.std()), 4),
            "cpu_baseline": {"value": round(ce * ce / cdt, 1), "unit": "cells/s", "cores": 1, "kind": "port",
                             "sample": "oracle.sgs.realize, one realisation of a %dx%d grid in %.1f s" % (ce, ce, cdt)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--sgs-reals", type=int, default=1024, help="realisations of the SGS row (multiple of 64)")
    ap.add_argument("--host-reals", type=int, default=24, help="realisations of the 2h row (1 GiB of host memory each)")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    import gss
    from gss import _lib
    fns = {"1h": cfg1_host, "2": cfg2_fftgs, "2h": cfg2_host, "3": cfg3_lugs, "4": cfg4_local, "4full": cfg4_full, "api": cfg_api, "hugek": cfg_hugek, "fftgs_gen": cfg_fftgs_generic, "idw": cfg_idw, "lwr": cfg_lwr, "sgs": cfg_sgs, "est_all": cfg_est_all, "cond_fftgs": cfg_cond_fftgs, "bigk": cfg_bigk, "lu": cfg_lu, "sgs_bigk": cfg_sgs_bigk}
    for c in a.configs.split(","):
        print(json.dumps(fns[c](a, gss, _lib)), flush=True)


if __name__ == "__main__":
    main()
