#!/usr/bin/env python3
"""bench.py -- headline benchmark of the kriging / FFTGS hot path on MI355X.

`python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no launcher environment the script starts N
fresh ranks itself (`python -m torch.distributed.run ... bench.py`, one rank per GPU over RCCL) BEFORE anything
touches the GPU, waits for them and exits with their status; under the driver's own `torch.distributed.run` it
checks that the launcher's world size equals N.  Rank 0 prints ONE JSON line.

A "step" is one whole pass of BASELINE.json configs[1] on each GPU: Ordinary Kriging of 1 000 scattered 3-D samples
onto 10^6 domain points with a Matern-3/2 variogram (global neighbourhood) -- fit (device factorisation of the
1001^2 system) + predict, with every input already resident in HBM when the timed region starts.  Multi-GPU is weak
scaling: each rank owns its own block of 10^6 domain points (independent given the factor, which every rank
recomputes because that is cheaper than a broadcast at n = 1000; `--factor-broadcast` exercises the RCCL broadcast
path instead).

Extra keys of the line:
  roofline      dominant kernel of the step (FP64-MFMA quadratic form), HIP-event timed inside this process
  cpu_baseline  oracle C restatement, single thread like the reference's loop krig.jl:180, bounded sample (N = 1 only)
  fftgs         second headline metric (BASELINE.json: "+ FFTGS 512^3 realisations/sec at 1/2/4/8"): configs[2],
                realisations sharded over the ranks ((seed, r)-keyed noise), spectrum computed on rank 0 and broadcast
  lugs          configs[3]: rank 0 runs the preprocess (lu.jl:76-169), broadcasts (L22, d2) over RCCL, every rank
                realises its share (lu.jl:171-224)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak (public spec)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ndata", type=int, default=1000)
    ap.add_argument("--npoints", type=int, default=1_000_000, help="domain points per GPU per step")
    ap.add_argument("--factor-broadcast", action="store_true")
    ap.add_argument("--native-bcast", action="store_true",
                    help="replicate the preprocess states with the library's own RCCL broadcast (gss_comm_init + "
                         "gss_state_bcast) instead of torch.distributed.broadcast")
    ap.add_argument("--sync-fit", action="store_true", help="gss_krig_create waits for the fit (default: "
                    "GSS_KRIG_ASYNC_FIT, the fit beside the first assembly, as the solver front-end runs it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=30000, help="points of the CPU baseline sample")
    ap.add_argument("--fftgs", type=int, default=512, help="FFTGS grid edge (0 disables the leg)")
    ap.add_argument("--fftgs-reals", type=int, default=64, help="FFTGS realisations per GPU in the timed region")
    ap.add_argument("--fftgs-batch", type=int, default=8, help="realisations per gss_fftgs_realize call")
    ap.add_argument("--lugs", type=int, default=128, help="LUGS grid edge, a quarter of the cells carry data "
                                                         "(configs[3]: 128; 0 disables the leg)")
    ap.add_argument("--lugs-reals", type=int, default=100, help="LUGS realisations in total (sharded over the ranks)")
    return ap.parse_args()


def launch_ranks(a):
    """Start `--gpus N` ranks as children of this (GPU-free) process; a process that has initialised the GPU is
    never re-exec'd."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not launched:
        sys.exit(launch_ranks(a))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}, but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the gfx950 path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("GSS_BENCH_BACKEND", "nccl")
    if local >= ndev:
        # GSS_BENCH_BACKEND=gloo: rehearsal of the N-rank path on fewer GPUs (ranks share devices; RCCL cannot do that)
        if backend == "nccl":
            raise SystemExit(f"rank {rank}: local rank {local} but only {ndev} GPU(s) visible")
        local = local % ndev
    torch.cuda.set_device(local)
    use_dist = launched
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {a.gpus}")

    import gss
    from gss import _lib, parallel
    from gss.engine import KrigHandle, OK

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        if use_dist:
            if backend == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            else:
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MAX)
                t = c
        return float(t.item())

    # ---- synthetic inputs ---------------------------------
    n, m = a.ndata, a.npoints
    x = np.random.default_rng(2).uniform(0.0, 100.0, (n, 3))
    z = np.random.default_rng(2 + 1000).normal(size=n)
    x0 = np.random.default_rng(3 + 7919 * rank).uniform(0.0, 100.0, (m, 3))
    x0_dev = torch.as_tensor(x0, device="cuda")
    vg = gss.MaternVariogram(range=30.0, sill=1.0, nugget=0.0, order=1.5)

    def step():
        if a.factor_broadcast and use_dist:
            h = KrigHandle(vg, OK, x, z, factor=(rank == 0))
            if a.native_bcast:
                parallel.native_comm()
                h.bcast_state(0)
            else:
                parallel.broadcast_(h.factor_tensor(), 0)
                h.adopt_factor()
        else:
            h = KrigHandle(vg, OK, x, z, async_fit=not a.sync_fit)   # as solve() does: the fit beside the assembly
        out = h.predict_global(x0_dev)
        return h, out

    for _ in range(a.warmup):
        h, out = step()
        torch.cuda.synchronize()
        h.close()

    _lib.profile_reset()
    _lib.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    keep = None
    for _ in range(a.steps):
        h, out = step()
        if keep is not None:
            keep[0].close()
        keep = (h, out)
    barrier()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    q_ms, q_n = _lib.profile_read("krig_quadform")
    r_ms, r_n = _lib.profile_read("krig_rhs")
    mu, var, st = keep[1]
    keep[0].close()

    dt = max_over_ranks(dt)
    value = world * m * a.steps / dt

    # ---- roofline of the dominant kernel (quadratic form: (n+nc)^2 flop per point) ----------
    N1 = n + 1
    flops_per_launch = float(N1) * N1 * (m * a.steps / max(q_n, 1))
    q_avg_ms = q_ms / max(q_n, 1)
    achieved = flops_per_launch / (q_avg_ms * 1e-3) / 1e12 if q_n else 0.0
    roofline = {"kernel": "krig_quadform_kernel", "bound": "mfma", "achieved": round(achieved, 3),
                "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                "traffic": None, "avg_launch_ms": round(q_avg_ms, 4), "launches": q_n,
                "assembly": {"kernel": "krig_rhs_kernel", "bound": "hbm",
                             "achieved": round(8.0 * n * m * a.steps / (r_ms * 1e-3) / 1e9, 1) if r_n else 0.0,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": round(r_ms / max(r_n, 1), 4),
                             "launches": r_n}}

    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process, so the value is
    # the one measured for this exact configuration with tools/pmc_krig.sh + tools/pmc_traffic.py (separate --pmc
    # passes, gfx950 FETCH_SIZE correction) and committed under profiles/.
    tfile = _latest_profile("krig_cfg2_pmc_traffic.json")
    if n == 1000 and m == 1_000_000 and tfile:
        tj = json.load(open(tfile))
        roofline["traffic"] = tj["hbm_bytes_per_step"] / max(q_n / a.steps, 1)
        roofline["traffic_unit"] = "B per launch (rocprofv3 PMC FETCH_SIZE*2+WRITE_SIZE, profiles/%s)" % os.path.basename(tfile)
        roofline["algorithmic_bytes"] = 8.0 * N1 * m / max(q_n / a.steps, 1)   # R read once: 8 (n+nc) B per point

    line = {"metric": "kriged domain points/sec (OK, 1000 3-D data, Matern-3/2, global neighbourhood)",
            "value": round(value, 1), "unit": "points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: OK, %d 3-D data -> %d domain points per GPU, Matern-3/2 range 30, "
                                   "global neighbourhood; step = device fit + predict" % (n, m),
                       "points_per_gpu": m, "ndata": n, "factor": "broadcast" if a.factor_broadcast else "replicated"},
            "roofline": roofline}

    if rank == 0:
        # sanity: a handful of points against the oracle so a wrong-but-fast kernel cannot post a number
        from oracle import kriging as K
        from oracle.variogram import Variogram
        ovg = Variogram("matern", range=30.0, nu=1.5)
        sel = np.linspace(0, m - 1, 64).astype(np.int64)
        rmu, rvar = K.exactsolve(K.OK, ovg, x, z, x0[sel])
        err = max(float(np.max(np.abs(mu[sel].cpu().numpy() - rmu))), float(np.max(np.abs(var[sel].cpu().numpy() - rvar))))
        line["parity_max_abs_err_64pts"] = err
        if not os.environ.get("GSS_BENCH_EXPERIMENT"):   # set only for deliberately-wrong timing experiments
            assert err < 1e-8 and int(st.sum().item()) == 0, f"parity check failed: {err}"
        else:
            line["EXPERIMENT_RESULTS_INVALID"] = True

        if world == 1 and not a.no_cpu_baseline:
            from oracle import cbind
            ns = min(a.cpu_sample, m)
            t1 = time.perf_counter()
            cbind.krig_global(ovg, 1, x, z, x0[:ns], nthreads=1)
            cdt = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": round(ns / cdt, 1), "unit": "points/s", "cores": 1, "kind": "port",
                                    "sample": "oracle/krig_oracle.c (fit + per-point solves as krig.jl:176,180), "
                                              "first %d of the 10^6 points, %.1f s, single thread like the "
                                              "reference loop; host has %d cores" % (ns, cdt, os.cpu_count())}
            # the same sample on every host core: the C port with its OpenMP point loop, and
            # the numpy / threaded-LAPACK oracle (all right-hand sides in one solve)
            ncore = os.cpu_count() or 1
            ns2 = min(m, 8 * ns)
            t1 = time.perf_counter()
            cbind.krig_global(ovg, 1, x, z, x0[:ns2], nthreads=ncore)
            cdt2 = time.perf_counter() - t1
            t1 = time.perf_counter()
            K.exactsolve(K.OK, ovg, x, z, x0[:ns])
            cdt3 = time.perf_counter() - t1
            line["cpu_baseline"]["all_cores"] = {
                "cores": ncore, "unit": "points/s", "sample_points": ns2,
                "c_port_openmp": round(ns2 / cdt2, 1), "numpy_lapack_oracle": round(ns / cdt3, 1),
                "note": "same workload, every host core: OpenMP over the point loop of the C port (%d points, %.1f s) and "
                        "the numpy oracle with threaded LAPACK / BLAS (%d points, %.1f s); reported beside the "
                        "single-thread figure, which is the one that mirrors the reference's loop" % (ns2, cdt2, ns, cdt3)}

    del x0_dev, mu, var, st, keep, out
    ctx = dict(a=a, gss=gss, _lib=_lib, parallel=parallel, torch=torch, np=np, rank=rank, world=world,
               barrier=barrier, max_over_ranks=max_over_ranks)
    if a.fftgs > 0:
        line["fftgs"] = fftgs_leg(ctx)
    if a.lugs > 0:
        line["lugs"] = lugs_leg(ctx)

    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def _latest_profile(suffix):
    """Newest committed profiles/rNN_<suffix> (the round that last re-measured it)."""
    d = os.path.join(ROOT, "profiles")
    if not os.path.isdir(d):
        return None
    c = sorted(f for f in os.listdir(d) if f.endswith(suffix) and f[:1] == "r")
    return os.path.join(d, c[-1]) if c else None


def fftgs_leg(c):
    """Second headline metric: unconditional FFTGS realisations/s on an edge^3 grid (configs[2]).  Weak scaling:
    every rank realises `--fftgs-reals` fields (rank k owns realisations k R .. (k+1) R - 1 of the ensemble; the
    noise of realisation r depends on (seed, r) only).  fft.jl:62 runs on rank 0, the state is broadcast."""
    a, gss, _lib, parallel, torch = c["a"], c["gss"], c["_lib"], c["parallel"], c["torch"]
    rank, world = c["rank"], c["world"]
    from gss.engine import FFTGSHandle
    e, R = a.fftgs, a.fftgs_reals
    share = "native" if a.native_bcast else "broadcast"
    vg = gss.ExponentialVariogram(range=50.0 * e / 512.0)
    N = e ** 3
    try:
        t0 = time.perf_counter()
        f = FFTGSHandle(vg, (e, e, e))
        torch.cuda.synchronize()
        pre_cold = time.perf_counter() - t0            # first create of the process (code objects, twiddles, pool)
        f.close()
        c["barrier"]()
        t0 = time.perf_counter()
        f = parallel.replicate_state(lambda compute: FFTGSHandle(vg, (e, e, e), spectrum=compute), share)
        c["barrier"]()
        pre_warm = c["max_over_ranks"](time.perf_counter() - t0)   # rank 0 computes, peers adopt the broadcast
    except _lib.GSSError as err:
        return {"error": str(err)}
    B = max(1, min(a.fftgs_batch, R))
    out = torch.empty((B, N), dtype=torch.float64, device="cuda")
    f.realize(4, rank * R, B, out=out)
    torch.cuda.synchronize()
    # (a) one realisation per call: the five passes back to back, each timed by its own HIP events
    _lib.profile_reset()
    _lib.profile_enable(True)
    for r in range(min(R, 16)):
        f.realize(4, rank * R + r, 1, out=out[:1])
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    names = ("fftgs_noise", "fftgs_fwd", "fftgs_phase", "fftgs_inv", "fftgs_p1", "fftgs_p2", "fftgs_p3", "fftgs_p4",
             "fftgs_p234", "fftgs_p5")   # p234: the three strided passes run slab by slab (GSS_FFTGS_SLAB)
    parts = {k: _lib.profile_read(k) for k in names}
    seq_ms = sum(v[0] / max(v[1], 1) for v in parts.values())
    # (b) the timed region: B realisations per call, HIP events on the launch stream around the whole region
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c["barrier"]()
    t0 = time.perf_counter()
    ev0.record()
    done = 0
    while done < R:
        nb = min(B, R - done)
        f.realize(4, rank * R + done, nb, out=out[:nb])
        done += nb
    ev1.record()
    c["barrier"]()
    dt = c["max_over_ranks"](time.perf_counter() - t0)
    kern_ms = ev0.elapsed_time(ev1) / R
    zc = out[0]
    svar = float((zc * zc).sum().item() / (N - 1))
    obuf = out
    # the per-GPU share of configs[2] end to end: preprocess (warm) + 32 realisations
    c["barrier"]()
    t0 = time.perf_counter()
    f2 = parallel.replicate_state(lambda compute: FFTGSHandle(vg, (e, e, e), spectrum=compute), share)
    for r0 in range(0, 32, B):
        f2.realize(4, rank * 32 + r0, min(B, 32 - r0), out=obuf[:min(B, 32 - r0)])
    c["barrier"]()
    dt32 = c["max_over_ranks"](time.perf_counter() - t0)
    f2.close()
    alg = 32.0 * N                                   # 32 N bytes per FP64 realisation
    res = {"metric": "FFTGS %d^3 realisations/sec" % e, "value": round(world * R / dt, 2), "unit": "realisations/s",
           "n_gpus": world, "scaling": "weak", "realisations_per_gpu": R,
           "ms_per_realisation": round(dt / R * 1e3, 3),
           "preprocess_s": {"cold": round(pre_cold, 4), "warm": round(pre_warm, 4),
                            "note": "cold = first gss_fftgs_create of the process; warm = rank 0 creates, peers adopt "
                                    "the RCCL broadcast of the state (N > 1) or a second create (N = 1)"},
           "end_to_end_32_per_gpu": {"value": round(world * 32 / dt32, 2), "unit": "realisations/s",
                                     "seconds": round(dt32, 4),
                                     "note": "configs[2] share of one GPU: preprocess + 32 realisations"},
           "roofline": {"bound": "hbm", "achieved": round(alg / (kern_ms * 1e-3) / 1e9, 1) if kern_ms else 0.0,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kern_ms else 0.0,
                        "traffic": None, "algorithmic_bytes": alg,
                        "kernels": "the five passes of one realisation; avg_ms = HIP events on the launch stream around "
                                   "the timed region / realisations (%d per gss_fftgs_realize call)" % B,
                        "avg_ms": round(kern_ms, 4), "sequential_ms": round(seq_ms, 4)},
           "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in parts.items() if v[1]},
           "sample_variance": svar}
    # SURVEY 8d's second number: the same realisations delivered to HOST memory, as the reference returns them
    # (fft.jl:173,197).  The library streams them out chunk by chunk behind the computation (csrc OutStream); beside it
    # the plain device -> pinned-host copy rate of this box, which bounds it.
    # (rank-local measurement inside the try, the collectives outside it: a rank that fails must not leave the
    # others alone in a barrier)
    import numpy as np
    Rh = max(2, min(8, R))
    loc, err_h = None, None
    try:
        hbuf = torch.empty((Rh, N), dtype=torch.float64, pin_memory=True)
        hbuf.zero_()
        torch.cuda.synchronize()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from copy_rate import d2h_copy_rate_gbs, copy_rate_fraction
        nb = min(Rh, obuf.shape[0])
        copy_gbs = d2h_copy_rate_gbs(torch, hbuf, obuf[:nb])          # warm, best of 4, the timed call's own buffer
        f.realize(4, rank * R, 1, out=hbuf[:1])                       # warm: bounce buffers, ring blocks
        t0 = time.perf_counter()
        f.realize(4, rank * R, Rh, out=hbuf)                          # returns when the last byte has arrived
        dt_local = time.perf_counter() - t0
        staged = _lib.stat("out_ring_bytes")
        same = bool(torch.equal(hbuf[0].cuda(), f.realize(4, rank * R, 1, out=obuf[:1])[0]))
        pg = np.empty((2, N))
        pg.fill(0.0)                                                  # touch the pages: first-touch faults are not PCIe
        t0 = time.perf_counter()
        f.realize(4, rank * R, 2, out=pg)
        dtp = time.perf_counter() - t0
        copy_gbs, frac, fnote = copy_rate_fraction(torch, Rh * N * 8 / dt_local / 1e9, copy_gbs, hbuf, obuf[:nb])
        loc = dict(dt=dt_local, copy_gbs=copy_gbs, frac=frac, fnote=fnote, staged=staged, same=same, dtp=dtp)
        del hbuf, pg
    except Exception as err:                                          # noqa: BLE001 -- the leg must not lose the line
        err_h = repr(err)
    failed = c["max_over_ranks"](0.0 if loc is not None else 1.0) > 0.0
    dth = c["max_over_ranks"](loc["dt"] if loc is not None else 0.0)   # slowest rank (all ranks transfer at once)
    if failed or loc is None:
        res["with_d2h"] = {"error": err_h or "a peer rank failed"}
    else:
        res["with_d2h"] = {"value": round(world * Rh / dth, 2), "unit": "realisations/s",
                           "realisations_per_gpu": Rh, "destination": "page-locked host memory",
                           "achieved_GBs": round(Rh * N * 8 / dth / 1e9, 2),
                           "pcie_copy_rate_GBs": round(loc["copy_gbs"], 2),
                           "frac_of_copy_rate": loc["frac"], "copy_rate_note": loc["fnote"],
                           "hbm_staged_bytes": loc["staged"], "bit_identical_to_device_path": loc["same"],
                           "pageable": {"value": round(2 / loc["dtp"], 2), "GBs": round(2 * N * 8 / loc["dtp"] / 1e9, 2),
                                        "note": "numpy destination: pinned bounce buffers + host copies in stream order"},
                           "note": "per-GPU figures (achieved, copy rate, pageable) are rank 0's; value = all ranks / slowest rank"}
    tfile = _latest_profile("fftgs_512_pmc_traffic.json")
    if e == 512 and tfile:
        tj = json.load(open(tfile))
        res["roofline"]["traffic"] = tj["hbm_bytes_per_realisation"]
        res["roofline"]["traffic_unit"] = "B per realisation (rocprofv3 PMC FETCH_SIZE*2+WRITE_SIZE, profiles/%s)" % os.path.basename(tfile)
    f.close()
    return res


def lugs_leg(c):
    """configs[3]: LUGS on a g x g grid with g^2/4 conditioning cells (128 -> 4 096 data, ns = 12 288), spherical
    range 20, `--lugs-reals` realisations sharded over the ranks.  Preprocess (lu.jl:76-169, dense Cholesky on the
    FP64 matrix cores) on rank 0 only, (L22, d2) broadcast over RCCL, lusim (lu.jl:198-224) on every rank."""
    a, gss, _lib, parallel, torch, np = c["a"], c["gss"], c["_lib"], c["parallel"], c["torch"], c["np"]
    rank, world = c["rank"], c["world"]
    from gss.engine import LUGSHandle
    g, Rtot = a.lugs, a.lugs_reals
    N, nd = g * g, g * g // 4
    cent = gss.CartesianGrid(g, g).centroids()
    dlocs = np.sort(np.random.default_rng(5).permutation(N)[:nd])
    z1 = np.random.default_rng(50).normal(size=nd)
    vg = gss.SphericalVariogram(range=20.0)
    ns = N - nd
    try:
        if rank == 0:                                   # warm the code objects and the buffer pool
            LUGSHandle(vg, cent, dlocs, z1).close()
        c["barrier"]()
        t0 = time.perf_counter()
        h = LUGSHandle(vg, cent, dlocs, z1, factor=(rank == 0))
        torch.cuda.synchronize()
        t_pre = time.perf_counter() - t0
        c["barrier"]()
        t0 = time.perf_counter()
        if c["a"].native_bcast and world > 1:
            parallel.native_comm()
            h.bcast_state(0)
        else:
            parallel.broadcast_(h.state_tensor(), 0)
            if rank != 0:
                h.adopt_state()
        c["barrier"]()
        t_bc = c["max_over_ranks"](time.perf_counter() - t0)
    except _lib.GSSError as err:
        return {"error": str(err)}
    t_pre0 = torch.tensor([t_pre if rank == 0 else 0.0], dtype=torch.float64)
    parallel.broadcast_(t_pre0, 0)
    t_pre = float(t_pre0.item())
    lo, hi = parallel.shard_range(Rtot, rank, world)
    h.realize(5, lo, max(hi - lo, 1), device=True)
    c["barrier"]()
    t0 = time.perf_counter()
    y, _ = h.realize(5, lo, hi - lo, device=True)
    c["barrier"]()
    dt = c["max_over_ranks"](time.perf_counter() - t0)
    ok = True
    if hi > lo:
        ok = bool(torch.equal(y[0, torch.as_tensor(dlocs, device="cuda")], torch.as_tensor(z1, device="cuda")))
    h.close()
    flops = nd ** 3 / 3 + nd * nd * ns + nd * ns * ns + ns ** 3 / 3
    return {"config": "configs[3]: LUGS %dx%d grid, %d conditioning cells, ns = %d, spherical range 20, %d realisations "
                      "over %d GPU(s)" % (g, g, nd, ns, Rtot, world),
            "n_gpus": world, "preprocess_s": round(t_pre, 4),
            "preprocess_roofline": {"bound": "mfma", "achieved": round(flops / t_pre / 1e12, 2),
                                    "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(flops / t_pre / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)},
            "state_bytes": 8 * (ns * ns + ns), "broadcast_s": round(t_bc, 4) if world > 1 else None,
            "broadcast_GBps": round(8 * (ns * ns + ns) / t_bc / 1e9, 1) if world > 1 else None,
            "realize_s": round(dt, 5), "value": round(Rtot / dt, 1), "unit": "realisations/s",
            "hard_data_honoured": ok}


if __name__ == "__main__":
    main()
