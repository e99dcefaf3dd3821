"""N > 1 path on CPU: two gloo ranks shard domain points / realisations (parallel.shard_range), use
the oracle stand-in engine for arithmetic, and must reproduce the single-process result exactly.
The only collective on the data path is the optional gather of results; the factor broadcast
plumbing (parallel.broadcast_) is exercised on a CPU tensor."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem_inputs():
    rng = np.random.default_rng(42)
    xy = rng.uniform(0, 50, (40, 2))
    z = rng.normal(size=40)
    dom = rng.uniform(0, 50, (101, 2))        # odd size -> uneven shards
    return xy, z, dom


class _NotPosDef:
    """Variogram stand-in whose covariance is not positive definite: makes rank 0's LUGS preprocess raise."""
    kind, sill, nugget, range, nu, radii = "gaussian", 1.0, 0.0, 1e9, 1.0, None
    regularize = False                 # without the Gaussian model's nugget epsilon (gss/variograms.py)

    def isstationary(self):
        return True


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gss
        from gss import parallel
        from oracle_engine import OracleEngine
        xy, z, dom = _problem_inputs()
        out = {}
        # kriging: domain points sharded, gathered back
        prob = gss.EstimationProblem(gss.georef({"z": z}, xy), gss.PointSet(dom), "z")
        solver = gss.KrigingSolver(("z", dict(variogram=gss.ExponentialVariogram(range=20.0), maxneighbors=8)),
                                   engine=OracleEngine)
        sol = gss.solve(prob, solver, gather=True)
        out["krig_mu"], out["krig_var"] = sol["z"], sol["z_variance"]
        local = gss.solve(prob, solver, gather=False)     # every rank keeps its own block
        lo, hi = parallel.shard_range(101, rank, world)
        out["local_len"] = (len(local["z"]), hi - lo)
        # FFTGS / LUGS: realisations sharded; realisation r depends only on (seed, r)
        grid = gss.CartesianGrid(16, 12)
        s1 = gss.solve(gss.SimulationProblem(grid, ("z", float), 5),
                       gss.FFTGS(("z", dict(variogram=gss.SphericalVariogram(range=5.0))), rng=9, engine=OracleEngine),
                       gather=True)
        out["fft"] = np.stack(s1["z"])
        S = gss.georef({"z": [0.0, 1.0]}, np.array([[2.0], [20.0]]))
        from oracle_engine import _LUGS
        before = _LUGS.ncomputed
        s2 = gss.solve(gss.SimulationProblem(S, gss.CartesianGrid(24), "z", 3),
                       gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=6.0))), rng=9, engine=OracleEngine),
                       gather=True)
        out["lu"] = np.stack(s2["z"])
        # preprocess once (lu.jl:76): only rank 0 factorises, the peer adopts the broadcast state
        out["lu_factorisations"] = _LUGS.ncomputed - before
        s2r = gss.solve(gss.SimulationProblem(S, gss.CartesianGrid(24), "z", 3),
                        gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=6.0))), rng=9, engine=OracleEngine,
                                 share="recompute"), gather=False)
        lo_r, hi_r = parallel.shard_range(3, rank, world)
        out["lu_local"] = (np.stack(s2r["z"]) if hi_r > lo_r else np.empty((0, 24)), lo_r, hi_r)
        # a failure of rank 0's preprocess reaches every rank as an exception, not as a hung broadcast
        try:
            gss.solve(gss.SimulationProblem(gss.CartesianGrid(24), ("z", float), 2),
                      gss.LUGS(("z", dict(variogram=_NotPosDef())), rng=9, engine=OracleEngine))
            out["error_propagated"] = False
        except Exception:                                   # noqa: BLE001
            out["error_propagated"] = True
        # IDW / LWR shard estimation points, SGS shards realisations (section 8f rows)
        si = gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=6, exponent=2)), engine=OracleEngine), gather=True)
        sl = gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=9)), engine=OracleEngine), gather=True)
        out["idw"], out["lwr"] = np.c_[si["z"], si["z_distance"]], np.c_[sl["z"], sl["z_variance"]]
        s3 = gss.solve(gss.SimulationProblem(S, gss.CartesianGrid(24), "z", 3),
                       gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=6.0), maxneighbors=5)), rng=9,
                               engine=OracleEngine), gather=True)
        out["sgs"] = np.stack(s3["z"])
        # factor broadcast plumbing
        t = torch.arange(1000, dtype=torch.float64) * (1.0 if rank == 0 else -1.0)
        parallel.broadcast_(t, src=0)
        out["bcast_ok"] = bool(torch.equal(t, torch.arange(1000, dtype=torch.float64)))
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions():
    sys.path.insert(0, os.path.join(ROOT, "geostatssolvers.jl_amd"))
    from gss.parallel import shard_range
    for total in (0, 1, 7, 8, 101, 10 ** 7):
        for ws in (1, 2, 3, 8):
            parts = [shard_range(total, r, ws) for r in range(ws)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(420)
@pytest.mark.parametrize("world", [2, 8])
def test_ranks_reproduce_single_process_results(world):
    """world = 2, and world = 8 -- the node the driver launches: more ranks than realisations (5 FFTGS, 3 LUGS / SGS), so
    several ranks hold EMPTY shards and still take part in the broadcasts and gathers."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=360) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single-process reference, same stand-in engine
    for pth in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd"), HERE):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import gss
    from oracle_engine import OracleEngine
    xy, z, dom = _problem_inputs()
    prob = gss.EstimationProblem(gss.georef({"z": z}, xy), gss.PointSet(dom), "z")
    solver = gss.KrigingSolver(("z", dict(variogram=gss.ExponentialVariogram(range=20.0), maxneighbors=8)),
                               engine=OracleEngine)
    ref = gss.solve(prob, solver)
    grid = gss.CartesianGrid(16, 12)
    f = gss.solve(gss.SimulationProblem(grid, ("z", float), 5),
                  gss.FFTGS(("z", dict(variogram=gss.SphericalVariogram(range=5.0))), rng=9, engine=OracleEngine))
    S = gss.georef({"z": [0.0, 1.0]}, np.array([[2.0], [20.0]]))
    l = gss.solve(gss.SimulationProblem(S, gss.CartesianGrid(24), "z", 3),
                  gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=6.0))), rng=9, engine=OracleEngine))
    ri = gss.solve(prob, gss.IDWSolver(("z", dict(maxneighbors=6, exponent=2)), engine=OracleEngine))
    rl = gss.solve(prob, gss.LWRSolver(("z", dict(maxneighbors=9)), engine=OracleEngine))
    rs = gss.solve(gss.SimulationProblem(S, gss.CartesianGrid(24), "z", 3),
                   gss.SGS(("z", dict(variogram=gss.SphericalVariogram(range=6.0), maxneighbors=5)), rng=9,
                           engine=OracleEngine))
    for rank in range(world):
        out = results[rank]
        assert np.array_equal(out["idw"], np.c_[ri["z"], ri["z_distance"]])
        assert np.array_equal(out["lwr"], np.c_[rl["z"], rl["z_variance"]])
        assert np.array_equal(out["sgs"], np.stack(rs["z"]))
        assert np.array_equal(out["krig_mu"], ref["z"]) and np.array_equal(out["krig_var"], ref["z_variance"])
        assert out["local_len"][0] == out["local_len"][1]
        assert np.array_equal(out["fft"], np.stack(f["z"])) and np.array_equal(out["lu"], np.stack(l["z"]))
        assert out["bcast_ok"]
        assert out["lu_factorisations"] == (1 if rank == 0 else 0)
        got, lo_r, hi_r = out["lu_local"]
        assert np.array_equal(got, np.stack(l["z"])[lo_r:hi_r])
        assert out["error_propagated"]


# ------------------------------------------------------------------------------------------------------------------
# The same sharding with the product engine: two gloo ranks share the one GPU of the test box (three processes on the
# card with the test runner; the driver runs the real one-rank-per-GPU RCCL case).
# ------------------------------------------------------------------------------------------------------------------
def _gpu_cases(gss):
    rng = np.random.default_rng(5)
    xy = rng.uniform(0, 50, (300, 2))
    z = rng.normal(size=300)
    dom = rng.uniform(0, 50, (5001, 2))
    data = gss.georef({"z": z}, xy)
    grid = gss.CartesianGrid(32, 32, 16)
    cdat = gss.georef({"z": rng.normal(size=30)}, rng.uniform(0, 16, (30, 3)))
    return {
        "krig": (gss.EstimationProblem(data, gss.PointSet(dom), "z"),
                 lambda: gss.KrigingSolver(("z", dict(variogram=gss.MaternVariogram(range=20.0, order=1.5))))),
        "knn": (gss.EstimationProblem(data, gss.PointSet(dom), "z"),
                lambda: gss.KrigingSolver(("z", dict(variogram=gss.ExponentialVariogram(range=20.0), maxneighbors=24,
                                                     degree=1)))),
        "idw": (gss.EstimationProblem(data, gss.PointSet(dom), "z"), lambda: gss.IDWSolver(("z", dict(maxneighbors=12)))),
        "fft": (gss.SimulationProblem(grid, ("z", float), 5),
                lambda: gss.FFTGS(("z", dict(variogram=gss.ExponentialVariogram(range=6.0))), rng=11)),
        "cfft": (gss.SimulationProblem(cdat, grid, "z", 3),
                 lambda: gss.FFTGS(("z", dict(variogram=gss.ExponentialVariogram(range=6.0))), rng=11)),
        "lu": (gss.SimulationProblem(gss.georef({"z": [0.0, 1.0, -1.0]}, np.array([[2.0, 3.0], [20.0, 9.0], [11.0, 11.0]])),
                                     gss.CartesianGrid(24, 16), "z", 5),
               lambda: gss.LUGS(("z", dict(variogram=gss.SphericalVariogram(range=6.0))), rng=9)),
    }


def _as_array(sol, kind):
    if kind in ("krig", "knn"):
        return np.c_[sol["z"], sol["z_variance"]]
    if kind == "idw":
        return np.c_[sol["z"], sol["z_distance"]]
    return np.stack(sol["z"])


def _gpu_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gss
        torch.cuda.set_device(0)
        out = {k: _as_array(gss.solve(prob, mk(), gather=True), k) for k, (prob, mk) in _gpu_cases(gss).items()}
        # the broadcast-adopt path spelled out on the C-ABI handles: the peer allocates only, cannot realise before
        # the state has arrived, and afterwards produces rank 0's realisations
        from gss import parallel
        from gss._lib import GSSError
        from gss.engine import FFTGSHandle, LUGSHandle
        f = FFTGSHandle(gss.ExponentialVariogram(range=6.0), (32, 32, 16), spectrum=(rank == 0))
        cent = gss.CartesianGrid(24, 16).centroids()
        l = LUGSHandle(gss.SphericalVariogram(range=6.0), cent, [5, 77, 200], [0.5, -1.0, 2.0], factor=(rank == 0))
        if rank != 0:
            for h in (f, l):
                try:
                    h.realize(3, 0, 1)
                    raise AssertionError("a handle without state must refuse to realise")
                except GSSError:
                    pass
        for h in (f, l):
            parallel.broadcast_(h.state_tensor(), 0)
            if rank != 0:
                h.adopt_state()
        out["fft_adopt"] = f.realize(3, 0, 2)
        out["lu_adopt"] = l.realize(3, 0, 2)[0]
        f.close()
        l.close()
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_reproduce_single_process_results():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for pth in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd"), HERE):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import gss
    from gss.engine import FFTGSHandle, LUGSHandle
    f = FFTGSHandle(gss.ExponentialVariogram(range=6.0), (32, 32, 16))
    l = LUGSHandle(gss.SphericalVariogram(range=6.0), gss.CartesianGrid(24, 16).centroids(), [5, 77, 200], [0.5, -1.0, 2.0])
    fref, lref = f.realize(3, 0, 2), l.realize(3, 0, 2)[0]
    for rank in (0, 1):
        assert np.array_equal(results[rank]["fft_adopt"], fref) and np.array_equal(results[rank]["lu_adopt"], lref)
    for kind, (prob, mk) in _gpu_cases(gss).items():
        ref = _as_array(gss.solve(prob, mk()), kind)
        for rank in (0, 1):
            got = results[rank][kind]
            assert got.shape == ref.shape, kind
            # shards are separate launches: the summation order inside a launch may differ from the single call
            assert np.allclose(got, ref, rtol=0, atol=1e-10, equal_nan=True), (kind, float(np.nanmax(np.abs(got - ref))))
