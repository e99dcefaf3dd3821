"""The library's own multi-GPU routes (include/gss.h "multi-GPU"; csrc/comm.hip), without torch.distributed:

* HIP IPC: an owner process computes the preprocess state (kriging factor, FFTGS spectrum, LUGS factor), exports an
  96-byte token; a second process -- here on the SAME device, on a node one per GPU -- creates its handles without
  state, imports, and must then produce the owner's results bit for bit.  Only the token crosses between the
  processes (a pipe), as a Julia host would send it with `remotecall`.
* RCCL: communicator of one rank (all this box can hold: RCCL refuses two ranks on one device); `gss_state_bcast` runs
  the real ncclBroadcast call path, loads RCCL at run time and leaves the state intact.

Reference: preprocess once, solvesingle mapped over workers -- /root/reference/src/simulation/fft.jl:62,145; lu.jl:76,171."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _paths():
    for p in (ROOT, os.path.join(ROOT, "geostatssolvers.jl_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _inputs():
    rng = np.random.default_rng(17)
    x = rng.uniform(0, 100, (400, 3))
    z = rng.normal(size=400)
    x0 = rng.uniform(0, 100, (3000, 3))
    return x, z, x0


def _handles(compute):
    import gss
    from gss.engine import FFTGSHandle, KrigHandle, LUGSHandle
    x, z, _ = _inputs()
    k = KrigHandle(gss.MaternVariogram(range=30.0, order=1.5), 2, x, z, degree=1, factor=compute)
    f = FFTGSHandle(gss.ExponentialVariogram(range=7.0), (64, 32, 32), spectrum=compute)
    cent = gss.CartesianGrid(40, 30).centroids()
    l = LUGSHandle(gss.SphericalVariogram(range=8.0), cent, [3, 500, 1100], [0.5, -1.0, 2.0], factor=compute)
    return k, f, l


def _results(k, f, l):
    _, _, x0 = _inputs()
    mu, var, _ = k.predict_global(x0)
    return dict(mu=mu, var=var, fft=f.realize(9, 4, 3), lu=l.realize(9, 4, 3)[0])


def _peer(conn):
    _paths()
    import torch
    torch.cuda.set_device(0)
    from gss._lib import GSSError
    k, f, l = _handles(False)
    try:
        f.realize(9, 4, 1)
        conn.send(("error", "a handle without state realised"))
        return
    except GSSError:
        pass
    tokens = conn.recv()                       # the only thing that travels: 3 x 96 bytes
    try:
        for h, t in zip((k, f, l), tokens):
            h.import_state(t)
        conn.send(("ok", _results(k, f, l)))
    except Exception as e:                     # noqa: BLE001
        conn.send(("error", repr(e)))
    conn.recv()                                # stay until the owner has compared (nothing else to keep alive here)


def test_ipc_import_in_a_second_process_reproduces_the_owner():
    _paths()
    import torch
    torch.cuda.set_device(0)
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    p = ctx.Process(target=_peer, args=(b,))
    p.start()
    k, f, l = _handles(True)
    ref = _results(k, f, l)
    tokens = [h.export_state() for h in (k, f, l)]
    assert all(len(t) == 96 for t in tokens)
    a.send(tokens)
    assert a.poll(300), "the peer process did not answer"
    status, got = a.recv()
    a.send("done")
    p.join(timeout=60)
    assert status == "ok", got
    assert p.exitcode == 0
    for key in ref:
        assert np.array_equal(got[key], ref[key]), key
    # a token for another grid is refused, not copied
    import gss
    from gss._lib import GSSError
    from gss.engine import FFTGSHandle
    other = FFTGSHandle(gss.ExponentialVariogram(range=7.0), (32, 32, 32), spectrum=False)
    with pytest.raises(GSSError, match="state sizes differ"):
        other.import_state(tokens[1])


def _route_peer(conn):
    """Second process of the route test: imports the owner's token along every route and reports what happened."""
    _paths()
    import os
    import torch
    torch.cuda.set_device(0)
    import gss
    from gss import _lib
    from gss._lib import GSSError
    from gss.engine import FFTGSHandle
    vg = gss.ExponentialVariogram(range=7.0)
    tok = conn.recv()
    out = {}
    try:
        b = FFTGSHandle(vg, (32, 32, 32), spectrum=False)
        b.import_state(tok)
        out["same"] = (_lib.stat("ipc_route"), b.realize(3, 0, 1))
        other = bytearray(tok)
        other[84:88] = (250).to_bytes(4, "little")               # pci_bus of a device this process does not see
        c = FFTGSHandle(vg, (32, 32, 32), spectrum=False)
        c.import_state(bytes(other))
        out["hidden"] = (_lib.stat("ipc_route"), c.realize(3, 0, 1))
        for forced in ("nopeer", "peer"):
            os.environ["GSS_IPC_FORCE_ROUTE"] = forced
            d = FFTGSHandle(vg, (32, 32, 32), spectrum=False)
            try:
                d.import_state(tok)
                msg = "imported"
            except GSSError as e:
                msg = str(e)
            try:
                d.realize(3, 0, 1)
                left = "has state"
            except GSSError as e:
                left = str(e)
            out[forced] = (_lib.stat("ipc_route"), msg, left)
        os.environ.pop("GSS_IPC_FORCE_ROUTE", None)
        conn.send(("ok", out))
    except Exception as e:                     # noqa: BLE001
        conn.send(("error", repr(e)))
    conn.recv()


def test_ipc_import_chooses_its_route_from_the_owners_device():
    """The token carries the PCI identity of the owner's device and gss_state_ipc_import decides before it maps anything:
    same device (what this box can really do), a visible peer (peer access enabled, then the copy), a device this process
    does not see (one process per GPU behind HIP_VISIBLE_DEVICES: the mapping itself is the test), or a visible device
    without peer access -- refused with a message that names the alternatives.  One GPU: the other routes are reached
    with a token whose bus number is not among the visible devices, and by GSS_IPC_FORCE_ROUTE (read per call) on the
    real token, which exercises the branch selection, the peer-access call (to the own device HIP answers with an error
    that the library must report, not crash on) and the refusal."""
    _paths()
    import torch
    torch.cuda.set_device(0)
    import gss
    from gss.engine import FFTGSHandle
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    p = ctx.Process(target=_route_peer, args=(b,))
    p.start()
    own = FFTGSHandle(gss.ExponentialVariogram(range=7.0), (32, 32, 32))
    ref = own.realize(3, 0, 1)
    a.send(own.export_state())
    assert a.poll(300), "the peer process did not answer"
    status, got = a.recv()
    a.send("done")
    p.join(timeout=60)
    assert status == "ok", got
    assert got["same"][0] == 0 and np.array_equal(got["same"][1], ref)
    assert got["hidden"][0] == 2 and np.array_equal(got["hidden"][1], ref)
    route, msg, left = got["nopeer"]
    assert route == 3 and "not peer-accessible" in msg and "gss_state_bcast" in msg and "no spectrum" in left
    route, msg, left = got["peer"]
    assert route == 1 and "enabling peer access" in msg and "no spectrum" in left


def test_rccl_route_with_a_single_rank_communicator():
    _paths()
    import torch
    torch.cuda.set_device(0)
    from gss import _lib
    assert _lib.comm_info() == (-1, 0)
    with pytest.raises(_lib.GSSError, match="no communicator"):
        k, f, l = _handles(True)
        f.bcast_state(0)
    ref = _results(k, f, l)
    uid = _lib.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    _lib.comm_init(uid, 0, 1)
    try:
        assert _lib.comm_info() == (0, 1)
        with pytest.raises(_lib.GSSError, match="exists already"):
            _lib.comm_init(uid, 0, 1)
        for h in (k, f, l):
            h.bcast_state(0)                   # ncclBroadcast(root = self): the state must come through unchanged
        got = _results(k, f, l)
        for key in ref:
            assert np.array_equal(got[key], ref[key]), key
        with pytest.raises(_lib.GSSError, match="root"):
            f.bcast_state(3)
    finally:
        _lib.comm_destroy()
    assert _lib.comm_info() == (-1, 0)
