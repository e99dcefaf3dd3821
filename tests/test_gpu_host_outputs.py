"""Host outputs of the simulation calls leave the device in chunks (csrc/common.hip, OutStream).

The reference hands every realisation back as a host vector (fft.jl:173,197; lu.jl:217-221; seq.jl:137-141).  The
library stages at most three chunks of realisations in HBM and copies chunk c out while chunk c + 1 is computed;
`GSS_OUT_CHUNK_MB` shrinks the chunk so that small problems run through many more chunks than the ring holds.
Bar: bit-identical to the device-resident path (same kernels, same order), for pageable destinations (pinned bounce
buffers + host copies in stream order) and page-locked ones (written by the DMA engine directly)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims", [(64, 64, 64), (40, 36, 20)])          # fused passes / rocFFT pipeline
def test_fftgs_host_chunks_equal_device_outputs(dims, monkeypatch):
    import torch
    import gss
    from gss.engine import FFTGSHandle
    N = int(np.prod(dims))
    h = FFTGSHandle(gss.ExponentialVariogram(range=9.0, sill=1.3), dims, mean=0.4)
    R = 23
    ref = h.realize(11, 5, R, device=True)
    torch.cuda.synchronize()
    ref = ref.cpu().numpy()
    monkeypatch.setenv("GSS_OUT_CHUNK_MB", str(max(1, (2 * N * 8) >> 20)))      # two realisations per chunk: 12 chunks
    host = h.realize(11, 5, R)                                                  # pageable numpy array
    assert isinstance(host, np.ndarray) and np.array_equal(host, ref)
    pin = h.realize(11, 5, R, pinned=True)                                      # page-locked destination
    assert isinstance(pin, np.ndarray) and np.array_equal(pin, ref)
    out = torch.empty((R, N), dtype=torch.float64, pin_memory=True)
    assert h.realize(11, 5, R, out=out) is out and np.array_equal(out.numpy(), ref)
    inds = np.random.default_rng(0).permutation(N)[: N // 3]
    sub = h.realize(11, 5, R, inds=inds)
    assert np.array_equal(sub, ref[:, inds])
    # supplied noise from host memory, one realisation at a time
    u = np.random.default_rng(1).uniform(size=(7, N))
    zd = h.realize(0, 0, 7, noise=torch.as_tensor(u, device="cuda"))
    torch.cuda.synchronize()
    assert np.array_equal(h.realize(0, 0, 7, noise=u), zd.cpu().numpy())
    monkeypatch.delenv("GSS_OUT_CHUNK_MB")
    assert np.array_equal(h.realize(11, 5, R), ref)                             # one chunk: everything at the end
    h.close()


def test_fftgs_ring_is_bounded(monkeypatch):
    """The HBM staged for host outputs is three chunks, whatever the number of realisations: 40 realisations on 128^3
    cells (16 MiB each) with a 16 MiB chunk move as 40 chunks through a ring of 48 MiB."""
    import torch
    import gss
    from gss import _lib
    from gss.engine import FFTGSHandle
    h = FFTGSHandle(gss.ExponentialVariogram(range=12.0), (128, 128, 128))
    monkeypatch.setenv("GSS_OUT_CHUNK_MB", "16")
    z = h.realize(3, 0, 40)
    assert _lib.stat("out_chunks") == 40 and _lib.stat("out_ring_bytes") == 3 * 16 * 2 ** 20
    zd = h.realize(3, 0, 40, device=True)
    torch.cuda.synchronize()
    assert np.array_equal(z, zd.cpu().numpy())
    h.close()
    _lib.trim_pool()
    assert _lib.stat("pool_bytes") == 0


def test_lugs_host_blocks_equal_device_outputs(monkeypatch):
    import torch
    import gss
    from gss.engine import LUGSHandle
    from oracle import fftgs as offt
    cent = offt.grid_centroids((70, 70))
    rng = np.random.default_rng(3)
    dl = np.sort(rng.choice(4900, 300, replace=False))
    z1 = rng.normal(size=300)
    h = LUGSHandle(gss.SphericalVariogram(range=9.0, nugget=0.05), cent, dl, z1)
    h2 = LUGSHandle(gss.ExponentialVariogram(range=7.0), cent, dl, z1)
    R = 100
    yd, wd = h.realize(5, 2, R, device=True)
    y2d, w2d = h2.realize(6, 2, R, rho=0.8, w1=wd, device=True)
    torch.cuda.synchronize()
    monkeypatch.setenv("GSS_OUT_CHUNK_MB", "1")            # 26 realisations per block: four blocks, three slots
    y, w = h.realize(5, 2, R)
    assert np.array_equal(y, yd.cpu().numpy()) and np.array_equal(w, wd.cpu().numpy())
    y2, w2 = h2.realize(6, 2, R, rho=0.8, w1=w)            # the first variable's normals from host memory (lu.jl:188-193)
    assert np.array_equal(y2, y2d.cpu().numpy()) and np.array_equal(w2, w2d.cpu().numpy())
    wn = rng.normal(size=(R, h.ns))
    yn, _ = h.realize(0, 0, R, noise=wn)
    ynd, _ = h.realize(0, 0, R, noise=torch.as_tensor(wn, device="cuda"))
    torch.cuda.synchronize()
    assert np.array_equal(yn, ynd.cpu().numpy())
    h.close()
    h2.close()


@pytest.mark.parametrize("per_realisation_paths", [False, True])
def test_sgs_host_blocks_equal_device_outputs(per_realisation_paths, monkeypatch):
    import torch
    import gss
    from gss.engine import SGSHandle
    from oracle import fftgs as offt
    cent = offt.grid_centroids((50, 50))
    N = 2500
    rng = np.random.default_rng(8)
    dl = np.sort(rng.choice(N, 40, replace=False))
    zd = rng.normal(size=40)
    R = 200 if not per_realisation_paths else 60
    path = rng.permutation(N) if not per_realisation_paths else np.stack([rng.permutation(N) for _ in range(R)])
    h = SGSHandle(gss.SphericalVariogram(range=8.0, nugget=0.05), cent, path, dl, zd, 0.1, 12, 1)
    ref = h.realize(21, 0, R, device=True)
    torch.cuda.synchronize()
    monkeypatch.setenv("GSS_OUT_CHUNK_MB", "1")            # 52 realisations per MiB: blocks of 64 lanes / 52 paths
    got = h.realize(21, 0, R)
    assert np.array_equal(got, ref.cpu().numpy())
    h.close()


def test_dev_to_host_copies_exactly():
    """gss_dev_to_host (the transfer path the twin uses for results composed on the device): pageable and page-locked
    destinations, sizes around the 32 MiB pieces of the bounce pipeline, data produced on the current stream just before."""
    import torch
    from gss.engine import to_host
    for n in (1, 1000, (32 << 20) // 8 - 1, (32 << 20) // 8 + 5, 3 * (32 << 20) // 8 + 17):
        t = torch.arange(n, dtype=torch.float64, device="cuda") * 0.5 + 1.0
        t = t * 2.0                                    # queued on the current stream: the copy must wait for it
        got = to_host(t)
        assert got.shape == (n,) and got[0] == 2.0 and got[-1] == (n - 1) + 2.0 and np.array_equal(got, t.cpu().numpy())
    ti = torch.arange(12, dtype=torch.int32, device="cuda").reshape(3, 4)
    assert np.array_equal(to_host(ti), np.arange(12, dtype=np.int32).reshape(3, 4))
