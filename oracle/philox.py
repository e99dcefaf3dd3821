"""Counter-based noise shared by oracle and device (test infrastructure only).

The reference draws `rand(rng, V, dims)` (`/root/reference/src/simulation/fft.jl:163`)
and `randn(rng, n)` (`src/simulation/lu.jl:209`) from a Julia RNG whose stream cannot be
reproduced outside Julia ("RNG parity is impossible").  The build's contract
instead: Philox4x32-10 keyed by `seed`, counter = (block_lo, block_hi, realisation, stream),
so realisation r's noise depends only on (seed, r) and sharding never changes results.

    block b -> 4 x u32 (x0..x3)
    ua = ((x0 << 32 | x1) >> 11) * 2^-53 ,  ub = ((x2 << 32 | x3) >> 11) * 2^-53
    uniform element e (stream 0): block = e >> 1, value = ua if e even else ub
    normal  element e (stream 1): block = e, value = sqrt(-2 ln(1 - ua)) * cos(2 pi ub)
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
STREAM_UNIFORM = 0
STREAM_NORMAL = 1


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; counters are uint32 arrays, key scalars."""
    c0 = np.asarray(c0, dtype=np.uint32).copy()
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint32), c0.shape).copy()
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint32), c0.shape).copy()
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint32), c0.shape).copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _pairs(seed: int, real: int, stream: int, blocks: np.ndarray):
    blocks = np.asarray(blocks, dtype=np.uint64)
    x0, x1, x2, x3 = philox4x32_10((blocks & MASK).astype(np.uint32),
                                   (blocks >> np.uint64(32)).astype(np.uint32),
                                   np.uint32(real & 0xFFFFFFFF), np.uint32(stream),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    two53 = 1.0 / 9007199254740992.0
    ua = (((x0.astype(np.uint64) << np.uint64(32)) | x1.astype(np.uint64)) >> np.uint64(11)).astype(np.float64) * two53
    ub = (((x2.astype(np.uint64) << np.uint64(32)) | x3.astype(np.uint64)) >> np.uint64(11)).astype(np.float64) * two53
    return ua, ub


def uniform(seed: int, real: int, n: int) -> np.ndarray:
    """n uniform doubles in [0,1) for realisation `real` (bit-exact with the device)."""
    nb = (n + 1) // 2
    ua, ub = _pairs(seed, real, STREAM_UNIFORM, np.arange(nb, dtype=np.uint64))
    out = np.empty(2 * nb)
    out[0::2] = ua
    out[1::2] = ub
    return out[:n]


def normal(seed: int, real: int, n: int) -> np.ndarray:
    """n standard normals (Box-Muller); device agrees to libm rounding (~1e-15)."""
    ua, ub = _pairs(seed, real, STREAM_NORMAL, np.arange(n, dtype=np.uint64))
    return np.sqrt(-2.0 * np.log(1.0 - ua)) * np.cos(2.0 * np.pi * ub)
