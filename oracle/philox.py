"""Counter-based randomness contract "mcle-philox-v1" (NumPy side).  TEST INFRASTRUCTURE.

The reference draws from NumPy's global legacy MT19937 (util/misc.py:354-355,
apps/awgn_modulators/simulate_psk.py:65,74); a GPU cannot reproduce that stream
cheaply, so the HIP pipelines use Philox4x32-10 (Salmon et al., SC'11; the
generator rocRAND calls ``philox4x32_10``) addressed so that every draw is a
pure function of (seed, realization index, stream, position):

    key     = (seed_lo, seed_hi)
    counter = (block, stream, realization_lo, realization_hi)

which is exactly rocRAND's ``rocrand_init(seed, subsequence=realization,
offset=4*(stream*2**32 + block))`` -- checked against rocRAND's own host engine
(known answers in tests/test_oracle_philox.py).  One block = 4 uint32 words.

Derived draws (identical formulas in pyphysim_amd/csrc/philox.hpp):

  symbols   index n of a stream: block n//16, word (n//4)%4, byte n%4;
            value = byte & (M-1)                      (M a power of two <= 256)
  uniforms  u_i = word(i) * 2**-32, block i//4, word i%4     (Jakes phi/psi)
  CN(0,1)   complex sample i: block i//2, words (2*(i%2), 2*(i%2)+1) = (x0, x1)
            rad = sqrt(-ln((x0 + 0.5) * 2**-32)),  v = x1 * 2**-32
            z = rad * (cos(2 pi v) + 1j sin(2 pi v))        (Box-Muller; E|z|^2 = 1)

This file is the oracle's copy of the contract; it shares no code with the
product.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint64(0x9E3779B9)
W1 = np.uint64(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
S32 = np.uint64(32)

STREAM_DATA = 0    # transmitted symbol indices
STREAM_NOISE = 1   # receiver AWGN
STREAM_CHAN = 2    # channel matrices / i.i.d. Rayleigh taps
STREAM_PHASE = 3   # Jakes phi / psi uniforms


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten Philox rounds on broadcastable uint32-valued arrays -> 4 uint32 arrays."""
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(
        *[np.asarray(a, dtype=np.uint64) & MASK for a in (c0, c1, c2, c3, k0, k1)])
    k0 = k0.copy()
    k1 = k1.copy()
    for rnd in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> S32) ^ c1 ^ k0
        n1 = p1 & MASK
        n2 = (p0 >> S32) ^ c3 ^ k1
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        if rnd != 9:
            k0 = (k0 + W0) & MASK
            k1 = (k1 + W1) & MASK
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def blocks(seed, realization, stream, block):
    """uint32 words [..., 4] for broadcastable realization / block index arrays."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    realization = np.asarray(realization, dtype=np.uint64)
    block = np.asarray(block, dtype=np.uint64)
    w = philox4x32_10(block & MASK, np.uint64(stream), realization & MASK,
                      realization >> S32, np.uint64(seed & 0xFFFFFFFF),
                      np.uint64(seed >> 32))
    return np.stack(w, axis=-1)


def words(seed, realization, stream, n):
    """First n uint32 words of a (realization, stream)."""
    nb = (n + 3) // 4
    return blocks(seed, realization, stream, np.arange(nb)).reshape(-1)[:n]


def symbols(seed, realization, n, M, stream=STREAM_DATA, offset=0):
    """n symbol indices in [0, M) starting at symbol position `offset`."""
    assert M & (M - 1) == 0 and 2 <= M <= 256
    pos = np.arange(offset, offset + n, dtype=np.int64)
    blk = pos // 16
    lo, hi = int(blk[0]) if n else 0, (int(blk[-1]) + 1 if n else 0)
    w = blocks(seed, realization, stream, np.arange(lo, hi)).astype(np.uint32)
    word = w[blk - lo, (pos // 4) % 4]
    byte = (word >> ((pos % 4) * 8).astype(np.uint32)) & np.uint32(0xFF)
    return (byte & np.uint32(M - 1)).astype(np.int64)


def uniforms(seed, realization, n, stream=STREAM_PHASE, offset=0):
    """n float64 uniforms in [0,1) starting at position `offset`."""
    pos = np.arange(offset, offset + n, dtype=np.int64)
    blk = pos // 4
    lo, hi = (int(blk[0]), int(blk[-1]) + 1) if n else (0, 0)
    w = blocks(seed, realization, stream, np.arange(lo, hi))
    return w[blk - lo, pos % 4].astype(np.float64) * 2.0 ** -32


def cnormal(seed, realization, n, stream, offset=0):
    """n complex128 CN(0,1) samples starting at sample position `offset`."""
    pos = np.arange(offset, offset + n, dtype=np.int64)
    blk = pos // 2
    lo, hi = (int(blk[0]), int(blk[-1]) + 1) if n else (0, 0)
    w = blocks(seed, realization, stream, np.arange(lo, hi))
    x0 = w[blk - lo, 2 * (pos % 2)].astype(np.float64)
    x1 = w[blk - lo, 2 * (pos % 2) + 1].astype(np.float64)
    rad = np.sqrt(-np.log((x0 + 0.5) * 2.0 ** -32))
    ang = 2.0 * np.pi * (x1 * 2.0 ** -32)
    return rad * (np.cos(ang) + 1j * np.sin(ang))
