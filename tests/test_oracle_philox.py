"""CPU: the NumPy statement of the mcle-philox-v1 contract.

Known answers come from rocRAND's own host engine (rocrand_init(seed, subsequence, offset) +
rocrand4) compiled in the build container; the first one is also the Random123 KAT."""
import numpy as np

from oracle import chains, philox as P

KAT = [  # (seed, subsequence, offset) -> 4 words
    ((0, 0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0x12345678abcdef, 5, 0), (0x8afcfd48, 0xfe731a5c, 0x4eb00499, 0x4799b473)),
    ((42, 1000000007, 4 * ((3 << 32) | 17)), (0xe88c4a6a, 0x154170de, 0xac495290, 0x4c94c07f)),
    ((0xffffffffffffffff, 0xffffffffffffffff, 4 * 0x3fffffffffffffff),
     (0x8c5f4338, 0x4a57523d, 0x7e300cb1, 0x411fcefd)),
]


def test_rocrand_known_answers():
    for (seed, sub, off), want in KAT:
        blk = off // 4
        got = P.blocks(seed, sub, blk >> 32, blk & 0xffffffff)
        assert tuple(int(x) for x in got) == want


def test_draw_layouts():
    seed, r = 99, 12345678901
    w = P.words(seed, r, P.STREAM_DATA, 64)
    s = P.symbols(seed, r, 100, 64)
    assert s[0] == (w[0] & 0xff) & 63 and s[5] == ((w[1] >> 8) & 0xff) & 63 and s[17] == ((w[4] >> 8) & 63)
    assert np.array_equal(P.symbols(seed, r, 40, 16, offset=37), P.symbols(seed, r, 77, 16)[37:])
    z = P.cnormal(seed, r, 50, P.STREAM_NOISE)
    assert np.array_equal(P.cnormal(seed, r, 20, P.STREAM_NOISE, offset=13), z[13:33])
    wn = P.words(seed, r, P.STREAM_NOISE, 8)
    rad = np.sqrt(-np.log((float(wn[2]) + 0.5) * 2.0 ** -32))
    ang = 2 * np.pi * float(wn[3]) * 2.0 ** -32
    assert abs(z[1] - rad * np.exp(1j * ang)) < 1e-15
    u = P.uniforms(seed, r, 10)
    assert u[5] == float(P.words(seed, r, P.STREAM_PHASE, 8)[5]) * 2.0 ** -32
    assert not np.array_equal(P.cnormal(seed, r + 1, 8, P.STREAM_NOISE), z[:8])


def test_moments():
    z = P.cnormal(7, 3, 200000, P.STREAM_NOISE)
    assert abs(np.mean(np.abs(z) ** 2) - 1.0) < 0.01 and abs(z.real.var() - 0.5) < 0.01
    assert abs(np.mean(z.real * z.imag)) < 0.01 and abs(z.mean()) < 0.01
    cnt = np.bincount(P.symbols(7, 3, 64000, 64), minlength=64)
    assert cnt.min() > 800 and cnt.max() < 1200


def test_philox_rng_is_position_addressed():
    a = chains.PhiloxRng(5, 9)
    first = a.cn(P.STREAM_NOISE, 10)
    second = a.cn(P.STREAM_NOISE, 4, 5)
    both = P.cnormal(5, 9, 30, P.STREAM_NOISE)
    assert np.array_equal(first, both[:10]) and np.array_equal(second.reshape(-1), both[10:])
    out = chains.chain_awgn(chains.PhiloxRng(5, 9), "qam", 16, 500, 8.0)
    assert 0 < out["symbol_errors"] < 500 and out["num_bits"] == 2000
