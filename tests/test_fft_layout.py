"""CPU: index algebra of csrc/fft.hpp restated in NumPy -- the DIF / DIT stage sequence, the
digit-reversal maps and the LDS bank swizzle -- so that the device code's structure is checked
without a GPU (the GPU tests check the kernels themselves)."""
import numpy as np


def shape(N):
    L = int(np.log2(N))
    return L // 2, L & 1


def pos_of_index(N, f):
    n4, h2 = shape(N)
    pos, size = 0, N
    for _ in range(n4):
        size >>= 2
        pos += (f & 3) * size
        f >>= 2
    return pos + (f & 1 if h2 else 0)


def swz(e):
    return e ^ (((e >> 4) & 3) * 5) ^ (((e >> 6) & 1) << 4)


def rot(a, inv):
    return a * (1j if inv else -1j)


def dif(x, inv, swizzle=False):
    N = len(x)
    idx = (lambda e: swz(e)) if swizzle else (lambda e: e)
    buf = np.zeros(N, complex)
    for e in range(N):
        buf[idx(e)] = x[e]
    n4, h2 = shape(N)
    tw = np.exp(-2j * np.pi * np.arange(N) / N)
    if inv:
        tw = tw.conj()
    s = N // 4
    for _ in range(n4):
        step = N // (4 * s)
        for bb in range(N // 4):
            k, g = bb & (s - 1), bb // s
            e0 = g * 4 * s + k
            i = [idx(e0 + j * s) for j in range(4)]
            x0, x1, x2, x3 = (buf[j] for j in i)
            a0, a1, a2, a3 = x0 + x2, x0 - x2, x1 + x3, rot(x1 - x3, inv)
            y = [a0 + a2, a1 + a3, a0 - a2, a1 - a3]
            if s > 1:
                for q in (1, 2, 3):
                    y[q] *= tw[q * k * step]
            for j in range(4):
                buf[i[j]] = y[j]
        s >>= 2
    if h2:
        for bb in range(N // 2):
            i0, i1 = idx(2 * bb), idx(2 * bb + 1)
            buf[i0], buf[i1] = buf[i0] + buf[i1], buf[i0] - buf[i1]
    return np.array([buf[idx(e)] for e in range(N)])


def test_dif_matches_numpy_fft_with_and_without_swizzle():
    rs = np.random.RandomState(1)
    for N in (16, 64, 128, 512, 1024):
        x = rs.randn(N) + 1j * rs.randn(N)
        pos = np.array([pos_of_index(N, f) for f in range(N)])
        for swizzle in (False, True) if N >= 128 else (False,):
            assert np.max(np.abs(dif(x, False, swizzle)[pos] - np.fft.fft(x))) < 1e-11
            assert np.max(np.abs(dif(x, True, swizzle)[pos] - np.fft.ifft(x) * N)) < 1e-11


def test_swizzle_is_a_bijection_and_conflict_free():
    e = np.arange(4096)
    assert sorted(swz(e)) == list(e)                      # permutation (within 128-element blocks)
    assert np.all(swz(e) // 128 == e // 128)
    N = 1024
    for s in (256, 64, 16, 4, 1):                         # every radix-4 stage, every half-wave
        for half in range(N // 4 // 32):
            lanes = np.arange(32) + 32 * half
            k, g = lanes & (s - 1), lanes // s
            for j in range(4):
                slots = swz(g * 4 * s + k + j * s) % 32
                assert len(set(slots.tolist())) == 32, (s, half, j)                  # ds_read_b64: 32 lanes over 32 slots
                for q in (0, 16):                                                     # ds_write_b64: 16 lanes over 16 slots
                    assert len(set((slots[q:q + 16] % 16).tolist())) == 16, (s, half, j, q)
    for base in range(0, N, 32):                          # contiguous aligned runs stay conflict free
        assert len(set((swz(base + np.arange(32)) % 32).tolist())) == 32


def test_swizzle_and_digit_reversal_are_linear_over_xor():
    """What lds_swz_r4 (fft.hpp) and the block-aligned symbol scatters of the fused kernels rely on: the swizzle and the
    digit reversal are bit permutations / XOR folds, so f(a ^ b) = f(a) ^ f(b); a butterfly's four positions are
    swz(e0) ^ swz(q s) and a DATA block's sixteen bins swz(rev(bin(d0))) ^ swz(rev(t))."""
    for N in (64, 128, 256, 512, 1024, 2048):
        a = np.arange(N)
        sw = np.array([swz(int(e)) for e in a])
        rv = np.array([pos_of_index(N, int(f)) for f in a])
        assert sorted(sw) == list(a) and sorted(rv) == list(a)
        for b in (1, 2, 3, 5, 16, 21, 48, N // 4, N // 2 + 3, N - 1):
            assert np.array_equal(sw[a ^ b], sw ^ sw[b]), (N, b)
            assert np.array_equal(rv[a ^ b], rv ^ rv[b]), (N, b)
        # radix-4 butterfly: e0 = g 4s + k, k < s, has the two bits of q s clear, so e0 + q s = e0 ^ q s
        s = 1
        while 4 * s <= N:
            for bb in range(0, N // 4, 7):
                k, g = bb & (s - 1), bb // s
                e0 = g * 4 * s + k
                for q in range(4):
                    assert e0 + q * s == e0 ^ (q * s)
                    assert sw[e0 + q * s] == sw[e0] ^ sw[q * s]
            s *= 4
        # full-band scatter on a 16-symbol boundary: bin(d0 + t) = bin(d0) ^ t for d0 a multiple of 16
        for d0 in range(0, N, 16):
            b0 = (d0 + N // 2) % N
            for t in range(16):
                assert (d0 + t + N // 2) % N == b0 ^ t
                assert sw[rv[b0 ^ t]] == sw[rv[b0]] ^ sw[rv[t]]
