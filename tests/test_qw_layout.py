"""CPU: the arithmetic skeleton and the LDS layouts of the quarter-wave config-4 kernel (csrc/pipeline_mimo_qw.hip), replayed in NumPy:
  * the decomposition of the 1024-point transforms -- wavefront j = time class n mod 4: first DIF stage evaluated per class from the
    bins, 256-point radix-4 stages in place, the digit-reversed sample map qw_mtime, the last DIT stage across the classes;
  * the label array (writer: one DATA block per thread; readers: a lane's 64 labels, a decode thread's 16) -- the same symbol at both ends;
  * the noise ledger: every Philox block of a realization evaluated exactly once, by the wavefront and lane the kernel says;
  * the transposition planes and the exchange under gfx950's bank rules for 8-byte accesses (as tests/test_f64_layout.py):
    ds_write_b64 four groups of 16 lanes, ds_read_b64 two groups of 32 lanes, bank pair = slot mod 32."""
import numpy as np

N = 1024
W = np.exp(-2j * np.pi / N)


def digitrev4(p, nd):
    r = 0
    for _ in range(nd):
        r = r * 4 + (p & 3)
        p >>= 2
    return r


def qw_slot(a, e):
    return a * 272 + e + (e >> 4)


def qw_mtime(h, c):
    return (c & 3) * 64 + (c >> 2) * 16 + (h & 3) * 4 + (h >> 2)


def _stage(v, s, inv, dit):
    """one in-place radix-4 stage of span s over a 256-vector: DIF = butterfly then twiddle, DIT = twiddle then butterfly"""
    sgn = 1.0 if inv else -1.0
    out = v.copy()
    B = 4 * s
    for base in range(0, len(v), B):
        for k in range(s):
            tw = [np.exp(sgn * 2j * np.pi * k * t / B) for t in range(4)]
            u = [v[base + k + t * s] * (tw[t] if dit else 1.0) for t in range(4)]
            for tp in range(4):
                acc = sum(u[t] * np.exp(sgn * 2j * np.pi * t * tp / 4) for t in range(4))
                out[base + k + tp * s] = acc * (1.0 if dit else tw[tp])
    return out


def test_quarter_wave_decomposition_is_the_1024_point_transform():
    rng = np.random.default_rng(1)
    X = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    x = np.zeros(N, complex)
    kp = np.arange(256)
    for j in range(4):
        v = np.conj(W) ** (j * kp) * sum((1j) ** (j * q) * X[kp + 256 * q] for q in range(4))      # first DIF stage, output j only
        for s in (64, 16, 4, 1):
            v = _stage(v, s, True, False)
        for h in range(16):
            for c in range(16):
                x[4 * qw_mtime(h, c) + j] = v[16 * h + c]              # register c of lane group h after the second pass
    assert np.max(np.abs(x - np.fft.ifft(X) * N)) < 1e-10
    y = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    T = np.zeros((4, 256), complex)
    for j in range(4):
        v = np.array([y[4 * qw_mtime(p >> 4, p & 15) + j] for p in range(256)])
        for s in (1, 4, 16, 64):
            v = _stage(v, s, False, True)
        T[j] = v
    Y = np.zeros(N, complex)
    for t in range(256):                                                # the decode thread that owns k' = t
        u = [T[j][t] * W ** (j * t) for j in range(4)]
        for q in range(4):
            Y[t + 256 * q] = sum(u[j] * (-1j) ** (j * q) for j in range(4))
    assert np.max(np.abs(Y - np.fft.fft(y))) < 1e-10
    assert all(qw_mtime(p >> 4, p & 15) == digitrev4(p, 4) for p in range(256))


def test_label_array_writer_and_readers_agree():
    stride = 80
    lab = np.full(4 * 16 * stride, -1, dtype=np.int64)
    for t in range(256):                                                # thread t draws block t: subcarriers 4 t .. 4 t + 3
        q, u = (t >> 6) ^ 2, (t & 63) >> 2
        for s in range(4):
            g = 4 * (t & 3) + s
            for a in range(4):
                nl = 16 * t + 4 * s + a                                 # symbol index: antenna nl % 4, subcarrier nl / 4
                lab[(a * 16 + g) * stride + 4 * u + q] = nl
    for a in range(4):
        for g in range(16):                                             # first-stage lane (a, g): word u = labels of bins g + 16 u + 256 q
            for u in range(16):
                for q in range(4):
                    k = g + 16 * u + 256 * q
                    d = k ^ 512                                         # full band: bin k carries subcarrier (k + N / 2) mod N
                    assert lab[(a * 16 + g) * stride + 4 * u + q] == 4 * d + a
    for t in range(256):                                                # decode thread t: byte q of word a = bin t + 256 q
        for a in range(4):
            for q in range(4):
                assert lab[(a * 16 + (t & 15)) * stride + 4 * (t >> 4) + q] == 4 * ((t + 256 * q) ^ 512) + a
    # bank check of the byte stores: per instruction (s, a) the 64 lanes of a wavefront hit 64 different banks
    for w in range(4):
        t = 64 * w + np.arange(64)
        for s in range(4):
            addr = (4 * (t & 3) + s) * stride + 4 * ((t & 63) >> 2) + ((t >> 6) ^ 2)
            assert len(set((addr // 4) % 64)) == 64


def test_noise_blocks_are_evaluated_once_each():
    cp, n_sym = 16, 1
    row = n_sym * (N + cp)
    seen = {}
    for j in range(4):
        pj = j & 1
        for lane in range(64):
            r, h = lane >> 4, lane & 15
            for cc in range(8):
                c = 8 * pj + cc
                i0 = r * row + cp + 4 * qw_mtime(h, c) + (j & 2)        # the pair's even sample
                assert i0 % 2 == 0
                blk = i0 >> 1
                assert blk not in seen
                seen[blk] = (j, lane, c)
    assert len(seen) == 4 * N // 2                                      # 2 048 blocks = 4 096 noise samples
    # ... and the words reach the right sample: wavefront j, lane, slot c holds sample r row + cp + 4 mtime + j
    for (blk, (j, lane, c)) in seen.items():
        for jj in (j, j ^ 1):
            n = 4 * qw_mtime(lane & 15, c) + jj
            assert ((lane >> 4) * row + cp + n) >> 1 == blk


def _read_conflicts(slots):
    extra = 0
    for g in (slots[:32], slots[32:]):
        for bank in set(s % 32 for s in g):
            extra += len(set(s for s in g if s % 32 == bank)) - 1
    return extra


def _write_conflicts(slots):
    extra = 0
    for q in range(4):
        g = slots[16 * q:16 * q + 16]
        for bank in set(s % 16 for s in g):
            extra += len(set(s for s in g if s % 16 == bank)) - 1
    return extra


def test_transposition_and_exchange_are_conflict_free_and_consistent():
    lanes = np.arange(64)
    a, g = lanes >> 4, lanes & 15
    plane = {}
    for u in range(16):                                                  # pass-1 layout: lane (a, g) register u = element g + 16 u
        slots = [qw_slot(int(a[l]), int(g[l]) + 16 * u) for l in range(64)]
        assert _write_conflicts(slots) == 0 and _read_conflicts(slots) == 0
        assert slots == [int(qw_slot(int(a[l]), int(g[l]))) + 17 * u for l in range(64)]       # base + compile-time offset
        for l in range(64):
            plane[slots[l]] = (int(a[l]), int(g[l]) + 16 * u)
    assert len(plane) == 1024 and max(plane) < 4 * 272
    for c in range(16):                                                  # pass-2 layout: lane (a, h) register c = element 16 h + c
        slots = [qw_slot(int(a[l]), 16 * int(g[l]) + c) for l in range(64)]
        assert _read_conflicts(slots) == 0 and _write_conflicts(slots) == 0
        assert slots == [int(qw_slot(int(a[l]), 16 * int(g[l]))) + c for l in range(64)]
        for l in range(64):
            assert plane[slots[l]] == (int(a[l]), 16 * int(g[l]) + c)
    for u in range(16):                                                  # exchange: lane (r, g) writes k' = g + 16 u at r 256 + k'
        slots = [int(a[l]) * 256 + int(g[l]) + 16 * u for l in range(64)]
        assert _write_conflicts(slots) == 0
    for w in range(4):                                                   # ... and thread t reads [j][r][t]
        t = 64 * w + lanes
        for jj in range(4):
            for rr in range(4):
                assert _read_conflicts([jj * 4 * 272 + rr * 256 + int(x) for x in t]) == 0
