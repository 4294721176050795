"""CPU model of the matrix-core FFT-1024 of csrc/pipeline_mimo_mfma.hip (16 x 16 x 4, DFT-16 passes as MFMA products).

What is checked here, without a GPU:
  * the algorithm and every lane <-> element map, by emulating v_mfma_f32_16x16x4_f32 in NumPy with the documented
    operand maps (A[i][k] <- lane i + 16k, B[k][j] <- lane j + 16k, D[row][col] -> lane col + 16 (row / 4), register
    row % 4; confirmed on the device by scripts/experiments/mfma_probe.hip) and comparing the three-pass transforms
    (DIF, DIT, inverse by the re<->im swap identity) with numpy.fft;
  * that the kernel's closed-form LDS offsets (p1_ld / p1_st / p2_ld / p2_st / mid_off) equal the swizzle
    f16_pos(p) = p ^ (f(p >> 6) << 2) they were derived from;
  * that every LDS access pattern of the kernel is bank-conflict free under the gfx950 rules of
    /opt/skills/guides/MI355X_MICROARCH.md (LDS section), except the b128 stores of the middle stage (2-way).
"""
import numpy as np

N = 1024
PLANE, ANT = 1040, 2080


def f(k):
    return (k & 7) ^ ((k & 1) << 3)


def swz(k):
    return f(k) << 2


def pos(p):
    return p ^ swz(p >> 6)


def mfma16(a, b, c):
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a[l]
        B[l >> 4, l & 15] = b[l]
    D = A @ B
    out = c.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[4 * (l >> 4) + r, l & 15]
    return out


def mats():
    ae, ao = np.zeros((64, 4)), np.zeros((64, 4))
    for l in range(64):
        i, g = l & 15, l >> 4
        u = i >> 1
        for t in range(4):
            e = 2 * t + (g >> 1)
            we = np.exp(-2j * np.pi * e * u / 8)
            wo = np.exp(-2j * np.pi * e * (2 * u + 1) / 16)
            for wv, arr in ((we, ae), (wo, ao)):
                if (i & 1) == 0:
                    arr[l, t] = -wv.imag if (g & 1) else wv.real
                else:
                    arr[l, t] = wv.real if (g & 1) else wv.imag
    return ae, ao


AE, AO = mats()


def dft16(load):
    """-> out[lane][x] complex = output 4g + x of group lane & 15 (dft16_mfma of the kernel)"""
    b = np.array([[load(l, t) for t in range(8)] for l in range(64)])
    ce, co = np.zeros((64, 4)), np.zeros((64, 4))
    for t in range(4):
        ce = mfma16(AE[:, t], b[:, t] + b[:, t + 4], ce)
        co = mfma16(AO[:, t], b[:, t] - b[:, t + 4], co)
    out = np.zeros((64, 4), complex)
    out[:, 0] = ce[:, 0] + 1j * ce[:, 1]
    out[:, 1] = co[:, 0] + 1j * co[:, 1]
    out[:, 2] = ce[:, 2] + 1j * ce[:, 3]
    out[:, 3] = co[:, 2] + 1j * co[:, 3]
    return out


def lane_consts(w, l):
    j, g = l & 15, l >> 4
    gb = g >> 1
    n2 = 16 * w + j
    k1p, m2p = 4 * w + (j >> 2), j & 3
    p2_base = 64 * k1p + (m2p | swz(k1p))
    return dict(j=j, g=g, gb=gb, n2=n2, k1p=k1p, m2p=m2p, plane=(g & 1) * PLANE,
                p1_ld=64 * gb + (n2 ^ (gb * 36)), p1_st=256 * g + (n2 ^ (16 * (g & 1))),
                p2_ld=p2_base ^ (4 * gb), p2_st=p2_base ^ (16 * g))


def p1_load_off(c, t):
    return c["plane"] + (c["p1_ld"] ^ (((2 * t) & 7) << 2)) + 128 * t


def p1_store_off(c, x):
    return (c["p1_st"] ^ ((x << 2) ^ ((x & 1) << 5))) + 64 * x


def mid_consts(w, l):
    kk = ((l >> 5) << 1) | (l & 1)
    j1 = (l >> 1) & 15
    k1 = 4 * w + kk
    return k1, j1, 64 * k1 + ((4 * j1) ^ swz(k1))


def dif(d, sign_fold=False):
    """d: one antenna's [re plane | im plane]; natural input -> position 64 k1 + 4 j1 + j2 holds X[k1 + 16 j1 + 256 j2]"""
    for w in range(4):                                   # P1 (wave-local in place: loads before stores)
        cs = [lane_consts(w, l) for l in range(64)]
        o = dft16(lambda l, t: d[p1_load_off(cs[l], t)])
        for l in range(64):
            for x in range(4):
                v = o[l, x] * np.exp(-2j * np.pi * (4 * cs[l]["g"] + x) * cs[l]["n2"] / 1024)
                off = p1_store_off(cs[l], x)
                d[off], d[off + PLANE] = v.real, v.imag
    for w in range(4):                                   # P2
        cs = [lane_consts(w, l) for l in range(64)]
        o = dft16(lambda l, t: d[cs[l]["plane"] + (cs[l]["p2_ld"] ^ (8 * t))])
        for l in range(64):
            for x in range(4):
                v = o[l, x] * np.exp(-2j * np.pi * (4 * cs[l]["g"] + x) * cs[l]["m2p"] / 64)
                if sign_fold and (cs[l]["k1p"] & 1) and (cs[l]["m2p"] & 1):
                    v = -v
                off = cs[l]["p2_st"] ^ (4 * x)
                d[off], d[off + PLANE] = v.real, v.imag


def test_address_formulas_equal_the_swizzle():
    for w in range(4):
        for l in range(64):
            c = lane_consts(w, l)
            for t in range(8):
                e = 2 * t + c["gb"]
                assert p1_load_off(c, t) == c["plane"] + pos(64 * e + c["n2"])
                assert c["plane"] + (c["p2_ld"] ^ (8 * t)) == c["plane"] + pos(64 * c["k1p"] + 4 * e + c["m2p"])
            for x in range(4):
                k = 4 * c["g"] + x
                assert p1_store_off(c, x) == pos(64 * k + c["n2"])
                assert c["p2_st"] ^ (4 * x) == pos(64 * c["k1p"] + 4 * k + c["m2p"])
            k1, j1, off = mid_consts(w, l)
            assert [off + m for m in range(4)] == [pos(64 * k1 + 4 * j1 + m) for m in range(4)]
    # the middle stage covers every butterfly of the wave's rows exactly once; lanes l, l ^ 1 are time neighbours
    for w in range(4):
        seen = {mid_consts(w, l)[:2] for l in range(64)}
        assert seen == {(k1, j1) for k1 in range(4 * w, 4 * w + 4) for j1 in range(16)}
        for l in range(0, 64, 2):
            a, b = mid_consts(w, l), mid_consts(w, l + 1)
            assert b[0] == a[0] + 1 and b[1] == a[1] and a[0] % 2 == 0


def _planes(x):
    d = np.zeros(ANT)
    for n in range(N):
        d[pos(n)], d[PLANE + pos(n)] = x[n].real, x[n].imag
    return d


def test_dif_dit_and_swap_inverse_against_numpy_fft():
    rng = np.random.default_rng(7)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    X = np.fft.fft(x)
    d = _planes(x)
    dif(d)
    got = np.zeros(N, complex)
    for q in range(256):                                 # P3: DFT-4 over the 4 consecutive positions
        k1, j1 = q >> 4, q & 15
        offs = [pos(4 * q + m) for m in range(4)]
        y = np.fft.fft(np.array([d[o] + 1j * d[o + PLANE] for o in offs]))
        for c in range(4):
            got[k1 + 16 * j1 + 256 * c] = y[c]
    assert np.max(np.abs(got - X)) < 1e-10
    # DIT from the digit-reversed arrangement: P3' (x W64), P2' (x W1024), P1'
    d = np.zeros(ANT)
    for k1 in range(16):
        for j1 in range(16):
            for j2 in range(4):
                p = pos(64 * k1 + 4 * j1 + j2)
                d[p], d[p + PLANE] = x[k1 + 16 * j1 + 256 * j2].real, x[k1 + 16 * j1 + 256 * j2].imag
    for q in range(256):
        j1 = q & 15
        offs = [pos(4 * q + m) for m in range(4)]
        y = np.fft.fft(np.array([d[o] + 1j * d[o + PLANE] for o in offs])) * np.exp(-2j * np.pi * np.arange(4) * j1 / 64)
        for m in range(4):
            d[offs[m]], d[offs[m] + PLANE] = y[m].real, y[m].imag
    for w in range(4):
        cs = [lane_consts(w, l) for l in range(64)]
        o = dft16(lambda l, t: d[cs[l]["plane"] + (cs[l]["p2_ld"] ^ (8 * t))])
        for l in range(64):
            for xo in range(4):
                m1 = 4 * cs[l]["g"] + xo
                v = o[l, xo] * np.exp(-2j * np.pi * (4 * m1 + cs[l]["m2p"]) * cs[l]["k1p"] / 1024)
                off = cs[l]["p2_st"] ^ (4 * xo)
                d[off], d[off + PLANE] = v.real, v.imag
    out = np.zeros(N, complex)
    for w in range(4):
        cs = [lane_consts(w, l) for l in range(64)]
        o = dft16(lambda l, t: d[p1_load_off(cs[l], t)])
        for l in range(64):
            for xo in range(4):
                out[64 * (4 * cs[l]["g"] + xo) + cs[l]["n2"]] = o[l, xo]
    assert np.max(np.abs(out - X)) < 1e-10
    # inverse by the swap identity, with the slot-order sign folded into P2 for odd rows (kernel's middle stage)
    d = _planes(X.imag + 1j * X.real)
    dif(d, sign_fold=True)
    for q in range(256):
        k1, j1 = q >> 4, q & 15
        offs = [pos(4 * q + m) for m in range(4)]
        y = np.fft.fft(np.array([d[o] + 1j * d[o + PLANE] for o in offs]))
        for s in range(4):
            c = s ^ (2 * (k1 & 1))
            assert abs((y[s].imag + 1j * y[s].real) / N - x[k1 + 16 * j1 + 256 * c]) < 1e-12


G32 = [range(0, 32), range(32, 64)]
G128R = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
         [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
         [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]
G128W = [range(8 * i, 8 * i + 8) for i in range(8)]


def _ways(addrs, groups, nbanks, width=1):
    worst = 0
    for grp in groups:
        banks = {}
        for l in grp:
            for k in range(width):
                banks.setdefault((addrs[l] + k) % nbanks, set()).add(addrs[l] + k)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def test_lds_accesses_are_bank_conflict_free():
    worst = {}
    for w in range(4):
        cs = [lane_consts(w, l) for l in range(64)]
        for a in range(4):
            for t in range(8):
                worst["p1 load"] = max(worst.get("p1 load", 0), _ways([a * ANT + p1_load_off(c, t) for c in cs], G32, 32))
                worst["p2 load"] = max(worst.get("p2 load", 0),
                                       _ways([a * ANT + c["plane"] + (c["p2_ld"] ^ (8 * t)) for c in cs], G32, 32))
            for x in range(4):
                for part in (0, PLANE):
                    worst["p1 store"] = max(worst.get("p1 store", 0),
                                            _ways([a * ANT + part + p1_store_off(c, x) for c in cs], G32, 32))
                    worst["p2 store"] = max(worst.get("p2 store", 0),
                                            _ways([a * ANT + part + (c["p2_st"] ^ (4 * x)) for c in cs], G32, 32))
            for part in (0, PLANE):
                mid = [a * ANT + part + mid_consts(w, l)[2] for l in range(64)]
                assert all(m % 4 == 0 for m in mid)
                worst["mid load"] = max(worst.get("mid load", 0), _ways(mid, G128R, 64, 4))
                worst["mid store"] = max(worst.get("mid store", 0), _ways(mid, G128W, 32, 4))
                sc = [a * ANT + part + pos((4 * (64 * w + l) + 512) & 1023) for l in range(64)]
                worst["scatter"] = max(worst.get("scatter", 0), _ways(sc, G128W, 32, 4))
    assert worst == {"p1 load": 1, "p2 load": 1, "p1 store": 1, "p2 store": 1, "mid load": 1, "mid store": 2,
                     "scatter": 1}, worst
