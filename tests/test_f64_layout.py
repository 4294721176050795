"""CPU: the LDS layout of the complex128 config-4 kernel (csrc/pipeline_mimo_planar.hip) -- planar doubles, position e
stored at lds_swz64(e) -- replayed under gfx950's bank rules for 8-byte accesses (MI355X_MICROARCH.md, LDS table):
  ds_read_b64 : two groups of 32 lanes, bank pair of an 8-byte slot = slot mod 32   -> 32 distinct slots per group
  ds_write_b64: four groups of 16 consecutive lanes, bank = dword mod 32             -> 16 distinct slots mod 16 per group
for every access of every stage of the kernel (scatter, five DIF stages, channel, five DIT stages, decode).
The complex64 instantiation (planes of floats, same element -> slot maps): ds_read_b32 / ds_write_b32 take two groups of 32
lanes with bank = dword mod 32 (a 2-way store conflict is free) -- for 4-byte slots that IS the read rule above, and the
replays below hold every shape, loads AND stores, to the read rule; measured: 384 conflict cycles per realization of 5 322
active ones (profiles/r04/c4md_pmc_summary.json), the data-dependent table look-ups of the modulator."""
import numpy as np


def swz64(e):
    e = np.asarray(e)
    return e ^ (((e >> 4) & 3) * 5) ^ (((e >> 6) & 1) << 4)


def read_conflicts(slots):
    """extra LDS cycles of one wave-wide ds_read_b64 (slots: 64 eight-byte slot indices)"""
    extra = 0
    for g in (slots[:32], slots[32:]):
        for bank in set(s % 32 for s in g):
            extra += len(set(s for s in g if s % 32 == bank)) - 1
    return extra


def write_conflicts(slots):
    extra = 0
    for q in range(4):
        g = slots[16 * q:16 * q + 16]
        for bank in set(s % 16 for s in g):
            extra += len(set(s for s in g if s % 16 == bank)) - 1
    return extra


def test_swizzle_is_a_bijection_within_128_element_blocks():
    e = np.arange(1024)
    assert sorted(swz64(e)) == list(e)
    assert np.all(swz64(e) // 128 == e // 128)


def test_every_stage_of_the_f64_kernel_is_conflict_free():
    N = 1024
    for wave in range(4):
        tid = 64 * wave + np.arange(64)
        for s in (256, 64, 16, 4, 1):                                  # radix-4 stages (DIF and DIT share positions)
            k, g = tid & (s - 1), tid // s
            e0 = g * 4 * s + k
            for q in range(4):
                slots = list(swz64(e0 + q * s))
                assert read_conflicts(slots) == 0, ("read", s, q, wave)
                assert write_conflicts(slots) == 0, ("write", s, q, wave)
        for it in range(2):                                            # channel: positions p0, p0 + N/4
            j = tid + 256 * it
            half, rest = j // (N // 4), j % (N // 4)
            p0 = 2 * half * (N // 4) + rest
            for p in (p0, p0 + N // 4):
                slots = list(swz64(p))
                assert read_conflicts(slots) == 0 and write_conflicts(slots) == 0
        for it in range(4):                                            # decode: bins of d = tid + 256 it (full band)
            d = tid + 256 * it
            slots = list(swz64((d + N // 2) % N))
            assert read_conflicts(slots) == 0
        for c in range(4):                                             # scatter: thread t fills bins 4 t' + c of its Philox block
            d = 4 * tid + c
            slots = list(swz64((d + N // 2) % N))
            assert write_conflicts(slots) == 0


def _family_spans(N):
    """radix-4 spans in DIF order (N/4 ... 1, or ... 2 followed by the radix-2 stage on a thread's four positions)"""
    log2 = N.bit_length() - 1
    return [(N // 4) >> (2 * st) for st in range(log2 // 2)], bool(log2 & 1)


def test_every_stage_of_the_f64_family_is_conflict_free():
    """round 4: fft_size 256 / 512 / 1024 / 2048, workgroups of (N / 4) x {1, 2} threads (two antennas per thread: one group
    per antenna pair) -- radix-4 stages, the radix-2 stage of 512 / 2048 (the span-1 access pattern), channel pairs
    (p0, p0 + N/4), decode bins, scatter blocks of 16 / Nt subcarriers."""
    for N in (256, 512, 1024, 2048):
        NB = N // 4
        spans, has2 = _family_spans(N)
        assert spans[-1] == (2 if has2 else 1)
        for groups in (1, 2):
            TB = NB * groups
            for wave in range(TB // 64):
                tid = 64 * wave + np.arange(64)
                bb = tid & (NB - 1)
                for s in spans:
                    k, g = bb & (s - 1), bb // s
                    e0 = g * 4 * s + k
                    for q in range(4):
                        slots = list(swz64(e0 + q * s))
                        assert read_conflicts(slots) == 0 and write_conflicts(slots) == 0, (N, s, q, wave)
                if has2:
                    for q in range(4):
                        slots = list(swz64(4 * bb + q))
                        assert read_conflicts(slots) == 0 and write_conflicts(slots) == 0, (N, "r2", q, wave)
                for j0 in range(0, N // 2, TB):                            # channel
                    j = tid + j0
                    half, rest = j // (N // 4), j % (N // 4)
                    p0 = 2 * half * (N // 4) + rest
                    for p in (p0, p0 + N // 4):
                        slots = list(swz64(p))
                        assert read_conflicts(slots) == 0 and write_conflicts(slots) == 0, (N, "chan", wave)
                for d0 in range(0, N, TB):                                 # decode (full band)
                    slots = list(swz64((tid + d0 + N // 2) % N))
                    assert read_conflicts(slots) == 0, (N, "decode", wave)
                for nt in (2, 4):                                          # scatter: block t = 16 / nt subcarriers
                    per = 16 // nt
                    for c in range(per):
                        d = per * tid + c
                        if d.max() < N:
                            slots = list(swz64((d + N // 2) % N))
                            # Nt = 4: conflict free.  Nt = 2 (stride 8 across the lanes): 2-way -- 8 LDS-array cycles
                            # against the 6 a ds_write_b64 takes to issue anyway; 16 of a realization's ~600 LDS accesses
                            assert write_conflicts(slots) <= (0 if nt == 4 else 32), (N, "scatter", nt, c, wave)


def test_the_radix4_swizzle_of_fft_hpp_is_not_enough_for_8_byte_stores():
    """lds_swz (fft.hpp) was built for the read rule; ds_write_b64's 16-lane groups collide on it for spans 4 and 1 --
    the 0.29 conflict fraction the first version of the kernel measured."""
    def swz(e):
        return e ^ (((e >> 5) & 3) * 5) ^ ((e >> 2) & 16)
    tid = np.arange(64)
    bad = 0
    for s in (4, 1):
        k, g = tid & (s - 1), tid // s
        e0 = g * 4 * s + k
        for q in range(4):
            bad += write_conflicts(list(swz(e0 + q * s)))
    assert bad > 0


# ---- variant 4 of the (1024, 4 x 4) kernel: radix-16 passes, one transform per wavefront (csrc/pipeline_mimo_planar.hip) ----
def swz16f(e):
    e = np.asarray(e)
    return e ^ ((e >> 4) & 31) ^ (((e >> 9) & 1) << 4)


def _rot(x, inv):
    return x * (1j if inv else -1j)


def _r4(x, inv):
    a0, a1, a2, a3 = x[0] + x[2], x[0] - x[2], x[1] + x[3], _rot(x[1] - x[3], inv)
    return [a0 + a2, a1 + a3, a0 - a2, a1 - a3]


def _r16_passes(p, inv, dit):
    """The three register passes, lane by lane, with the device's index maps (pass A / B / C of r16_pass, r16_pass_c) and its
    twiddle factorisation w^((k + 64 q) m) = w^(k m) x (16th root)^(q m)."""
    N = 1024
    w = np.exp(-2j * np.pi * np.arange(N) / N)
    tw = (lambda i: np.conj(w[i % N])) if inv else (lambda i: w[i % N])
    p = p.copy()

    def pass_c():
        for gi in range(64):
            for c in range(4):
                e = [16 * gi + 4 * c + m for m in range(4)]
                p[e] = _r4(p[e], inv)

    def pass_16(el, t1, root, t2):
        v = [[p[el(m, q)] for q in range(4)] for m in range(4)]
        if not dit:
            for q in range(4):
                y = _r4([v[m][q] for m in range(4)], inv)
                for m in range(4):
                    v[m][q] = y[m] * tw(t1(m)) * tw(root * q * m)
            for m in range(4):
                z = _r4(v[m], inv)
                for q in range(4):
                    p[el(m, q)] = z[q] * tw(t2(q))
        else:
            for m in range(4):
                v[m] = _r4([v[m][q] * tw(t2(q)) for q in range(4)], inv)
            for q in range(4):
                y = _r4([v[m][q] * tw(t1(m)) * tw(root * q * m) for m in range(4)], inv)
                for m in range(4):
                    p[el(m, q)] = y[m]

    def pass_a():
        for k in range(64):
            pass_16(lambda m, q: k + 64 * q + 256 * m, lambda m: k * m, 64, lambda q: 4 * k * q)

    def pass_b():
        for gi in range(64):
            G, k4 = gi >> 2, gi & 3
            pass_16(lambda m, q: 64 * G + k4 + 4 * q + 16 * m, lambda m: 16 * k4 * m, 64, lambda q: 64 * k4 * q)

    for step in ((pass_c, pass_b, pass_a) if dit else (pass_a, pass_b, pass_c)):
        step()
    return p


def _pos_of_index(n, f):
    pos, size = 0, n
    while size > 1:
        size >>= 2
        pos += (f & 3) * size
        f >>= 2
    return pos


def test_radix16_passes_are_the_transform_in_the_radix4_arrangement():
    rs = np.random.RandomState(5)
    x = rs.randn(1024) + 1j * rs.randn(1024)
    perm = np.array([_pos_of_index(1024, f) for f in range(1024)])        # fft.hpp fft_pos_of_index<1024>
    assert np.allclose(_r16_passes(x, False, False)[perm], np.fft.fft(x))
    assert np.allclose(_r16_passes(x, True, False)[perm], np.fft.ifft(x) * 1024)
    xs = np.empty(1024, complex)
    xs[perm] = x
    assert np.allclose(_r16_passes(xs, False, True), np.fft.fft(x))
    assert np.allclose(_r16_passes(xs, True, True), np.fft.ifft(x) * 1024)


def test_radix16_swizzle_meets_both_bank_rules_for_every_access_shape():
    assert sorted(swz16f(np.arange(1024))) == list(range(1024))
    # XOR-linearity (the device adds offsets by XOR on the swizzled base): base and offset occupy disjoint bits
    for gi in range(64):
        for q in range(4):
            for m in range(4):
                assert swz16f(gi + 64 * q + 256 * m) == swz16f(gi) ^ swz16f(64 * q + 256 * m)
                b = 64 * (gi >> 2) + (gi & 3)
                assert swz16f(b + 4 * q + 16 * m) == swz16f(b) ^ swz16f(4 * q + 16 * m)
                assert swz16f(16 * gi + 4 * q + m) == swz16f(16 * gi) ^ swz16f(4 * q + m)
    lane = np.arange(64)
    shapes = []
    shapes += [lane + 64 * q + 256 * m for q in range(4) for m in range(4)]                               # pass A
    shapes += [64 * (lane >> 2) + (lane & 3) + 4 * q + 16 * m for q in range(4) for m in range(4)]        # pass B
    shapes += [16 * lane + 4 * c + m for c in range(4) for m in range(4)]                                 # pass C
    shapes += [np.arange(b, b + 64) for b in range(0, 1024, 64)]                                          # decode, channel p0
    shapes += [(2 * (j // 256) * 256 + j % 256) + off for b in range(0, 512, 64) for off in (0, 256)
               for j in [np.arange(b, b + 64)]]                                                           # channel pairs
    for w in range(4):                                    # the fused middle stage: thread -> group g, positions 4 g + d
        g = (lane & 15) | (w << 4) | (((lane >> 5) & 1) << 6) | (((lane >> 4) & 1) << 7)
        assert w > 0 or sorted(set(int(v) for ww in range(4) for v in ((lane & 15) | (ww << 4) | (((lane >> 5) & 1) << 6) |
                                                                        (((lane >> 4) & 1) << 7)))) == list(range(256))
        assert np.array_equal(g[32:], g[:32] ^ 64)        # lanes l, l ^ 32: groups g, g ^ 64 = time samples m, m + 1
        shapes += [4 * g + d for d in range(4)]
    for e in shapes:
        slots = list(swz16f(e))
        assert read_conflicts(slots) == 0 and write_conflicts(slots) == 0
    for c in range(4):                                                                                    # scatter, Nt = 4
        for wave in range(4):
            d = 4 * (64 * wave + lane) + c
            assert write_conflicts(list(swz16f((d + 512) % 1024))) == 0
    # the aligned scatter's pos0 ^ t rule: sixteen / Nt consecutive bins from a multiple of it differ in bits the fold leaves alone
    for bin0 in range(0, 1024, 4):
        for t in range(4):
            assert swz16f(bin0 + t) == swz16f(bin0) ^ t


# ---- config 3 with one realization per wavefront (csrc/siso_tdl_wave.hpp: k_run_ofdm_tdl_wave): its index maps, restated ----
def test_wave_kernel_position_to_bin_map_and_inverse_band_map():
    """The equaliser walks POSITIONS p = lane + 64 k: bin f = F(lane) | F(64 k) with F(64 k) = ((k & 3) << 2) | (k >> 2), and
    the data index d of bin f is the inverse of fft.hpp's ofdm_bin (full band and partial band)."""
    def index_of_pos(p):                         # fft.hpp fft_index_of_pos<1024>
        f, size, mul = 0, 1024, 1
        for _ in range(5):
            size >>= 2
            q, p = divmod(p, size)
            f += q * mul
            mul <<= 2
        return f

    def ofdm_bin(d, n, used):                    # fft.hpp ofdm_bin
        if used == n:
            return (d + n // 2) % n
        h = used // 2
        return n - h + d if d < h else 1 + (d - h)

    for lane in range(64):
        f_lane = ((lane & 3) << 8) | (((lane >> 2) & 3) << 6) | (((lane >> 4) & 3) << 4)
        for k in range(16):
            f = f_lane | ((k & 3) << 2) | (k >> 2)
            assert f == index_of_pos(lane + 64 * k)
            assert _pos_of_index(1024, f) == lane + 64 * k
            assert swz16f(lane + 64 * k) == swz16f(lane) ^ swz16f(64 * k)
    N = 1024
    for U in (1024, 1000, 608, 2):
        hU, seen = U // 2, []
        for f in range(N):
            if U == N:
                d = (f + N // 2) & (N - 1)
            elif f >= N - hU:
                d = f - (N - hU)
            elif 1 <= f <= hU:
                d = hU + f - 1
            else:
                continue
            assert ofdm_bin(d, N, U) == f
            seen.append(d)
        assert sorted(seen) == list(range(U))


def test_wave_kernel_noise_pairing_draws_every_block_once():
    """Lanes l (even), l + 1 hold samples m, m + 1 of combination c (m = l + 64 c): with an even stream offset they share NOISE
    block (offset + m) >> 1.  The even lane draws combinations 0 .. 7, the odd lane 8 .. 15; after the swap every (lane,
    combination) has the half of the block that is its own sample, and every block of the symbol is drawn exactly once."""
    for nbase in (16, 1040 + 16, 0):             # even offsets (sym0 + cp)
        drawn = {}
        got = {}
        for lane in range(64):
            odd = lane & 1
            for j in range(8):
                c = j + 8 * odd
                m = lane + 64 * c
                blk = (nbase + m) >> 1
                assert blk not in drawn
                drawn[blk] = (lane, c)
                za, zb = (blk, 0), (blk, 1)       # the block's first and second sample
                partner = lane ^ 1
                if odd:
                    got[(partner, c)] = za       # sent: the even partner's sample of combination 8 + j
                    got[(lane, c)] = zb
                else:
                    got[(lane, c)] = za
                    got[(partner, c)] = zb
        assert len(drawn) == 512 and sorted(drawn) == list(range(nbase >> 1, (nbase >> 1) + 512))
        for lane in range(64):
            for c in range(16):
                i0 = nbase + lane + 64 * c
                assert got[(lane, c)] == (i0 >> 1, i0 & 1)


def test_wave_kernel_prefixed_delay_line():
    """xp[P + m] = x[m], xp[j] = x[N - P + j] (written by the lanes whose sample index reaches into the last P): the read
    xp[P + m - d] is x[(m - d) mod N] for every delay d <= P, and the reads of a wavefront are consecutive words."""
    N = 1024
    x = np.arange(N)
    for P in (16, 48, 64, 256):
        xp = np.full(N + P, -1)
        for lane in range(64):
            for c in range(16):
                m = lane + 64 * c
                xp[P + m] = x[m]
                if c >= 12 and m >= N - P:
                    xp[m - (N - P)] = x[m]
        assert (xp >= 0).all()
        for d in (0, 1, P // 2, P):
            for c in range(16):
                lanes = np.arange(64)
                src = P + lanes - d + 64 * c
                assert np.array_equal(xp[src], x[(lanes + 64 * c - d) % N])
                assert np.array_equal(np.diff(src), np.ones(63, dtype=int))
