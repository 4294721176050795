"""CPU: the index arithmetic of the round-6 kernels replayed in NumPy (no GPU, no library call):
  * csrc/pipeline_mimo_pw.hip / pipeline_mimo_fw.hip -- the decomposition over NW = fft_size / 256 wavefronts per realization for NW = 1,
    2, 4, 8: first radix-NW DIF stage per time class (the radix-8 form (A0 + i^J A2) + w^J (A1 + i^J A3) literally), 256-point
    transforms, the sample map, the last radix-NW DIT stage as the reading lane evaluates it for ITS 16 / NW elements (radix-8: two
    4-point transforms and the w'^q rotations literally) -- and which lane ends up with which bins;
  * the label array: one DATA block per thread in, sixteen labels per (antenna, group) row for the first stage, sixteen CONSECUTIVE
    bytes per reading lane for the decode -- the same symbol at all three ends; the 2 x 2 form of the full-wave kernel (two
    realizations per wavefront) likewise;
  * the noise ledger: every Philox NOISE block of a realization evaluated exactly once, by the wavefront / lane / register the kernels
    name, for NW = 1 (lanes l, l ^ 4) and NW = 2, 4, 8 (wavefronts j, j ^ 1);
  * the block-diagonal A operand of the 2 x 2 contractions under v_mfma_f64_4x4x4's lane maps;
  * csrc/walk_f64.hpp -- the packed walk's index space (pair -> realization, at most two realizations per pass), the label runs (<= 9
    DATA blocks, byte offsets inside 144), the noise block of a lane pair, the wave total by row_shr / row_bcast adds, and the level-domain
    error count of four QAM decisions (qam_count4) against the label-domain count."""
import numpy as np


def pw_mtime(h, c):
    return (c & 3) * 64 + (c >> 2) * 16 + (h & 3) * 4 + (h >> 2)


def _first_stage(X, NW, j):
    """Z_j[k'] = conj(W_N)^(j k') sum_q e^(+2 pi i j q / NW) X[k' + 256 q], as the kernels group it"""
    N = 256 * NW
    kp = np.arange(256)
    Xq = [X[kp + 256 * q] for q in range(NW)]
    if NW == 1:
        S = Xq[0]
    elif NW == 2:
        S = Xq[0] - Xq[1] if j else Xq[0] + Xq[1]
    elif NW == 4:
        A = Xq[0] - Xq[2] if j & 1 else Xq[0] + Xq[2]
        B = Xq[1] - Xq[3] if j & 1 else Xq[1] + Xq[3]
        S = [A + B, A + 1j * B, A - B, A - 1j * B][j]
    else:
        A = [Xq[q] - Xq[q + 4] if j & 1 else Xq[q] + Xq[q + 4] for q in range(4)]
        rot = (1j) ** (j & 3)
        B0, B1 = A[0] + rot * A[2], A[1] + rot * A[3]
        h = 0.70710678118654752440
        wj = {0: 1, 2: 1j, 4: -1, 6: -1j, 1: h * (1 + 1j), 3: h * (-1 + 1j), 5: h * (-1 - 1j), 7: h * (1 - 1j)}[j]
        S = B0 + wj * B1
    return np.exp(2j * np.pi * j * kp / N) * S


def _last_stage(Yj, NW, kp):
    """Y[k' + 256 q], q = 0 .. NW - 1, from the NW partial transforms at k' (forward), as the reading lane forms them"""
    N = 256 * NW
    x = [Yj[jj] * np.exp(-2j * np.pi * jj * kp / N) for jj in range(NW)]
    if NW == 1:
        return [x[0]]
    if NW == 2:
        return [x[0] + x[1], x[0] - x[1]]
    f4 = lambda u: [u[0] + u[1] + u[2] + u[3], (u[0] - u[2]) - 1j * (u[1] - u[3]), (u[0] + u[2]) - (u[1] + u[3]), (u[0] - u[2]) + 1j * (u[1] - u[3])]
    if NW == 4:
        return f4(x)
    E, O = f4([x[0], x[2], x[4], x[6]]), f4([x[1], x[3], x[5], x[7]])
    h = 0.70710678118654752440
    t = [O[0], h * (1 - 1j) * O[1], -1j * O[2], h * (-1 - 1j) * O[3]]
    return [E[q] + t[q] for q in range(4)] + [E[q] - t[q] for q in range(4)]


def test_part_wave_decomposition_is_the_transform_for_every_size():
    rng = np.random.default_rng(5)
    for NW in (1, 2, 4, 8):
        N = 256 * NW
        X = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        x = np.zeros(N, complex)
        for j in range(NW):
            v = np.fft.ifft(_first_stage(X, NW, j)) * 256               # the wavefront's 256-point inverse transform
            for h in range(16):
                for c in range(16):                                     # register c of lane group h holds sample time NW * mtime + j
                    m = pw_mtime(h, c)
                    x[NW * m + j] = v[m]
        assert np.max(np.abs(x - np.fft.ifft(X) * N)) < 1e-9 * N, NW
        assert sorted(pw_mtime(h, c) for h in range(16) for c in range(16)) == list(range(256))
        # receive: per time class a forward 256-point transform, then the last stage for the reading lane's elements
        y = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        Yj = [np.fft.fft(y[j::NW]) for j in range(NW)]
        want = np.fft.fft(y)
        UU = 16 // NW
        seen = np.zeros(N, int)
        for jw in range(NW):                                            # wavefront jw, lane (r, g): k' = g + 16 (UU jw + uu)
            for g in range(16):
                for uu in range(UU):
                    kp = g + 16 * (UU * jw + uu)
                    out = _last_stage([Y[kp] for Y in Yj], NW, kp)
                    for q in range(NW):                                 # register q + NW uu = bin k' + 256 q
                        assert abs(out[q] - want[kp + 256 * q]) < 1e-9 * N
                        seen[kp + 256 * q] += 1
        assert np.all(seen == 1), NW                                    # every bin decoded by exactly one (wavefront, lane, register)


def test_label_array_writer_and_both_readers_agree():
    """Thread t of 64 NW: DATA block t = subcarriers d = 4 t + s (word s), antenna a (byte a); bin k = d ^ (N / 2) (full band) ->
    byte [a][g = k' mod 16][q + NW u], k' = k mod 256 = g + 16 u, q = k div 256.  First stage: lane (a, g) reads the row; decode: lane
    (a, g) of wavefront jw reads the sixteen bytes at 16 jw = labels of the bins q + NW uu -> k' + 256 q, u = (16 / NW) jw + uu."""
    for NW in (1, 2, 4, 8):
        N, stride = 256 * NW, 16 * NW + (16 if NW > 1 else 0)
        lab = np.full((64, stride), -1, int)
        truth = {}
        for t in range(64 * NW):
            q, u = (t >> 6) ^ (NW // 2), (t & 63) >> 2
            if NW == 1:
                q, u = 0, (t >> 2) ^ 8
            for s in range(4):
                d = 4 * t + s
                k = d ^ (N // 2)
                g = 4 * (t & 3) + s
                assert (k & 255) == g + 16 * u and (k >> 8) == q          # the kernel's shortcut for (g, u, q)
                for a in range(4):
                    assert lab[a * 16 + g, NW * u + q] == -1
                    lab[a * 16 + g, NW * u + q] = 1000 * a + d          # "the label of antenna a on subcarrier d"
                    truth[(a, k)] = 1000 * a + d
        assert np.all(lab[:, :16 * NW] >= 0)
        UU = 16 // NW
        for a in range(4):
            for g in range(16):
                for u in range(16):                                     # first stage: bytes q + NW u of the row
                    for q in range(NW):
                        assert lab[a * 16 + g, q + NW * u] == truth[(a, g + 16 * u + 256 * q)]
                for jw in range(NW):                                    # decode: sixteen consecutive bytes at 16 jw
                    row = lab[a * 16 + g, 16 * jw:16 * jw + 16]
                    for uu in range(UU):
                        for q in range(NW):
                            assert row[q + NW * uu] == truth[(a, g + 16 * (UU * jw + uu) + 256 * q)]


def test_full_wave_two_by_two_labels_and_block_diagonal_contraction():
    """NA = 2: lane t = (rz, tb): realization rz = t div 32 draws DATA block tb of ITS symbol; label 4 s + b of the block sits at stream
    position p = 16 tb + 4 s + b = d NT + a.  Consumer lane = (rz NT + a) 16 + g reads sixteen bytes [k div 16], k = d ^ 128."""
    NT = 2
    lab = np.full((64, 16), -1, int)
    for t in range(64):
        rz, tb = t // 32, t % 32
        for s in range(4):
            for b in range(4):
                p = 16 * tb + 4 * s + b
                d, a = p // NT, p % NT
                k = d ^ 128
                row = (rz * NT + a) * 16 + (k & 15)
                assert lab[row, k >> 4] == -1
                lab[row, k >> 4] = 100000 * rz + 1000 * a + k
    for rz in range(2):
        for a in range(2):
            for g in range(16):
                for u in range(16):
                    assert lab[(rz * 2 + a) * 16 + g, u] == 100000 * rz + 1000 * a + g + 16 * u
    # v_mfma_f64_4x4x4 (profiles/r03/f64_rates.txt): A_b[i][k] <- lane 4 b + i + 16 k, B_b[k][j] <- lane 4 b + j + 16 k, D_b[i][j] -> lane
    # 4 b + j + 16 i.  Lane (row, col) supplies H_rz[col mod 2][row mod 2] where (col mod 4) div 2 == row div 2, else 0.
    rng = np.random.default_rng(2)
    H = rng.standard_normal((2, 2, 2))                                   # [rz][r][a]
    Tm = rng.standard_normal((4, 16))                                    # [row = (rz, a)][col]
    A_lane = np.zeros((4, 16))
    for row in range(4):
        for col in range(16):
            rzc, rr, rzr, aa = (col & 3) // 2, (col & 3) % 2, row // 2, row % 2
            A_lane[row, col] = H[rzr, rr, aa] if rzc == rzr else 0.0
    D = np.zeros((4, 16))
    for b in range(4):
        for i in range(4):
            for jj in range(4):
                D[i, 4 * b + jj] = sum(A_lane[k, 4 * b + i] * Tm[k, 4 * b + jj] for k in range(4))
    for rz in range(2):
        for r in range(2):
            want = sum(H[rz, r, a] * Tm[rz * 2 + a] for a in range(2))
            assert np.allclose(D[rz * 2 + r], want)


def test_noise_ledger_every_block_once():
    cp, n_sym = 16, 2
    for NW in (1, 2, 4, 8):
        N = 256 * NW
        row = n_sym * (N + cp)
        for os in range(n_sym):
            owner = {}
            for j in range(NW):
                pj = j & 1
                for ln in range(64):
                    r, h = ln >> 4, ln & 15
                    if NW == 1:                                          # full-wave: lanes l, l ^ 4; the lane evaluates registers 8 par + cc
                        par = (h >> 2) & 1
                        i0 = r * row + os * (N + cp) + cp + 4 * (h & 3) + ((h >> 2) & 2)
                        for cc in range(8):
                            c = 8 * par + cc
                            bi = (i0 >> 1) + (cc & 3) * 32 + (cc >> 2) * 8 + par * 16
                            n_even = pw_mtime(h & ~4, c)                 # the pair's even sample time
                            assert 2 * bi == r * row + os * (N + cp) + cp + n_even
                            for lane_of, parity in ((ln, par), (ln ^ 4, 1 - par)):
                                hh = lane_of & 15
                                assert pw_mtime(hh, c) == n_even + parity    # register c of that lane holds the pair's sample `parity`
                            assert bi not in owner
                            owner[bi] = (j, ln, c)
                    else:
                        i00 = r * row + os * (N + cp) + cp + (j & ~1) + 32 * NW * pj + NW * pw_mtime(h, 0)
                        for cc in range(8):
                            c = 8 * pj + cc
                            bi = (i00 >> 1) + (NW // 2) * pw_mtime(0, cc)
                            n_mine = NW * pw_mtime(h, c) + j               # my sample time in register c; the partner's is n_mine ^ 1
                            assert 2 * bi == r * row + os * (N + cp) + cp + (n_mine & ~1)
                            assert bi not in owner
                            owner[bi] = (j, ln, c)
            first = (os * (N + cp) + cp) // 2
            want = {(r * row) // 2 + first + p for r in range(4) for p in range(N // 2)}
            assert set(owner) == want, (NW, os)                          # every NOISE block of the symbol, once


def test_packed_walk_index_space_and_label_runs():
    """walk_f64.hpp: lane pair p = p0 + lane = rloc * NP + rem advanced by 64 per pass (NP >= 64: one wrap at most), the label runs of
    a pass ((segment, stream) -> <= 9 DATA blocks from base >> 4), a lane's two labels as one 16-bit word, the noise block of the pair."""
    for n_symbols, per_wave, S in ((128, 16, 3), (130, 16, 3), (200, 16, 3), (254, 8, 6), (500, 8, 6), (1000, 7, 4)):
        NP = n_symbols // 2
        n_pairs = per_wave * NP
        rloc, rem = np.zeros(64, int), np.arange(64)
        covered = np.zeros((per_wave, NP), int)
        for p0 in range(0, n_pairs, 64):
            valid = p0 + np.arange(64) < n_pairs
            assert np.all((rloc * NP + rem)[valid] == (p0 + np.arange(64))[valid])
            rlo, t_lo = rloc[0], 2 * rem[0]
            seg = rloc - rlo
            assert seg[valid].max() <= 1                                  # a pass covers at most two realizations
            for ln in np.flatnonzero(valid):
                covered[rloc[ln], rem[ln]] += 1
                t = 2 * rem[ln]
                for k in range(S):
                    q = k * n_symbols + t
                    base = k * n_symbols + (0 if seg[ln] else t_lo)
                    off = q - (base & ~15)
                    assert 0 <= off and off + 1 < 144 and off % 2 == 0      # both labels inside the run's nine blocks, 16-bit aligned
                    blk = (base >> 4) + off // 16                         # the producer lane (run, off // 16) evaluated DATA block blk
                    assert blk == q >> 4 and (q + 1) >> 4 == blk          # ... = the block of symbols q and q + 1
            rem = rem + 64
            wrap = rem >= NP
            rem = np.where(wrap, rem - NP, rem)
            rloc = rloc + wrap
        assert np.all(covered == 1)
    # noise: the two columns of a lane = the two samples of ONE block of row (k R + a): block ((k R + a) NS + t) >> 1
    for ns in (128, 200):
        for rowi in range(6):
            for t in range(0, ns, 2):
                i0, i1 = rowi * ns + t, rowi * ns + t + 1
                assert i0 >> 1 == i1 >> 1 == (rowi * ns + t) >> 1


def test_wave_total_by_dpp_adds():
    """walk_wave_total: row_shr 1, 2, 4, 8 inside rows of sixteen (sources outside the row add 0), row_bcast:15 into rows 1 and 3,
    row_bcast:31 into rows 2 and 3 -- lane 63 holds the sum."""
    rng = np.random.default_rng(3)
    v = rng.integers(0, 1 << 20, 64).astype(np.int64)
    total = int(v.sum())
    for sh in (1, 2, 4, 8):
        w = v.copy()
        for lane in range(64):
            if (lane & 15) >= sh:
                w[lane] = v[lane] + v[lane - sh]
        v = w
    w = v.copy()
    for lane in range(64):
        if (lane >> 4) in (1, 3):
            w[lane] = v[lane] + v[((lane >> 4) << 4) - 1]
    v = w
    w = v.copy()
    for lane in range(32, 64):
        w[lane] = v[lane] + v[31]
    assert int(w[63]) == total


def test_level_domain_count_equals_the_label_domain_count():
    """qam_count4 on (decided levels ^ sent levels) = symbol and bit errors of (decided label ^ sent label), for every pair of labels of
    16- / 64- / 256-QAM (label = binary row << hb | binary column, level = b ^ (b >> 1) per field)."""
    for hb in (1, 2, 3, 4):
        fm = (1 << hb) - 1
        m1 = ((fm >> 1) | ((fm >> 1) << hb)) & 0xFF
        m2 = ((fm >> 2) | ((fm >> 2) << hb)) & 0xFF
        M = 1 << (2 * hb)
        to_level = lambda w: w ^ ((w >> 1) & m1)
        for a in range(M):
            for b in range(0, M, max(1, M // 64)):
                x = to_level(a) ^ to_level(b)
                y = x ^ ((x >> 1) & m1)
                y ^= (y >> 2) & m2
                assert (x != 0) == (a != b)
                assert bin(y).count("1") == bin(a ^ b).count("1"), (hb, a, b)
