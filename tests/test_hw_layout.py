"""CPU: the index arithmetic of csrc/siso_tdl_hw.hpp (config 3 at fft_size 2048 with two wavefronts per realization, round 6) replayed in
NumPy -- no GPU, no library call; the oracle package supplies only the reference's subcarrier map (oracle/ofdm.py restates
modulators/ofdm.py:188-224):
  * the decomposition: x[2 m + j] as a 1024-point inverse transform of (X[k'] + (-1)^j X[k' + 1024]) w2048^(-j k'), and
    Y[k' + 1024 q] = A0[k'] + (-1)^q w2048^k' A1[k'] from the two partial forward transforms -- against numpy.fft on 2048 points;
  * the bins a wavefront forms its inputs from: data_of_bin (the kernel's inverse of the subcarrier map) for full and partial bands,
    the labels a lane reads for its sixteen positions;
  * the parity planes: the delayed sample x[n - d] of n = 2 m + j read at plane (j - d) & 1, index Pp + m + floor((j - d) / 2), for every
    delay up to the envelope's 256 and every sample, prefix included;
  * the noise ledger: with an even row start every NOISE block is evaluated once, by wavefront j = (register >= 8), and each
    wavefront's register c receives the half of block nbase / 2 + m that is ITS sample; with an odd start the unpaired rule;
  * the equaliser's twiddle w2048^(f d) = w^(F(lane) d) x w^(F(64 k) d) x (-1)^(q d) and the data position of bin f."""
import numpy as np

from oracle import ofdm as oofdm

N, H = 2048, 1024


def fft_index_of_pos(p, n=H):
    """position after the radix-4 DIF stages -> bin index: the base-4 digit reversal (csrc/fft.hpp: fft_index_of_pos)"""
    digits = int(round(np.log(n) / np.log(4)))
    f = 0
    for _ in range(digits):
        f = f * 4 + (p & 3)
        p >>= 2
    return f


def data_of_bin(f, U):
    """siso_tdl_hw.hpp: data position of bin f, -1 where the bin is unused"""
    hU = U // 2
    if U == N:
        return (f + N // 2) & (N - 1)
    neg, pos = f >= N - hU, 1 <= f <= hU
    return f - (N - hU) if neg else (hU + f - 1 if pos else -1)


def test_transmit_and_receive_decomposition_against_numpy_fft():
    rs = np.random.RandomState(1)
    X = rs.randn(N) + 1j * rs.randn(N)
    kp = np.arange(H)
    w = np.exp(-2j * np.pi * kp / N)                                          # the forward table's w2048^k'
    x_ref = np.fft.ifft(X) * N                                                # unnormalized inverse transform
    for j in (0, 1):
        a = (X[:H] - X[H:]) * np.conj(w) if j else X[:H] + X[H:]
        xj = np.fft.ifft(a) * H
        assert np.allclose(xj, x_ref[j::2], rtol=0, atol=1e-9)                # wavefront j holds the samples n = 2 m + j
    y = rs.randn(N) + 1j * rs.randn(N)
    Y_ref = np.fft.fft(y)
    A0, A1 = np.fft.fft(y[0::2]), np.fft.fft(y[1::2])
    for q in (0, 1):
        Yq = A0 - w * A1 if q else A0 + w * A1
        assert np.allclose(Yq, Y_ref[q * H:(q + 1) * H], rtol=0, atol=1e-9)   # wavefront q decides the bins k' + 1024 q


def test_positions_cover_every_bin_once_and_labels_follow_the_subcarrier_map():
    # lane gi, register c <-> position gi + 64 c <-> k' = F(gi) | F(64 c): a bit permutation, so the OR is a sum
    seen = np.zeros(H, dtype=int)
    for gi in range(64):
        fl = fft_index_of_pos(gi)
        for c in range(16):
            kq = fl | fft_index_of_pos(64 * c)
            assert kq == fft_index_of_pos(gi + 64 * c) == fl + fft_index_of_pos(64 * c)
            seen[kq] += 1
    assert (seen == 1).all()
    for U in (N, 1200, 2, 2046):
        used = oofdm.used_subcarrier_indexes(N, U)                             # data d -> bin used[d]
        inv = np.full(N, -1)
        inv[used] = np.arange(U)
        assert [data_of_bin(f, U) for f in range(N)] == inv.tolist()
        # the transmit inputs: X[k'] and X[k' + H] from the label bytes, zero where the bin is unused
        rs = np.random.RandomState(U)
        labels = rs.randint(0, 64, size=U)
        table = rs.randn(64) + 1j * rs.randn(64)
        bins = np.zeros(N, dtype=complex)
        bins[used] = table[labels]
        X = np.array([table[labels[data_of_bin(f, U)]] if data_of_bin(f, U) >= 0 else 0.0 for f in range(N)])
        assert np.array_equal(X, bins)


def test_parity_planes_hold_every_delayed_sample_where_the_kernel_reads_it():
    rs = np.random.RandomState(2)
    x = rs.randn(N) + 1j * rs.randn(N)                                        # one symbol's time samples (prefix = its own end)
    for dmax in (0, 1, 4, 17, 200, 255, 256):
        Pp = ((dmax + 1) // 2 + 15) & ~15                                      # run_siso_tdl_hw: x_elems - H
        planes = []
        for q in (0, 1):                                                      # plane q: x[2 m + q] at Pp + m, the last Pp samples in front
            pl = np.zeros(H + Pp, dtype=complex)
            pl[Pp:] = x[q::2]
            pl[:Pp] = x[q::2][H - Pp:]
            planes.append(pl)
        for d in sorted({0, min(1, dmax), dmax // 2, max(dmax - 1, 0), dmax}):
            for j in (0, 1):
                e = j - d
                q, o = e & 1, e >> 1                                           # Python's & and >> on negative ints = two's complement / floor
                assert o >= -Pp
                m = np.arange(H)
                got = planes[q][Pp + m + o]
                want = x[(2 * m + j - d) % N]                                  # cp >= d: the sample d behind n, cyclically
                assert np.array_equal(got, want), (dmax, d, j)


def test_noise_ledger_even_and_odd_row_starts():
    for nbase in (0, 16, 2064, 4128 + 16, 33, 2081):
        need = {}                                                             # stream sample index -> (wavefront, lane, register)
        for j in (0, 1):
            for gi in range(64):
                for c in range(16):
                    need[nbase + 2 * (gi + 64 * c) + j] = (j, gi, c)
        assert sorted(need) == list(range(nbase, nbase + N))
        if nbase % 2 == 0:
            evaluated = {}
            for j in (0, 1):                                                  # wavefront j draws the blocks of registers 8 j .. 8 j + 7
                for gi in range(64):
                    for cc in range(8):
                        m = gi + 64 * (8 * j + cc)
                        blk = (nbase >> 1) + m
                        assert blk not in evaluated
                        evaluated[blk] = (j, gi, cc)
                        # its own half = sample j of the block, the partner's = sample 1 - j, same lane, register 8 j + cc of either
                        assert need[2 * blk + j] == (j, gi, 8 * j + cc) and need[2 * blk + 1 - j] == (1 - j, gi, 8 * j + cc)
            assert sorted(evaluated) == list(range(nbase // 2, nbase // 2 + H))    # every block once
            # after the exchange: register cc of BOTH wavefronts takes the blocks of wavefront 0, register 8 + cc those of wavefront 1
            for j in (0, 1):
                for gi in range(64):
                    for cc in range(8):
                        lo_src, hi_src = evaluated[(nbase >> 1) + gi + 64 * cc], evaluated[(nbase >> 1) + gi + 64 * (8 + cc)]
                        assert lo_src == (0, gi, cc) and hi_src == (1, gi, cc)
        else:                                                                 # unpaired: block i0 >> 1, half i0 & 1
            for (i0, (j, gi, c)) in need.items():
                assert i0 == nbase + 2 * (gi + 64 * c) + j and (i0 >> 1) * 2 + (i0 & 1) == i0


def test_equaliser_twiddle_factors_and_sign():
    w = np.exp(-2j * np.pi * np.arange(N) / N)
    for d in (0, 1, 2, 3, 17, 64, 255, 256):
        for q in (0, 1):
            for gi in (0, 1, 5, 37, 63):
                for k in (0, 3, 9, 15):
                    f = (fft_index_of_pos(gi) | fft_index_of_pos(64 * k)) + H * q
                    lane_part = w[(fft_index_of_pos(gi) * d) & (N - 1)]
                    uni_part = w[(fft_index_of_pos(64 * k) * d) & (N - 1)]
                    sign = -1.0 if (q & d & 1) else 1.0
                    assert abs(lane_part * uni_part * sign - w[(f * d) & (N - 1)]) < 1e-12
