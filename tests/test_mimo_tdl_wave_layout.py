"""CPU: the index maps of the one-receive-antenna-per-wavefront kernel of the frequency-selective MIMO-OFDM link
(csrc/mimo_tdl_wave.hpp, round 5) replayed in NumPy -- nothing here needs a GPU:
  * the decode's work items: every position of every size exactly once, the BQ subcarriers of an item are f0 + j N / BQ, and the
    twiddle identity w^((f0 + j N / BQ) d) = w^(f0 d) (-i)^(j d) / (-1)^(j d) that lets the S Nr Nt products mean x twiddle serve BQ bins;
  * H(f) by delay classes and a radix-BQ butterfly equals the literal sum over the taps;
  * the Gram accumulation row by row (mimo.hpp: blast_gram_row / blast_solve_gram) equals sqrt(Nt) solve(H^H H + nv I, H^H y) for
    every 1 <= Nt <= Nr <= 4 (mimo/mimo.py:287-309, :597-607);
  * the record layout of k_mimo_tdl_symbol_polys<T, true>: every coefficient of a receive antenna on its own (register, lane);
  * LDS bank conflicts of the decode's reads under the 32-lane read rule (tests/test_f64_layout.py);
  * the delay-class POSITIONS of the decode (MimoTdlParams::cls_code) and its two loops + butterfly against the literal H(f)."""
import numpy as np
import pytest

from test_f64_layout import read_conflicts, swz16f, swz64


def _shape(n):
    log2 = n.bit_length() - 1
    return log2 // 2, bool(log2 & 1)


def pos_of_index(n, f):                               # fft.hpp fft_pos_of_index<N>
    n4, has2 = _shape(n)
    pos, size = 0, n
    for _ in range(n4):
        size >>= 2
        pos += (f & 3) * size
        f >>= 2
    if has2:
        pos += f & 1
    return pos


def index_of_pos(n, p):                               # fft.hpp fft_index_of_pos<N>
    n4, has2 = _shape(n)
    f, size, mul = 0, n, 1
    for _ in range(n4):
        size >>= 2
        q = p // size
        p -= q * size
        f += q * mul
        mul <<= 2
    if has2:
        f += p * mul
    return f


def posj(n, bq, j):                                   # mimo_wave_posj<N, BQ>(j)
    return pos_of_index(n, j * (n // bq))


def deposit(n, jm, q):                                # mimo_wave_deposit<N, JM>(q)
    p, b = 0, 0
    for bit in range(n.bit_length() - 1):
        if (jm >> bit) & 1:
            continue
        p |= ((q >> b) & 1) << bit
        b += 1
    return p


def p0_of(n, bq, ln, wi):                             # mimo_wave_p0<N, BQ>(ln, wi)
    if n == 1024:
        eb = 4 // bq
        gq, e, h = wi // eb, wi % eb, (ln >> 5) & 1
        g = (ln & 15) | (gq << 4) | (h << 6) | (((ln >> 4) & 1) << 7)
        return 4 * g + e
    jm = 0
    for j in range(bq):
        jm |= posj(n, bq, j)
    return deposit(n, jm, ln + 64 * wi)


@pytest.mark.parametrize("n", [256, 512, 1024, 2048])
@pytest.mark.parametrize("bq", [1, 2, 4])
def test_decode_work_items_cover_every_position_once_and_hold_strided_bins(n, bq):
    wi_count = n // (64 * bq)
    if wi_count < 1:
        pytest.skip("fewer than one work item")
    seen = np.zeros(n, dtype=int)
    swz = swz16f if n == 1024 else swz64
    worst = 0
    for wi in range(wi_count):
        slots = [[] for _ in range(bq)]
        for ln in range(64):
            p0 = p0_of(n, bq, ln, wi)
            f0 = index_of_pos(n, p0)
            assert f0 < n // bq
            for j in range(bq):
                p = p0 ^ posj(n, bq, j)
                assert p == p0 + posj(n, bq, j)                   # the j bits are clear in p0
                seen[p] += 1
                assert index_of_pos(n, p) == f0 + j * (n // bq)
                assert int(swz(p)) == int(swz(p0)) ^ int(swz(posj(n, bq, j)))      # the swizzles are XOR-linear
                slots[j].append(int(swz(p)))
        for j in range(bq):
            worst = max(worst, read_conflicts(slots[j]))
    assert np.all(seen == 1)
    # conflict free at 256 / 1024; two-way (32 extra cycles per wave-wide read) at 512 / 2048 where the radix-2 stage pairs
    # neighbouring positions -- BQ reads per antenna and work item next to several hundred VALU instructions
    assert worst == 0 if n in (256, 1024) else worst <= 32


@pytest.mark.parametrize("bq", [2, 4])
def test_twiddle_classes_and_butterfly_give_the_frequency_response(bq):
    rs = np.random.RandomState(5)
    n, S = 1024, 7
    delays = np.array([0, 1, 2, 3, 4, 9, 14])
    mean = rs.randn(S, 3, 2) + 1j * rs.randn(S, 3, 2)                       # [tap][rx][tx]
    tw = np.exp(-2j * np.pi * np.arange(n) / n)
    for f0 in (0, 1, 77, n // bq - 1):
        u = np.zeros((bq, 3, 2), dtype=complex)
        for s in range(S):
            u[delays[s] & (bq - 1)] += mean[s] * tw[(f0 * delays[s]) % n]
        for j in range(bq):
            if bq == 2:
                h = u[0] + (-1.0) ** j * u[1]
            else:
                h = sum(u[c] * (-1j) ** (j * c) for c in range(4))          # the forward radix-4 butterfly
            f = f0 + j * (n // bq)
            want = sum(mean[s] * tw[(f * delays[s]) % n] for s in range(S))
            assert np.allclose(h, want, rtol=0, atol=1e-12)


@pytest.mark.parametrize("nt,nr", [(1, 1), (1, 2), (2, 2), (1, 3), (2, 3), (3, 3), (1, 4), (2, 4), (3, 4), (4, 4)])
@pytest.mark.parametrize("nv", [0.0, 0.01])
def test_gram_row_accumulation_is_the_blast_solve(nt, nr, nv):
    rs = np.random.RandomState(nt * 10 + nr)
    H = rs.randn(nr, nt) + 1j * rs.randn(nr, nt)
    y = rs.randn(nr) + 1j * rs.randn(nr)
    A = np.zeros((nt, nt), dtype=complex)
    b = np.zeros(nt, dtype=complex)
    for r in range(nr):                               # blast_gram_row: lower triangle A[i][k] += conj(h[i]) h[k], b[i] += conj(h[i]) y
        h = H[r]
        for k in range(nt):
            A[k, k] += abs(h[k]) ** 2
            for i in range(k + 1, nt):
                A[i, k] += np.conj(h[i]) * h[k]
        b += np.conj(h) * y[r]
    # blast_solve_gram: Cholesky of the lower triangle (+ nv on the diagonal), two substitutions, x sqrt(Nt)
    L = np.zeros((nt, nt), dtype=complex)
    for j in range(nt):
        for i in range(j, nt):
            a = A[i, j] + (nv if i == j else 0.0)
            a -= sum(L[i, k] * np.conj(L[j, k]) for k in range(j))
            L[i, j] = np.sqrt(a.real) if i == j else a / L[j, j].real
    z = np.zeros(nt, dtype=complex)
    for i in range(nt):
        z[i] = (b[i] - sum(L[i, k] * z[k] for k in range(i))) / L[i, i].real
    for i in range(nt - 1, -1, -1):
        z[i] = (z[i] - sum(np.conj(L[k, i]) * z[k] for k in range(i + 1, nt))) / L[i, i].real
    x = z * np.sqrt(nt)
    want = np.sqrt(nt) * np.linalg.solve(H.conj().T @ H + nv * np.eye(nt), H.conj().T @ y)   # mimo.py:287-309, :597-607
    assert np.allclose(x, want, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("S,nt,nr,K", [(5, 4, 4, 2), (5, 4, 4, 5), (8, 4, 4, 12), (3, 2, 3, 3), (1, 1, 1, 2), (8, 1, 4, 7)])
def test_wave_record_layout_gives_every_coefficient_its_own_lane(S, nt, nr, K):
    nq, lw = (K + 2) // 2, 2 * S * nt                 # mimo_tdl_nq, the lanes of a parked register
    assert lw <= 64
    seen = set()
    for s in range(S):
        for r in range(nr):
            for a in range(nt):
                for m in range(K + 1):
                    where = (r * nq + (m >> 1)) * lw + 2 * (s * nt + a) + (m & 1)
                    assert where not in seen and where < nr * nq * lw
                    seen.add(where)
    assert len(seen) == S * nr * nt * (K + 1)
    rec = nr * nq * lw + S * nr * nt                  # mimo_tdl_wave_rec: coefficients, then the [S][Nr][Nt] means
    assert rec >= len(seen) + S * nr * nt


def class_positions(delays):
    """MimoTdlParams::cls_code as pipeline_mimo_tdl.hip fills it: even-delay taps at positions 0 .., odd-delay ones at 7, 6, .."""
    code, ne, no = [-1] * 8, 0, 0
    for s, d in enumerate(delays):
        c = (s << 16) | d
        if d & 1:
            code[7 - no] = c
            no += 1
        else:
            code[ne] = c
            ne += 1
    return code, ne, no


@pytest.mark.parametrize("delays", [(0, 1, 2, 3, 4), (0, 2, 4, 6, 8), (1, 3, 5, 7, 9, 11), (1, 2, 4, 6, 8, 10, 12, 14),
                                    (0, 1, 7, 33, 64, 65, 130, 200), (5,), (0,), (0, 7, 17, 31)])
def test_class_positions_and_the_two_loops_give_both_bins(delays):
    """The decode's two straight loops over the class positions (positions 0 .. ne - 1, then 7, 6, .. 8 - no) followed by the
    butterfly equal the literal frequency response at f0 and f0 + N / 2 for every split of <= 8 taps into delay parities."""
    rs = np.random.RandomState(len(delays) * 13 + delays[-1])
    n, S = 1024, len(delays)
    code, ne, no = class_positions(delays)
    assert ne + no == S and all(c == -1 for c in code[ne:8 - no])            # the two runs never overlap: ne + no <= 8
    assert [c >> 16 for c in code[:ne]] == [s for s in range(S) if delays[s] % 2 == 0]
    assert [code[7 - k] >> 16 for k in range(no)] == [s for s in range(S) if delays[s] % 2 == 1]
    mean = rs.randn(S, 4, 4) + 1j * rs.randn(S, 4, 4)
    tw = np.exp(-2j * np.pi * np.arange(n) / n)
    for f0 in (0, 3, 250, n // 2 - 1):
        wt = [tw[(f0 * (c & 0xFFFF)) % n] if c >= 0 else 0.0 for c in code]     # Wt[p]: the tap at position p
        u0 = sum((mean[code[k] >> 16] * wt[k] for k in range(ne)), np.zeros((4, 4), dtype=complex))
        u1 = sum((mean[code[7 - k] >> 16] * wt[7 - k] for k in range(no)), np.zeros((4, 4), dtype=complex))
        for j, h in enumerate((u0 + u1, u0 - u1)):
            f = f0 + j * (n // 2)
            want = sum(mean[s] * tw[(f * delays[s]) % n] for s in range(S))
            assert np.allclose(h, want, rtol=0, atol=1e-12)
