"""GPU: the packed symbol walk of round 6 (complex128, and complex64 at the end of the file) (csrc/walk_f64.hpp: the lane pairs of a chunk of realizations as one index
space, decision form fixed at compile time, records in LDS) -- the default of mcle_run_ia / mcle_run_bd in complex128 for an even
number of columns >= 128 -- against the per-realization walks of rounds 2-5 (option walk_legacy = 1) and against the oracle
(oracle/chains.py under the same Philox keying).  Every comparison is per realization and exact: the two kernels make the same
draws position by position and evaluate the estimates in the same association."""
import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 27182818


def _same(a, b):
    (ra, sa, ba), (rb, sb, bb) = a, b
    assert np.array_equal(sa, sb), np.flatnonzero(sa != sb)[:8]
    assert np.array_equal(ba, bb)
    for k in ("sym_errors", "bit_errors", "sym_errors_sq", "bit_errors_sq", "n_realizations", "n_skipped", "n_symbols", "n_bits"):
        assert ra[k] == rb[k], k


# (constellation, M, kind, method): every decision form of the walk -- QAM margin certificate, slicer, quadrant certificate,
# the generic search (8-PSK: sector certificate inside the candidate grid; BPSK-like 2 points), and a 256-point table
FORMS = [("qam", 16, _lib.CONST_QAM, _lib.DEMOD_MINDIST), ("qam", 16, _lib.CONST_QAM, _lib.DEMOD_QAM_SLICER),
         ("psk", 4, _lib.CONST_GENERIC, _lib.DEMOD_MINDIST), ("psk", 8, _lib.CONST_GENERIC, _lib.DEMOD_MINDIST),
         ("qam", 256, _lib.CONST_QAM, _lib.DEMOD_MINDIST)]


@pytest.mark.parametrize("form", FORMS, ids=lambda f: "%s%d-%d" % (f[0], f[1], f[3]))
@pytest.mark.parametrize("n_symbols", [128, 130, 200, 254, 256, 1000])
def test_ia_packed_walk_equals_the_per_realization_walk(engine, form, n_symbols):
    """Config 5's shape at column counts on both sides of the pass boundaries (64 pairs: one realization per pass; 65: a pass
    straddles two; 100: the benchmark; 127 / 128; 500), counts not a multiple of the chunk (16) so that the last chunk is short,
    15 dB so that there ARE errors."""
    mod, M, kind, method = form
    engine.set_constellation(chains.constellation(mod, M), kind)
    nv = 1.0 / omodem.dB2Linear(15.0 if M <= 16 else 30.0)
    for first, count in ((0, 1), (5, 15), (1000, 16 * 37 + 9)):
        new = engine.run_ia(n_symbols, nv, SEED, first, count, method=method, dtype="f64", per_realization=True)[:3]
        with engine.options(walk_legacy=1):
            old = engine.run_ia(n_symbols, nv, SEED, first, count, method=method, dtype="f64", per_realization=True)[:3]
        _same(new, old)
        assert new[0]["sym_errors"] > 0 or count < 15


@pytest.mark.parametrize("form", FORMS[:4], ids=lambda f: "%s%d-%d" % (f[0], f[1], f[3]))
@pytest.mark.parametrize("shape", [(2, 1, 200), (3, 1, 500), (2, 2, 130), (3, 2, 500), (3, 2, 128), (3, 2, 254)])
def test_bd_packed_walk_equals_the_per_realization_walk(engine, form, shape):
    """Block diagonalisation with two / three users of one / two antennas (the compile-time shapes of the packed walk), with and
    without a path-loss matrix, chunk of 8."""
    mod, M, kind, method = form
    K, nr, n_symbols = shape
    engine.set_constellation(chains.constellation(mod, M), kind)
    nv = 1.0 / (10.0 ** 1.0)
    pl = np.abs(np.random.default_rng(3).normal(1.0, 0.3, (K, K))) + 0.2
    for first, count, pathloss in ((0, 1, None), (77, 8 * 41 + 5, None), (500, 300, pl)):
        kw = dict(method=method, dtype="f64", per_realization=True, pathloss=pathloss)
        new = engine.run_bd(K, nr, n_symbols, 1.0, nv, SEED, first, count, **kw)
        with engine.options(walk_legacy=1):
            old = engine.run_bd(K, nr, n_symbols, 1.0, nv, SEED, first, count, **kw)
        _same(new, old)


def test_requests_outside_the_packed_walk_keep_their_kernels(engine):
    """Odd column counts, fewer than 128 columns, four users and three antennas per user run the walks of rounds 2-5 whatever the
    option says: same counters with walk_legacy 0 and 1, and equal to the oracle on a few realizations."""
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    nv = 1.0 / omodem.dB2Linear(15.0)
    for ns in (1, 63, 126, 201):
        a = engine.run_ia(ns, nv, SEED, 3, 50, dtype="f64", per_realization=True)[:3]
        with engine.options(walk_legacy=1):
            b = engine.run_ia(ns, nv, SEED, 3, 50, dtype="f64", per_realization=True)[:3]
        _same(a, b)
        want = [chains.chain_ia(chains.PhiloxRng(SEED, r), mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=ns, snr_db=15.0) for r in range(3, 9)]
        assert [int(v) for v in a[1][:6]] == [o["symbol_errors"] for o in want]
    engine.set_constellation(chains.constellation("psk", 4), _lib.CONST_GENERIC)
    for K, nr, ns in ((4, 2, 200), (2, 3, 200), (3, 2, 201), (3, 2, 100)):
        a = engine.run_bd(K, nr, ns, 1.0, 0.1, SEED, 3, 50, dtype="f64", per_realization=True)
        with engine.options(walk_legacy=1):
            b = engine.run_bd(K, nr, ns, 1.0, 0.1, SEED, 3, 50, dtype="f64", per_realization=True)
        _same(a, b)


@pytest.mark.parametrize("n_symbols", [130, 200])
def test_ia_packed_walk_against_the_oracle(engine, n_symbols):
    """600 realizations of the oracle chain per column count (the 5 000 of config 5 itself: tests/test_gpu_oracle_depth.py), both
    demodulators; 130 columns put a realization boundary inside most passes."""
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    first, count = 31000, 600
    out = [chains.chain_ia(chains.PhiloxRng(SEED, r), mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=n_symbols, snr_db=17.0)
           for r in range(first, first + count)]
    want_se = np.array([o["symbol_errors"] for o in out], dtype=np.int64)
    want_be = np.array([o["bit_errors"] for o in out], dtype=np.int64)
    for method in (_lib.DEMOD_MINDIST, _lib.DEMOD_QAM_SLICER):
        res, se, be = engine.run_ia(n_symbols, 1.0 / omodem.dB2Linear(17.0), SEED, first, count, method=method, dtype="f64",
                                    per_realization=True)[:3]
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be)
        assert res["n_symbols"] == 3 * n_symbols and res["n_realizations"] == count


def test_bd_packed_walk_against_the_oracle(engine):
    """400 realizations of the oracle's block-diagonalisation chain at a column count that is not a multiple of 128 (K = 3, 2 x 2,
    4-PSK, 254 columns, 10 dB)."""
    engine.set_constellation(chains.constellation("psk", 4), _lib.CONST_GENERIC)
    nv = 0.1
    first, count = 900, 400
    out = [chains.chain_bd(chains.PhiloxRng(SEED, r), mod="psk", M=4, K=3, nr=2, NSymbs=254, iPu=1.0, noise_var=nv, canonical=True)
           for r in range(first, first + count)]
    res, se, be = engine.run_bd(3, 2, 254, 1.0, nv, SEED, first, count, method=_lib.DEMOD_MINDIST, dtype="f64", per_realization=True)
    assert np.array_equal(se, np.array([o["symbol_errors"] for o in out])) and np.array_equal(be, np.array([o["bit_errors"] for o in out]))
    assert res["sym_errors"] > 0


# ---- complex64: the same packing (k_link_walk<float>; the estimates of a pass decided together, the slicer in the packed level domain) ----
def _close32(a, b, n_sym_per_realization, ties=3):
    """Two complex64 statements of the same link: decisions differ at rounding-level ties only."""
    (ra, sa, ba), (rb, sb, bb) = a, b
    n = len(sa)
    assert ra["n_realizations"] == rb["n_realizations"] == n and ra["n_skipped"] == rb["n_skipped"] and ra["n_symbols"] == rb["n_symbols"]
    assert np.max(np.abs(sa.astype(np.int64) - sb.astype(np.int64))) <= ties
    assert abs(int(sa.astype(np.int64).sum()) - int(sb.astype(np.int64).sum())) <= 2e-5 * n * n_sym_per_realization + 3
    assert abs(int(ba.astype(np.int64).sum()) - int(bb.astype(np.int64).sum())) <= 4e-5 * n * n_sym_per_realization + 4


@pytest.mark.parametrize("form", FORMS, ids=lambda f: "%s%d-%d" % (f[0], f[1], f[3]))
@pytest.mark.parametrize("n_symbols", [128, 130, 200, 1000])
def test_ia_packed_walk_in_complex64(engine, form, n_symbols):
    """Against the round-5 complex64 walk (option walk_legacy = 1) and against the complex128 packed walk on the same indices."""
    mod, M, kind, method = form
    engine.set_constellation(chains.constellation(mod, M), kind)
    nv = 1.0 / omodem.dB2Linear(15.0 if M <= 16 else 30.0)
    for first, count in ((0, 1), (1000, 16 * 37 + 9)):
        new = engine.run_ia(n_symbols, nv, SEED, first, count, method=method, dtype="f32", per_realization=True)[:3]
        with engine.options(walk_legacy=1):
            old = engine.run_ia(n_symbols, nv, SEED, first, count, method=method, dtype="f32", per_realization=True)[:3]
        f64 = engine.run_ia(n_symbols, nv, SEED, first, count, method=method, dtype="f64", per_realization=True)[:3]
        _close32(new, old, 3 * n_symbols)
        _close32(new, f64, 3 * n_symbols)


@pytest.mark.parametrize("form", FORMS[:4], ids=lambda f: "%s%d-%d" % (f[0], f[1], f[3]))
@pytest.mark.parametrize("shape", [(2, 1, 200), (3, 1, 500), (2, 2, 130), (3, 2, 500), (3, 2, 254)])
def test_bd_packed_walk_in_complex64(engine, form, shape):
    mod, M, kind, method = form
    K, nr, n_symbols = shape
    engine.set_constellation(chains.constellation(mod, M), kind)
    nv = 1.0 / (10.0 ** 1.0)
    pl = np.abs(np.random.default_rng(3).normal(1.0, 0.3, (K, K))) + 0.2
    for first, count, pathloss in ((0, 1, None), (77, 8 * 41 + 5, None), (500, 300, pl)):
        kw = dict(method=method, per_realization=True, pathloss=pathloss)
        new = engine.run_bd(K, nr, n_symbols, 1.0, nv, SEED, first, count, dtype="f32", **kw)
        with engine.options(walk_legacy=1):
            old = engine.run_bd(K, nr, n_symbols, 1.0, nv, SEED, first, count, dtype="f32", **kw)
        f64 = engine.run_bd(K, nr, n_symbols, 1.0, nv, SEED, first, count, dtype="f64", **kw)
        _close32(new, old, K * nr * n_symbols)
        _close32(new, f64, K * nr * n_symbols, ties=4)
