"""GPU: the quarter-wave complex128 config-4 kernels at fft_size 1024, 4 x 4, full band, even cyclic prefix, decisions by slicer or
certificate: csrc/pipeline_mimo_pw.hip with NW = 4 (channel AND decode on the matrix cores: the default of mcle_run_mimo_ofdm since
the end of round 6, option f64_threads = 0 / 263, 264 = two wavefronts per SIMD) and csrc/pipeline_mimo_qw.hip (the first edition, VALU
decode: f64_threads = 260 / 262); 261 selects the planar kernel they replaced -- per-realization symbol AND bit error counts equal to the oracle chain's
(oracle/chains.py::chain_mimo_ofdm, pinned to the reference by tests/golden/c4_mimo_ofdm.npz) on every corner of its envelope,
equal to the planar kernel's over a thousand realizations per case, and requests outside the envelope served by the planar kernel.
Reference: apps/mimo/simulate_mimo.py:68-142, mimo/mimo.py:609-660, modulators/ofdm.py:394-466."""
import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 60221407

# inside the envelope: full band, even prefix, square QAM (margin certificate / slicer) or QPSK (quadrant certificate)
INSIDE = [dict(mod="qam", M=64, snr_db=25.0),                                              # BASELINE config 4
          dict(mod="qam", M=16, snr_db=18.0, cp_size=8, mmse=False, n_ofdm_sym=3),         # ZF, three OFDM symbols
          dict(mod="qam", M=256, snr_db=32.0, cp_size=0),
          dict(mod="qpsk", M=4, snr_db=8.0, cp_size=32, n_ofdm_sym=2),                     # quadrant certificate
          dict(mod="qam", M=4, snr_db=6.0, cp_size=144)]
# outside: partial band, odd prefix, a constellation without a certificate -> the planar kernel answers
OUTSIDE = [dict(mod="qam", M=64, snr_db=25.0, num_used=600, n_ofdm_sym=2),
           dict(mod="qam", M=16, snr_db=18.0, cp_size=7),
           dict(mod="psk", M=8, snr_db=14.0, n_ofdm_sym=2, cp_size=32)]


def _set(engine, kw):
    engine.set_constellation(chains.constellation(kw["mod"], kw["M"]), _lib.CONST_QAM if kw["mod"] == "qam" else _lib.CONST_GENERIC)


def _run(engine, kw, first, count, method, threads):
    nv = 1.0 / omodem.dB2Linear(kw["snr_db"])
    with engine.options(f64_threads=threads):
        return engine.run_mimo_ofdm(4, 4, 1024, kw.get("cp_size", 16), kw.get("num_used") or 1024, kw.get("n_ofdm_sym", 1), nv, SEED,
                                    first, count, mmse=kw.get("mmse", True), method=method, dtype="f64", per_realization=True)


def _oracle(kw, first, count):
    okw = dict(mod=kw["mod"], M=kw["M"], nt=4, nr=4, fft_size=1024, cp_size=kw.get("cp_size", 16), num_used=kw.get("num_used"),
               n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], mmse=kw.get("mmse", True))
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    return (np.array([w["symbol_errors"] for w in want]), np.array([w["bit_errors"] for w in want]), want[0]["num_symbols"],
            want[0]["num_bits"])


@pytest.mark.parametrize("case", range(len(INSIDE) + len(OUTSIDE)))
def test_counts_equal_the_oracle(engine, case):
    kw = (INSIDE + OUTSIDE)[case]
    _set(engine, kw)
    first, count = (1 << 36) + 1009, 6
    want_se, want_be, nsym, nbits = _oracle(kw, first, count)
    methods = [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if kw["mod"] == "qam" else [])
    for method in methods:
        for threads in (0, 263, 264, 260, 262, 261):
            res, se, be = _run(engine, kw, first, count, method, threads)
            assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (case, method, threads, se, want_se)
            assert res["n_realizations"] == count and res["n_skipped"] == 0 and res["n_symbols"] == nsym and res["n_bits"] == nbits
            assert res["sym_errors"] == int(want_se.sum()) and res["sym_errors_sq"] == int((want_se.astype(np.int64) ** 2).sum())


@pytest.mark.parametrize("case", range(len(INSIDE)))
def test_equal_to_the_planar_kernel_over_a_thousand_realizations(engine, case):
    """Both are complex128 statements of the same link with differently rounded transforms: a rounding-level tie may differ once
    in ~1e7 symbols.  Same sums of squares under any split of the range (launch slices are invisible)."""
    kw = INSIDE[case]
    _set(engine, kw)
    n = 1100
    for method in [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if kw["mod"] == "qam" else []):
        qw, se, be = _run(engine, kw, 77, n, method, 0)
        pl, se_p, be_p = _run(engine, kw, 77, n, method, 261)
        assert np.count_nonzero(se != se_p) <= 1 and np.max(np.abs(se.astype(np.int64) - se_p.astype(np.int64))) <= 1
        assert np.count_nonzero(be != be_p) <= 1
        assert qw["n_realizations"] == pl["n_realizations"] == n and qw["n_skipped"] == pl["n_skipped"]
        a = _run(engine, kw, 77, 300, method, 0)[0]
        b = _run(engine, kw, 377, n - 300, method, 0)[0]
        for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations", "n_skipped"):
            assert qw[k] == a[k] + b[k], k
        again = _run(engine, kw, 77, n, method, 0)
        assert np.array_equal(again[1], se) and np.array_equal(again[2], be)             # bit-identical from run to run


def test_headline_geometry_against_the_oracle_over_2048_realizations(engine):
    """BASELINE config 4 (4 x 4 MMSE, 64-QAM, OFDM(1024, 16), 25 dB) on the quarter-wave kernel, both demodulators and both register
    bounds: every per-realization count of 2 048 consecutive realizations (8.4e6 symbols) equal to the oracle's."""
    kw = INSIDE[0]
    _set(engine, kw)
    first, count = 987654321, 2048
    want_se, want_be, _, _ = _oracle(kw, first, count)
    assert want_se.sum() > 1e5
    for threads in (0, 264, 260, 262):
        for method in (_lib.DEMOD_MINDIST, _lib.DEMOD_QAM_SLICER):
            res, se, be = _run(engine, kw, first, count, method, threads)
            assert np.array_equal(se, want_se), (threads, method, np.flatnonzero(se != want_se)[:5])
            assert np.array_equal(be, want_be)
            assert res["sym_errors"] == int(want_se.sum()) and res["bit_errors"] == int(want_be.sum())


def test_singular_channels_are_skipped_like_the_planar_kernel(engine):
    """Zero forcing at infinite SNR over many realizations: whatever the record kernel flags as skipped is skipped by both."""
    kw = dict(mod="qam", M=16, snr_db=300.0, mmse=False)
    _set(engine, kw)
    a = _run(engine, kw, 0, 4096, _lib.DEMOD_MINDIST, 260)[0]
    b = _run(engine, kw, 0, 4096, _lib.DEMOD_MINDIST, 261)[0]
    assert a["n_skipped"] == b["n_skipped"] and a["n_realizations"] == b["n_realizations"] and a["sym_errors"] == b["sym_errors"]
