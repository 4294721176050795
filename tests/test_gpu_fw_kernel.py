"""GPU: the full-wave and half-wave complex128 kernels of config 4's link at fft_size 256 / 512 (csrc/pipeline_mimo_fw.hip: one
realization per wavefront; csrc/pipeline_mimo_pw.hip with NW = 2 / 8: two / eight wavefronts per realization at 512 / 2048; channel and decode on
v_mfma_f64_4x4x4; the defaults of mcle_run_mimo_ofdm at these sizes for 4 x 4, full band, even cyclic prefix, decisions by slicer or
certificate since round 6; option f64_threads = 260 asks for them explicitly, 262 bounds the registers for two wavefronts per
SIMD, 261 selects the planar kernel they replaced), and the same decomposition at 1024 with the decode on the matrix cores
(pipeline_mimo_pw.hip with NW = 4, f64_threads = 263 / 264; the quarter-wave kernel pipeline_mimo_qw.hip stays the default there) -- per-realization symbol AND bit error counts equal to the oracle
chain's (oracle/chains.py::chain_mimo_ofdm) on every corner of its envelope, equal to the planar kernel's over two thousand
realizations per case, and requests outside the envelope served by the planar kernel.
Reference: apps/mimo/simulate_mimo.py:68-142, mimo/mimo.py:609-660, modulators/ofdm.py:52-94, :394-466."""
import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 66260701

# inside the envelope: full band, even prefix, square QAM (margin certificate / slicer) or QPSK (quadrant certificate)
INSIDE = [dict(mod="qam", M=64, snr_db=25.0),                                              # BASELINE config 4
          dict(mod="qam", M=16, snr_db=18.0, cp_size=8, mmse=False, n_ofdm_sym=3),         # ZF, three OFDM symbols
          dict(mod="qam", M=256, snr_db=32.0, cp_size=0),
          dict(mod="qpsk", M=4, snr_db=8.0, cp_size=32, n_ofdm_sym=2),                     # quadrant certificate
          dict(mod="qam", M=4, snr_db=6.0, cp_size=144),
          dict(mod="psk", M=4, snr_db=8.0, cp_size=18)]                                   # the reference's PSK(4): on-axis certificate
# outside: partial band, odd prefix, a constellation without a certificate -> the planar kernel answers
OUTSIDE = [dict(mod="qam", M=64, snr_db=25.0, num_used=200, n_ofdm_sym=2),
           dict(mod="qam", M=16, snr_db=18.0, cp_size=7),
           dict(mod="psk", M=8, snr_db=14.0, n_ofdm_sym=2, cp_size=32)]


def _set(engine, kw):
    engine.set_constellation(chains.constellation(kw["mod"], kw["M"]), _lib.CONST_QAM if kw["mod"] == "qam" else _lib.CONST_GENERIC)


def _run(engine, kw, first, count, method, threads, fft=256):
    nv = 1.0 / omodem.dB2Linear(kw["snr_db"])
    with engine.options(f64_threads=threads):
        return engine.run_mimo_ofdm(4, 4, fft, kw.get("cp_size", 16), kw.get("num_used") or fft, kw.get("n_ofdm_sym", 1), nv, SEED,
                                    first, count, mmse=kw.get("mmse", True), method=method, dtype="f64", per_realization=True)


def _oracle(kw, first, count, fft=256):
    okw = dict(mod=kw["mod"], M=kw["M"], nt=4, nr=4, fft_size=fft, cp_size=kw.get("cp_size", 16), num_used=kw.get("num_used"),
               n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], mmse=kw.get("mmse", True))
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    return (np.array([w["symbol_errors"] for w in want]), np.array([w["bit_errors"] for w in want]), want[0]["num_symbols"],
            want[0]["num_bits"])


# (fft_size, the f64_threads values that select the round-6 kernel there, the planar kernel's)
SIZES = {256: ((0, 260, 262), 261), 512: ((0, 260, 262), 261), 1024: ((263, 264), 261), 2048: ((0, 260), 261)}


@pytest.mark.parametrize("fft", [256, 512, 1024, 2048])
@pytest.mark.parametrize("case", range(len(INSIDE) + len(OUTSIDE)))
def test_counts_equal_the_oracle(engine, case, fft):
    kw = (INSIDE + OUTSIDE)[case]
    _set(engine, kw)
    first, count = (1 << 36) + 1009, 21 if fft < 1024 else (6 if fft == 1024 else 3)
    want_se, want_be, nsym, nbits = _oracle(kw, first, count, fft)
    methods = [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if kw["mod"] == "qam" else [])
    for method in methods:
        for threads in SIZES[fft][0] + (SIZES[fft][1],):
            res, se, be = _run(engine, kw, first, count, method, threads, fft)
            assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (case, method, threads, se, want_se)
            assert res["n_realizations"] == count and res["n_skipped"] == 0 and res["n_symbols"] == nsym and res["n_bits"] == nbits
            assert res["sym_errors"] == int(want_se.sum()) and res["sym_errors_sq"] == int((want_se.astype(np.int64) ** 2).sum())


@pytest.mark.parametrize("fft", [256, 512, 1024, 2048])
@pytest.mark.parametrize("case", range(len(INSIDE)))
def test_equal_to_the_planar_kernel_over_two_thousand_realizations(engine, case, fft):
    """Both are complex128 statements of the same link with differently rounded transforms: a rounding-level tie may differ once
    in ~1e7 symbols.  Same sums of squares under any split of the range (launch slices are invisible)."""
    kw = INSIDE[case]
    _set(engine, kw)
    n = 2203 if fft < 1024 else (1100 if fft == 1024 else 600)
    new = SIZES[fft][0][0]
    for method in [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if kw["mod"] == "qam" else []):
        qw, se, be = _run(engine, kw, 77, n, method, new, fft)
        pl, se_p, be_p = _run(engine, kw, 77, n, method, 261, fft)
        assert np.count_nonzero(se != se_p) <= 1 and np.max(np.abs(se.astype(np.int64) - se_p.astype(np.int64))) <= 1
        assert np.count_nonzero(be != be_p) <= 1
        assert qw["n_realizations"] == pl["n_realizations"] == n and qw["n_skipped"] == pl["n_skipped"]
        a = _run(engine, kw, 77, 300, method, new, fft)[0]
        b = _run(engine, kw, 377, n - 300, method, new, fft)[0]
        for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations", "n_skipped"):
            assert qw[k] == a[k] + b[k], k
        again = _run(engine, kw, 77, n, method, new, fft)
        assert np.array_equal(again[1], se) and np.array_equal(again[2], be)             # bit-identical from run to run


@pytest.mark.parametrize("fft", [256, 512, 1024, 2048])
def test_benchmark_link_against_the_oracle_at_depth(engine, fft):
    """Config 4's link (4 x 4 MMSE, 64-QAM, OFDM(fft, 16), 25 dB) on the round-6 kernel of the size, both demodulators and both register
    bounds: every per-realization count of 4 096 (256), 2 048 (512), 1 024 (1024) consecutive realizations -- 4.2e6 symbols each -- equal
    to the oracle's."""
    kw = INSIDE[0]
    _set(engine, kw)
    first, count = 987654321, 4096 * 256 // fft
    want_se, want_be, _, _ = _oracle(kw, first, count, fft)
    assert want_se.sum() > 1e5
    for threads in SIZES[fft][0][-2:]:
        for method in (_lib.DEMOD_MINDIST, _lib.DEMOD_QAM_SLICER):
            res, se, be = _run(engine, kw, first, count, method, threads, fft)
            assert np.array_equal(se, want_se), (threads, method, np.flatnonzero(se != want_se)[:5])
            assert np.array_equal(be, want_be)
            assert res["sym_errors"] == int(want_se.sum()) and res["bit_errors"] == int(want_be.sum())


def test_singular_channels_are_skipped_like_the_planar_kernel(engine):
    """Zero forcing at infinite SNR over many realizations: whatever the record kernel flags as skipped is skipped by both."""
    kw = dict(mod="qam", M=16, snr_db=300.0, mmse=False)
    _set(engine, kw)
    for fft in (256, 512, 1024, 2048):
        a = _run(engine, kw, 0, 4096, _lib.DEMOD_MINDIST, SIZES[fft][0][0], fft)[0]
        b = _run(engine, kw, 0, 4096, _lib.DEMOD_MINDIST, 261, fft)[0]
        assert a["n_skipped"] == b["n_skipped"] and a["n_realizations"] == b["n_realizations"] and a["sym_errors"] == b["sym_errors"]
    a = _run(engine, kw, 0, 64, _lib.DEMOD_MINDIST, 260)[0]
    b = _run(engine, kw, 0, 64, _lib.DEMOD_MINDIST, 261)[0]
    assert a["n_skipped"] == b["n_skipped"] and a["n_realizations"] == b["n_realizations"] and a["sym_errors"] == b["sym_errors"]


# ---- 2 x 2 at fft_size 256: two realizations per wavefront (pipeline_mimo_fw.hip, NA = 2) ----
def _run22(engine, kw, first, count, method, threads):
    nv = 1.0 / omodem.dB2Linear(kw["snr_db"])
    with engine.options(f64_threads=threads):
        return engine.run_mimo_ofdm(2, 2, 256, kw.get("cp_size", 16), kw.get("num_used") or 256, kw.get("n_ofdm_sym", 1), nv, SEED,
                                    first, count, mmse=kw.get("mmse", True), method=method, dtype="f64", per_realization=True)


@pytest.mark.parametrize("case", range(len(INSIDE) + len(OUTSIDE)))
def test_two_by_two_counts_equal_the_oracle(engine, case):
    """Odd and even counts (an odd count leaves the last wavefront's second half without a realization), every corner of the envelope
    and requests outside it, all four kernel selections."""
    kw = (INSIDE + OUTSIDE)[case]
    _set(engine, kw)
    for first, count in (((1 << 36) + 77, 23), (5, 1), (900, 8)):
        okw = dict(mod=kw["mod"], M=kw["M"], nt=2, nr=2, fft_size=256, cp_size=kw.get("cp_size", 16), num_used=kw.get("num_used"),
                   n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], mmse=kw.get("mmse", True))
        want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
        want_se, want_be = np.array([w["symbol_errors"] for w in want]), np.array([w["bit_errors"] for w in want])
        for method in [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if kw["mod"] == "qam" else []):
            for threads in (0, 260, 262, 261):
                res, se, be = _run22(engine, kw, first, count, method, threads)
                assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (case, method, threads, se, want_se)
                assert res["n_realizations"] == count and res["n_skipped"] == 0 and res["n_symbols"] == want[0]["num_symbols"]
                assert res["sym_errors"] == int(want_se.sum()) and res["sym_errors_sq"] == int((want_se.astype(np.int64) ** 2).sum())


def test_two_by_two_against_the_oracle_and_the_planar_kernel_at_depth(engine):
    """(256, 2 x 2), 64-QAM, 22 dB: 4 097 realizations (an odd count) against the oracle, both demodulators and register bounds; the
    planar kernel on the same range; zero forcing at infinite SNR skips what the planar kernel skips."""
    kw = dict(mod="qam", M=64, snr_db=22.0)
    _set(engine, kw)
    first, count = 424242, 4097
    okw = dict(mod="qam", M=64, nt=2, nr=2, fft_size=256, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=22.0, mmse=True)
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se, want_be = np.array([w["symbol_errors"] for w in want]), np.array([w["bit_errors"] for w in want])
    assert want_se.sum() > 1e4
    for threads in (0, 262):
        for method in (_lib.DEMOD_MINDIST, _lib.DEMOD_QAM_SLICER):
            res, se, be = _run22(engine, kw, first, count, method, threads)
            assert np.array_equal(se, want_se), (threads, method, np.flatnonzero(se != want_se)[:5])
            assert np.array_equal(be, want_be)
    pl = _run22(engine, kw, first, count, _lib.DEMOD_MINDIST, 261)
    assert np.count_nonzero(pl[1] != want_se) <= 1
    zf = dict(mod="qam", M=16, snr_db=300.0, mmse=False)
    _set(engine, zf)
    a = _run22(engine, zf, 0, 4097, _lib.DEMOD_MINDIST, 0)[0]
    b = _run22(engine, zf, 0, 4097, _lib.DEMOD_MINDIST, 261)[0]
    assert a["n_skipped"] == b["n_skipped"] and a["n_realizations"] == b["n_realizations"] and a["sym_errors"] == b["sym_errors"]


# ---- complex64 at fft_size 256 (k_run_mimo_ofdm_fw<float, ...>: the contractions as reduce-scatters on the VALU) ----
@pytest.mark.parametrize("na", [4, 2])
@pytest.mark.parametrize("case", range(len(INSIDE)))
def test_complex64_full_wave_against_the_planar_kernel_and_the_oracle(engine, case, na):
    """The complex64 criteria of tests/test_gpu_planar_f32.py: |dSER| <= 1e-4 against the oracle on 24 realizations, per-realization
    differences of boundary ties only against the planar complex64 kernel (f64_threads = 261) and against the complex128 full-wave
    kernel over 1 203 realizations (an odd count: the 2 x 2 form's last wavefront is half empty), both register bounds."""
    kw = INSIDE[case]
    _set(engine, kw)
    nv = 1.0 / omodem.dB2Linear(kw["snr_db"])
    args = (na, na, 256, kw.get("cp_size", 16), 256, kw.get("n_ofdm_sym", 1), nv, SEED)
    okw = dict(mod=kw["mod"], M=kw["M"], nt=na, nr=na, fft_size=256, cp_size=kw.get("cp_size", 16), num_used=None,
               n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], mmse=kw.get("mmse", True))
    first, n = 4242, 1203
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + 24)]
    nsym = want[0]["num_symbols"]
    for method in [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if kw["mod"] == "qam" else []):
        run = lambda thr, dt: engine.run_mimo_ofdm(*args, first, n, mmse=kw.get("mmse", True), method=method, dtype=dt, per_realization=True) \
            if thr is None else None
        outs = {}
        for thr in (0, 262, 261):
            with engine.options(f64_threads=thr):
                outs[thr] = engine.run_mimo_ofdm(*args, first, n, mmse=kw.get("mmse", True), method=method, dtype="f32", per_realization=True)
        f64 = engine.run_mimo_ofdm(*args, first, n, mmse=kw.get("mmse", True), method=method, dtype="f64", per_realization=True)
        for thr in (0, 262):
            res, se, be = outs[thr]
            assert res["n_realizations"] + res["n_skipped"] == n and res["n_symbols"] == nsym
            assert abs(int(se[:24].sum()) - sum(w["symbol_errors"] for w in want)) <= 1e-4 * 24 * nsym + 3
            for other in (outs[261], f64):
                assert np.max(np.abs(se.astype(np.int64) - other[1].astype(np.int64))) <= 4
                assert abs(int(se.astype(np.int64).sum()) - int(other[1].astype(np.int64).sum())) <= 2e-5 * n * nsym + 4
                assert res["n_skipped"] == other[0]["n_skipped"]
