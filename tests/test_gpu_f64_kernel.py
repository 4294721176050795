"""GPU: the complex128 config-4 kernel family (csrc/pipeline_mimo_planar.hip: planar LDS, table Box-Muller, certified /
pruned min-distance search; fft_size 256 .. 2048, 2x2 / 4x4 / 2x4) -- per-realization error counts equal to the oracle
chain's under the same Philox keying, and equal to the generic radix-4 kernel it replaces (engine option f64_generic), on
every corner of its envelope."""
import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 777001

CASES = [dict(mod="qam", M=64, snr_db=25.0),                                               # BASELINE config 4
         dict(mod="qam", M=64, snr_db=25.0, num_used=600, n_ofdm_sym=2),                   # partial band, two OFDM symbols
         dict(mod="qam", M=16, snr_db=18.0, cp_size=7, mmse=False),                        # odd CP: unpaired noise draws; ZF
         dict(mod="qam", M=256, snr_db=32.0, cp_size=0),
         dict(mod="psk", M=8, snr_db=14.0, num_used=1022, n_ofdm_sym=3, cp_size=33),       # no slicer: candidate grid only
         dict(mod="qpsk", M=4, snr_db=8.0, num_used=2)]


def _run(engine, kw, first, count, method, generic=False, threads=0, variant=0):
    nv = 1.0 / omodem.dB2Linear(kw["snr_db"])
    with engine.options(f64_generic=1 if generic else 0, f64_threads=threads, f64_variant=variant):
        return engine.run_mimo_ofdm(4, 4, 1024, kw.get("cp_size", 16), kw.get("num_used") or 1024, kw.get("n_ofdm_sym", 1),
                                    nv, SEED, first, count, mmse=kw.get("mmse", True), method=method, dtype="f64",
                                    per_realization=True)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_f64_kernel_counts_equal_the_oracle(engine, case):
    kw = CASES[case]
    kind = _lib.CONST_QAM if kw["mod"] == "qam" else _lib.CONST_GENERIC
    engine.set_constellation(chains.constellation(kw["mod"], kw["M"]), kind)
    first, count = (1 << 35) + 17, 5
    okw = dict(mod=kw["mod"], M=kw["M"], nt=4, nr=4, fft_size=1024, cp_size=kw.get("cp_size", 16),
               num_used=kw.get("num_used"), n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], mmse=kw.get("mmse", True))
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want])
    want_be = np.array([w["bit_errors"] for w in want])
    methods = [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if kw["mod"] == "qam" else [])
    for method in methods:
        # radix-16 passes with the fused middle stage (default) / with the separate channel stage / radix-4: two antennas per
        # thread / four antennas per thread
        for threads, variant in ((0, 0), (261, 0), (257, 0), (512, 0), (256, 0)):
            res, se, be = _run(engine, kw, first, count, method, threads=threads, variant=variant)
            assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (case, method, threads, variant, se, want_se)
            assert res["n_realizations"] == count and res["sym_errors"] == int(want_se.sum())
            assert res["n_symbols"] == want[0]["num_symbols"] and res["n_bits"] == want[0]["num_bits"]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_f64_kernel_equals_the_generic_kernel(engine, case):
    """1 000+ realizations per case against k_run_mimo_ofdm<double, 1024, 4>: the same counts realization by realization
    (both are complex128 statements of the same link; a rounding-level tie may differ once in ~1e7 symbols), the same
    sums of squares, any split of the range."""
    kw = CASES[case]
    kind = _lib.CONST_QAM if kw["mod"] == "qam" else _lib.CONST_GENERIC
    engine.set_constellation(chains.constellation(kw["mod"], kw["M"]), kind)
    n = 1031
    method = _lib.DEMOD_MINDIST
    new, se, be = _run(engine, kw, 5, n, method, threads=512)
    alt, se_a, be_a = _run(engine, kw, 5, n, method, threads=256)
    assert np.array_equal(se, se_a) and np.array_equal(be, be_a) and new == alt          # same arithmetic, other thread map
    r16, se_r, be_r = _run(engine, kw, 5, n, method, threads=261)   # radix-16 passes (planar): one more rounding per layer-1 twiddle
    assert np.count_nonzero(se != se_r) <= 1 and np.max(np.abs(se.astype(int) - se_r.astype(int))) <= 1
    assert r16["n_realizations"] == new["n_realizations"] and r16["n_skipped"] == new["n_skipped"]
    old, se_o, be_o = _run(engine, kw, 5, n, method, generic=True)
    assert np.count_nonzero(se != se_o) <= 1 and np.max(np.abs(se.astype(int) - se_o.astype(int))) <= 1
    assert abs(new["sym_errors"] - old["sym_errors"]) <= 1 and new["n_realizations"] == old["n_realizations"] == n
    a = _run(engine, kw, 5, 400, method)[0]
    b = _run(engine, kw, 405, n - 400, method)[0]
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations", "n_skipped"):
        assert r16[k] == a[k] + b[k], k


def test_f64_kernel_against_the_oracle_over_2000_realizations(engine):
    """The headline geometry (BASELINE config 4: 4x4 MMSE, 64-QAM, OFDM(1024, 16), 25 dB), both thread maps, both demodulators,
    the certificate on and off: per-realization symbol AND bit error counts equal to the oracle chain's on 2 048 consecutive
    realizations (8.4e6 symbols; the bench's ser_abs_err_vs_oracle compares sums only)."""
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    kw = CASES[0]
    first, count = 123456789, 2048
    okw = dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=25.0, mmse=True)
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want])
    want_be = np.array([w["bit_errors"] for w in want])
    assert want_se.sum() > 1e5                                  # a realization in outage is thousands of errors
    for threads, variant in ((0, 0), (261, 0), (257, 0), (512, 0), (256, 0)):
        for method, nocert in ((_lib.DEMOD_MINDIST, 0), (_lib.DEMOD_MINDIST, 1), (_lib.DEMOD_QAM_SLICER, 0)):
            with engine.options(demod_nocert=nocert):
                res, se, be = _run(engine, kw, first, count, method, threads=threads, variant=variant)
            assert np.array_equal(se, want_se), (threads, variant, method, nocert, np.flatnonzero(se != want_se)[:5])
            assert np.array_equal(be, want_be), (threads, method, nocert)
            assert res["sym_errors"] == int(want_se.sum()) and res["bit_errors"] == int(want_be.sum())


# ---- round 4: the kernel as a family (fft_size 256 .. 2048, 2x2 / 4x4 / 2x4), not a benchmark point ----
SHAPES = [(256, 2, 2), (256, 4, 4), (512, 2, 2), (512, 4, 4), (1024, 2, 2), (2048, 2, 2), (2048, 4, 4), (1024, 2, 4), (256, 2, 4),
          # Nt < Nr and three receive antennas (Blast takes any Nr x Nt, mimo/mimo.py:264-309)
          (512, 1, 2), (1024, 3, 4), (2048, 1, 4), (256, 2, 3), (1024, 3, 3), (512, 1, 3), (2048, 3, 4)]
SHAPE_CASES = [dict(mod="qam", M=64, snr_db=25.0),
               dict(mod="qam", M=16, snr_db=17.0, used_frac=0.6, n_ofdm_sym=2, cp_size=7, mmse=False),   # partial band, odd CP, ZF
               dict(mod="psk", M=8, snr_db=13.0, n_ofdm_sym=2, cp_size=33)]                               # candidate grid only


def _shape_kwargs(kw, fft, nt, nr):
    used = fft if "used_frac" not in kw else 2 * int(kw["used_frac"] * fft / 2)
    return dict(mod=kw["mod"], M=kw["M"], nt=nt, nr=nr, fft_size=fft, cp_size=kw.get("cp_size", 16), num_used=used,
                n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], mmse=kw.get("mmse", True))


def _run_shape(engine, okw, first, count, method, generic=False):
    nv = 1.0 / omodem.dB2Linear(okw["snr_db"])
    with engine.options(f64_generic=1 if generic else 0):
        return engine.run_mimo_ofdm(okw["nt"], okw["nr"], okw["fft_size"], okw["cp_size"], okw["num_used"], okw["n_ofdm_sym"],
                                    nv, SEED, first, count, mmse=okw["mmse"], method=method, dtype="f64", per_realization=True)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%dx%d" % s)
@pytest.mark.parametrize("case", range(len(SHAPE_CASES)))
def test_f64_family_counts_equal_the_oracle(engine, shape, case):
    """Every geometry of the family against the oracle chain realization by realization (both demodulators), and -- where
    the generic kernel k_run_mimo_ofdm<double, N, NA> covers the shape -- against that one over 300 realizations."""
    fft, nt, nr = shape
    okw = _shape_kwargs(SHAPE_CASES[case], fft, nt, nr)
    kind = _lib.CONST_QAM if okw["mod"] == "qam" else _lib.CONST_GENERIC
    engine.set_constellation(chains.constellation(okw["mod"], okw["M"]), kind)
    first, count = (1 << 33) + 5, 4
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want])
    want_be = np.array([w["bit_errors"] for w in want])
    methods = [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if okw["mod"] == "qam" else [])
    for method in methods:
        res, se, be = _run_shape(engine, okw, first, count, method)
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (shape, case, method, se, want_se)
        assert res["n_symbols"] == want[0]["num_symbols"] and res["n_bits"] == want[0]["num_bits"]
    if nt == nr and nr != 3 and (fft, nr) != (2048, 4):       # (the generic kernel: 2x2 / 4x4 only, and 181 KiB of LDS at 2048 x 4)
        n = 300
        new, se, be = _run_shape(engine, okw, 9, n, _lib.DEMOD_MINDIST)
        old, se_o, be_o = _run_shape(engine, okw, 9, n, _lib.DEMOD_MINDIST, generic=True)
        assert np.count_nonzero(se != se_o) <= 1 and np.max(np.abs(se.astype(int) - se_o.astype(int))) <= 1
        assert new["n_realizations"] == old["n_realizations"] == n and new["n_skipped"] == old["n_skipped"]


def test_shapes_outside_the_envelope_are_refused(engine):
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    for nt, nr, fft, dtype in [(4, 2, 1024, "f64"), (2, 2, 96, "f64"), (5, 5, 1024, "f64"),
                               (2, 4, 128, "f64")]:
        with pytest.raises((_lib.McleError, ValueError)):
            engine.run_mimo_ofdm(nt, nr, fft, 16, fft, 1, 0.01, SEED, 0, 4, dtype=dtype)


def test_f64_kernel_skips_singular_channels_like_the_generic_kernel(engine):
    """Zero forcing at a very high SNR: no errors, and the skip flag path (non-positive pivot) matches."""
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    kw = dict(mod="qam", M=16, snr_db=300.0, mmse=False)
    new = _run(engine, kw, 0, 3000, _lib.DEMOD_QAM_SLICER)[0]
    old = _run(engine, kw, 0, 3000, _lib.DEMOD_QAM_SLICER, generic=True)[0]
    assert new["n_skipped"] == old["n_skipped"] and new["n_realizations"] == old["n_realizations"]
    assert new["sym_errors"] == old["sym_errors"] <= 3


def test_launch_slices_are_invisible(engine):
    """The two-launch pipelines bound their record buffers by running (records, link) pairs over slices of the realization
    range -- 2^18 realizations for the complex128 config-4 kernel, 64 MiB of fading records for config 3: a call that crosses
    slice boundaries must give the counters of the same range taken in pieces (and per-realization counts in the same
    order)."""
    from pyphysim_amd.channels import discretize_profile
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    nv = 1.0 / omodem.dB2Linear(25.0)
    n = (1 << 18) + 4099                       # crosses the slice of k_mimo_filters_planar / k_run_mimo_ofdm_planar
    whole, se, _ = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, 0, n, method=_lib.DEMOD_QAM_SLICER, dtype="f64",
                                        per_realization=True)
    a = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, 0, 1 << 18, method=_lib.DEMOD_QAM_SLICER, dtype="f64")
    b, se_b, _ = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, 1 << 18, 4099, method=_lib.DEMOD_QAM_SLICER,
                                      dtype="f64", per_realization=True)
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations", "n_skipped"):
        assert whole[k] == a[k] + b[k], k
    assert np.array_equal(se[1 << 18:], se_b)
    # config 3, complex64: 160 B of fading record per realization -> a slice is 419 428 realizations
    engine.set_constellation(chains.constellation("qpsk", 4), _lib.CONST_GENERIC)
    Ts = 1.0 / (15e3 * 1024)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    args = (1024, 16, 1024, 1, 0.01, p_lin, d_idx, SEED)
    n3 = 419428 + 8191
    w3, s3, _ = engine.run_ofdm_tdl(*args, 3, n3, Fd=10.0, Ts=Ts, L=8, dtype="f32", per_realization=True)
    a3 = engine.run_ofdm_tdl(*args, 3, 419428, Fd=10.0, Ts=Ts, L=8, dtype="f32")
    b3, sb3, _ = engine.run_ofdm_tdl(*args, 3 + 419428, 8191, Fd=10.0, Ts=Ts, L=8, dtype="f32", per_realization=True)
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations"):
        assert w3[k] == a3[k] + b3[k], k
    assert np.array_equal(s3[419428:], sb3)
