"""GPU: the fused on-chip pipelines against the NumPy oracle under COMMON RANDOM NUMBERS
(identical mcle-philox-v1 keying on both sides), plus size-independent properties at the full
BASELINE.json sizes.  f64 instantiation: per-realization error counts bit-exact; f32: |dSER| and
|dBER| <= 1e-4 absolute (north_star tolerance)."""
import json
import os

import numpy as np
import pytest

from oracle import chains, channels as och, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 20260927


def oracle_counts(fn, first, count, **kw):
    se, be = [], []
    for r in range(first, first + count):
        out = fn(chains.PhiloxRng(SEED, r), **kw)
        se.append(out["symbol_errors"])
        be.append(out["bit_errors"])
    return np.array(se), np.array(be), out["num_symbols"], out["num_bits"]


def check(res, se, be, want_se, want_be, nsym, nbits, exact):
    cnt = res
    assert cnt["n_realizations"] == len(want_se) and cnt["n_skipped"] == 0
    assert cnt["n_symbols"] == nsym and cnt["n_bits"] == nbits
    assert cnt["sym_errors"] == int(se.sum()) and cnt["bit_errors"] == int(be.sum())
    assert cnt["sym_errors_sq"] == int((se.astype(np.int64) ** 2).sum())
    assert cnt["bit_errors_sq"] == int((be.astype(np.int64) ** 2).sum())
    if exact:
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be)
    else:
        n = len(want_se)
        assert abs(int(se.sum()) - int(want_se.sum())) / (n * nsym) <= 1e-4
        assert abs(int(be.sum()) - int(want_be.sum())) / (n * nbits) <= 1e-4


@pytest.mark.parametrize("dt,exact", [("f64", True), ("f32", False)])
@pytest.mark.parametrize("mod,M,N,snr", [("qam", 16, 10000, 10.0), ("qam", 64, 777, 18.0), ("psk", 8, 4096, 9.0),
                                         ("bpsk", 2, 50, 2.0)])
def test_awgn_pipeline(engine, dt, exact, mod, M, N, snr):
    table = chains.constellation(mod, M)
    engine.set_constellation(table, _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)
    first, count = 1000003, 6
    want_se, want_be, nsym, nbits = oracle_counts(chains.chain_awgn, first, count, mod=mod, M=M, N=N, snr_db=snr)
    nv = 1.0 / omodem.dB2Linear(snr)
    res, se, be = engine.run_awgn(N, nv, SEED, first, count, dtype=dt, per_realization=True)
    check(res, se, be, want_se, want_be, nsym, nbits, exact)
    if mod == "qam":
        res2, se2, be2 = engine.run_awgn(N, nv, SEED, first, count, method=_lib.DEMOD_QAM_SLICER, dtype=dt,
                                         per_realization=True)
        assert np.max(np.abs(se2.astype(int) - se.astype(int))) <= (0 if exact else 1)


@pytest.mark.parametrize("dt,exact", [("f64", True), ("f32", False)])
@pytest.mark.parametrize("L", [8, 16, 5])      # 8 / 16: the rotation-recurrence instantiations; 5: one sincos per sample
def test_flat_fading_pipeline(engine, dt, exact, L):
    kw = dict(mod="qam", M=64, N=20000, snr_db=20.0, Fd=100.0, Ts=1e-3, L=L)
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    first, count = 77, 3
    want_se, want_be, nsym, nbits = oracle_counts(chains.chain_flat_jakes, first, count, **kw)
    res, se, be = engine.run_flat_fading(kw["N"], 1.0 / omodem.dB2Linear(20.0), SEED, first, count, Fd=100.0,
                                         Ts=1e-3, L=L, dtype=dt, per_realization=True)
    check(res, se, be, want_se, want_be, nsym, nbits, exact)


def test_flat_fading_pipeline_at_the_full_config2_size(engine):
    """VERDICT r02 (parity soft spot): BASELINE config 2 at its own size -- 64-QAM, 1e5 symbols per realization, Jakes
    Fd 100 Hz / Ts 1 ms / L 8, i.e. all seven 16 384-symbol chunks of a realization and Jakes phases out to t = 100 s --
    against oracle.chains.chain_flat_jakes on the same Philox draws: complex128 counts exact (both demodulators), complex64
    (matrix-core kernel and the VALU kernel) within 1e-4."""
    kw = dict(mod="qam", M=64, N=100000, snr_db=20.0, Fd=100.0, Ts=1e-3, L=8)
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    first, count = (1 << 34) + 3, 2
    want_se, want_be, nsym, nbits = oracle_counts(chains.chain_flat_jakes, first, count, **kw)
    nv = 1.0 / omodem.dB2Linear(20.0)
    for method in (_lib.DEMOD_MINDIST, _lib.DEMOD_QAM_SLICER):
        res, se, be = engine.run_flat_fading(100000, nv, SEED, first, count, Fd=100.0, Ts=1e-3, L=8, method=method,
                                             dtype="f64", per_realization=True)
        check(res, se, be, want_se, want_be, nsym, nbits, True)
    with engine.options(jakes_direct=1):    # one sincos per ray and sample instead of the 16-symbol rotation recurrence
        res, se, be = engine.run_flat_fading(100000, nv, SEED, first, count, Fd=100.0, Ts=1e-3, L=8, dtype="f64",
                                             per_realization=True)
    check(res, se, be, want_se, want_be, nsym, nbits, True)
    for no_mfma in (0, 1):
        with engine.options(no_mfma=no_mfma):
            res, se, be = engine.run_flat_fading(100000, nv, SEED, first, count, Fd=100.0, Ts=1e-3, L=8, dtype="f32",
                                                 per_realization=True)
        check(res, se, be, want_se, want_be, nsym, nbits, False)
        assert np.max(np.abs(se.astype(int) - want_se)) <= 6          # a handful of rounding-level ties in 1e5 symbols


@pytest.mark.parametrize("dt,exact", [("f64", True), ("f32", False)])
@pytest.mark.parametrize("mod,M,N,snr", [("qam", 16, 1000, 15.0), ("qam", 64, 20000, 28.0), ("psk", 8, 4097, 16.0),
                                         ("bpsk", 2, 333, 5.0)])
def test_flat_rayleigh_pipeline(engine, dt, exact, mod, M, N, snr):
    """a7: the rayleigh_iid branch of the single-carrier kernel against oracle.chains.chain_flat_rayleigh under the
    same Philox keying (h = CHAN sample n, noise = NOISE sample n): f64 counts exact, f32 |dSER| <= 1e-4; both of
    the oracle's forms address the same draws; slicer == min-distance on QAM."""
    engine.set_constellation(chains.constellation(mod, M), _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)
    first, count = 900001, 5
    kw = dict(mod=mod, M=M, N=N, snr_db=snr)
    want_se, want_be, nsym, nbits = oracle_counts(chains.chain_flat_rayleigh, first, count, **kw)
    nb_se, nb_be, _, _ = oracle_counts(chains.chain_flat_rayleigh, first, count, form="notebook", **kw)
    assert np.array_equal(want_se, nb_se) and np.array_equal(want_be, nb_be)
    nv = 1.0 / omodem.dB2Linear(snr)
    res, se, be = engine.run_flat_fading(N, nv, SEED, first, count, rayleigh_iid=True, dtype=dt, per_realization=True)
    check(res, se, be, want_se, want_be, nsym, nbits, exact)
    if mod == "qam":
        res2, se2, be2 = engine.run_flat_fading(N, nv, SEED, first, count, rayleigh_iid=True,
                                                method=_lib.DEMOD_QAM_SLICER, dtype=dt, per_realization=True)
        assert np.max(np.abs(se2.astype(int) - se.astype(int))) <= (0 if exact else 2)
    # split invariance: the counters of two half ranges add up to the whole range's
    a = engine.run_flat_fading(N, nv, SEED, first, 2, rayleigh_iid=True, dtype=dt)
    b = engine.run_flat_fading(N, nv, SEED, first + 2, 3, rayleigh_iid=True, dtype=dt)
    assert a["sym_errors"] + b["sym_errors"] == res["sym_errors"]
    assert a["bit_errors_sq"] + b["bit_errors_sq"] == res["bit_errors_sq"]


@pytest.mark.parametrize("dt,exact", [("f64", True), ("f32", False)])
@pytest.mark.parametrize("case", [0, 1, 2, 3, 4])
def test_ofdm_tdl_pipeline(engine, dt, exact, case):
    kws = [dict(mod="qpsk", M=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=20.0, Fd=10.0,
                Ts=1.0 / (15e3 * 1024), L=8, tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0),
                tap_delays_samples=(0, 1, 2, 3, 4)),
           dict(mod="qam", M=16, fft_size=64, cp_size=16, num_used=52, n_ofdm_sym=3, snr_db=25.0, Fd=50.0, Ts=1e-6,
                L=8, tap_powers_dB=(0.0, -5.0, -10.0), tap_delays_samples=(0, 3, 7)),
           # delays beyond the cyclic prefix: inter-symbol interference from the previous OFDM symbol
           dict(mod="qam", M=16, fft_size=128, cp_size=4, num_used=100, n_ofdm_sym=4, snr_db=30.0, Fd=200.0, Ts=2e-6,
                L=12, tap_powers_dB=(0.0, -2.0, -6.0, -12.0), tap_delays_samples=(0, 3, 9, 20)),
           # the same through the four-realizations-per-pass kernel (FFT 256): ISI via its tail buffer
           dict(mod="qam", M=16, fft_size=256, cp_size=4, num_used=200, n_ofdm_sym=3, snr_db=30.0, Fd=200.0, Ts=2e-6,
                L=12, tap_powers_dB=(0.0, -2.0, -6.0, -12.0), tap_delays_samples=(0, 3, 9, 20)),
           # Doppler beyond the polynomial tap model: falls back to the single-realization kernel
           dict(mod="psk", M=8, fft_size=64, cp_size=8, num_used=None, n_ofdm_sym=2, snr_db=22.0, Fd=3000.0, Ts=1e-5,
                L=8, tap_powers_dB=(0.0, -6.0), tap_delays_samples=(0, 2))]
    kw = kws[case]
    engine.set_constellation(chains.constellation(kw["mod"], kw["M"]),
                             _lib.CONST_QAM if kw["mod"] == "qam" else _lib.CONST_GENERIC)
    first, count = 5, 7              # not a multiple of the four slots of a pass
    want_se, want_be, nsym, nbits = oracle_counts(chains.chain_ofdm_tdl, first, count, **kw)
    p_lin, d_idx = och.discretize_profile(np.array(kw["tap_powers_dB"]), np.array(kw["tap_delays_samples"]) * kw["Ts"],
                                          kw["Ts"])
    res, se, be = engine.run_ofdm_tdl(kw["fft_size"], kw["cp_size"], kw["num_used"] or kw["fft_size"],
                                      kw["n_ofdm_sym"], 1.0 / omodem.dB2Linear(kw["snr_db"]), p_lin, d_idx, SEED,
                                      first, count, Fd=kw["Fd"], Ts=kw["Ts"], L=kw["L"], dtype=dt,
                                      per_realization=True)
    check(res, se, be, want_se, want_be, nsym, nbits, exact)


def test_ofdm_tdl_batched_kernel_equals_single(engine):
    """Config 3 runs four realizations per workgroup pass (pipeline_siso_tdl.hip); the engine option single_tdl forces the
    single-realization kernel.  Same per-realization counts in f64 for every count mod 4, near-identical in f32,
    and sums independent of how a range is split."""
    engine.set_constellation(chains.constellation("qpsk", 4), _lib.CONST_GENERIC)
    Ts = 1.0 / (15e3 * 1024)
    p_lin, d_idx = och.discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    args = (1024, 16, 1024, 1, 0.01, p_lin, d_idx, SEED)

    def run(first, count, dt, single):
        with engine.options(single_tdl=1 if single else 0):
            return engine.run_ofdm_tdl(*args, first, count, Fd=10.0, Ts=Ts, L=8, dtype=dt, per_realization=True)
    for count in (1, 2, 3, 4, 5, 1027):
        rb, sb, bb = run(11, count, "f64", False)
        rs, ss, bs = run(11, count, "f64", True)
        assert np.array_equal(sb, ss) and np.array_equal(bb, bs) and rb["n_realizations"] == count
        assert rb["sym_errors"] == rs["sym_errors"] and rb["sym_errors_sq"] == rs["sym_errors_sq"]
    rb, sb, _ = run(0, 4099, "f32", False)
    rs, ss, _ = run(0, 4099, "f32", True)
    assert np.abs(sb.astype(int) - ss.astype(int)).max() <= 2
    assert abs(rb["sym_errors"] - rs["sym_errors"]) <= 1e-4 * 4099 * 1024
    a = run(0, 4099, "f32", False)[0]
    b = run(0, 1234, "f32", False)[0]
    c = run(1234, 2865, "f32", False)[0]
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "n_realizations"):
        assert a[k] == b[k] + c[k]


@pytest.mark.parametrize("fft,cp,used,nsym", [(64, 5, 52, 3), (128, 9, 100, 2), (256, 20, 256, 2), (512, 36, 300, 2),
                                              (2048, 144, 1200, 1)])
def test_ofdm_tdl_every_fft_size(engine, fft, cp, used, nsym):
    """Every power-of-two OFDM size: the batched kernel (64 ... 2048) and the single-realization kernel it falls
    back to (other sizes, or when the four-realization pass does not fit LDS) give the oracle's counts in f64 --
    ragged used-subcarrier counts, several OFDM symbols, a CP shorter than the delay spread."""
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    Ts = 1e-6
    kw = dict(mod="qam", M=16, fft_size=fft, cp_size=cp, num_used=used, n_ofdm_sym=nsym, snr_db=22.0, Fd=60.0, Ts=Ts,
              L=8, tap_powers_dB=(0.0, -4.0, -8.0), tap_delays_samples=(0, 3, 7))
    p_lin, d_idx = och.discretize_profile(np.array(kw["tap_powers_dB"]), np.array(kw["tap_delays_samples"]) * Ts, Ts)
    first, count = 2, 6
    want_se, want_be, n_sym, n_bits = oracle_counts(chains.chain_ofdm_tdl, first, count, **kw)
    for single in (False, True):
        with engine.options(single_tdl=1 if single else 0):
            res, se, be = engine.run_ofdm_tdl(fft, cp, used, nsym, 1.0 / omodem.dB2Linear(22.0), p_lin, d_idx, SEED, first,
                                              count, Fd=60.0, Ts=Ts, L=8, dtype="f64", per_realization=True)
        check(res, se, be, want_se, want_be, n_sym, n_bits, True)
    if fft == 64:
        for bad in (32, 4096, 96):                      # outside the fused kernels: a clear error, never a wrong answer
            with pytest.raises(Exception, match="fft_size"):
                engine.run_ofdm_tdl(bad, 8, bad, 1, 0.01, p_lin, d_idx, SEED, 0, 4, Fd=60.0, Ts=Ts, L=8)


@pytest.mark.parametrize("dt,exact", [("f64", True), ("f32", False)])
@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
def test_mimo_ofdm_pipeline(engine, dt, exact, case):
    kws = [dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=25.0,
                mmse=True),
           dict(mod="qam", M=16, nt=2, nr=2, fft_size=64, cp_size=8, num_used=48, n_ofdm_sym=2, snr_db=15.0,
                mmse=False),
           dict(mod="qam", M=16, nt=4, nr=4, fft_size=256, cp_size=7, num_used=200, n_ofdm_sym=2, snr_db=18.0,
                mmse=True),
           dict(mod="qam", M=16, nt=2, nr=2, fft_size=128, cp_size=9, num_used=100, n_ofdm_sym=2, snr_db=16.0,
                mmse=True),
           dict(mod="qam", M=64, nt=4, nr=4, fft_size=512, cp_size=36, num_used=300, n_ofdm_sym=1, snr_db=24.0,
                mmse=True),
           dict(mod="qam", M=16, nt=2, nr=2, fft_size=2048, cp_size=144, num_used=1200, n_ofdm_sym=1, snr_db=17.0,
                mmse=False)]
    kw = kws[case]
    engine.set_constellation(chains.constellation(kw["mod"], kw["M"]), _lib.CONST_QAM)
    first, count = 31, 6
    want_se, want_be, nsym, nbits = oracle_counts(chains.chain_mimo_ofdm, first, count, **kw)
    res, se, be = engine.run_mimo_ofdm(kw["nt"], kw["nr"], kw["fft_size"], kw["cp_size"],
                                       kw["num_used"] or kw["fft_size"], kw["n_ofdm_sym"],
                                       1.0 / omodem.dB2Linear(kw["snr_db"]), SEED, first, count, mmse=kw["mmse"],
                                       dtype=dt, per_realization=True)
    check(res, se, be, want_se, want_be, nsym, nbits, exact)
    res2, se2, be2 = engine.run_mimo_ofdm(kw["nt"], kw["nr"], kw["fft_size"], kw["cp_size"],
                                          kw["num_used"] or kw["fft_size"], kw["n_ofdm_sym"],
                                          1.0 / omodem.dB2Linear(kw["snr_db"]), SEED, first, count, mmse=kw["mmse"],
                                          method=_lib.DEMOD_QAM_SLICER, dtype=dt, per_realization=True)
    assert np.max(np.abs(se2.astype(int) - se.astype(int))) <= (0 if exact else 1)


def test_full_size_properties_c4(engine):
    """BASELINE config 4 shape (4x4, 64-QAM, OFDM-1024) at batch sizes the oracle cannot reach."""
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    args = dict(nt=4, nr=4, fft_size=1024, cp_size=16, num_used=1024, n_ofdm_sym=1)
    nv = 1.0 / omodem.dB2Linear(25.0)
    n = 20000
    whole, se, be = engine.run_mimo_ofdm(noise_var=nv, seed=SEED, first=0, count=n, dtype="f32",
                                         per_realization=True, **args)
    # (1) shard invariance: any split of the realization range gives identical integer counters
    parts = [engine.run_mimo_ofdm(noise_var=nv, seed=SEED, first=a, count=b - a, dtype="f32", **args)
             for a, b in ((0, 1), (1, 7777), (7777, n))]
    for k in ("n_realizations", "sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_skipped"):
        assert whole[k] == sum(p[k] for p in parts), k
    # (2) same call twice is bit-identical (no order-dependent arithmetic in the counters)
    again = engine.run_mimo_ofdm(noise_var=nv, seed=SEED, first=0, count=n, dtype="f32", **args)
    assert again == whole
    # (3) slicer and exhaustive minimum distance agree on every realization
    _, se_s, be_s = engine.run_mimo_ofdm(noise_var=nv, seed=SEED, first=0, count=n, dtype="f32",
                                         method=_lib.DEMOD_QAM_SLICER, per_realization=True, **args)
    # (f32: the two decision rules round differently on symbols that sit on a boundary to ~1e-7 of the grid
    # spacing: of the 8.2e7 symbols here ~ 8.2e7 * 3e-7 = 25 are expected to; f64 below agrees exactly)
    assert np.count_nonzero(se != se_s) <= 80 and abs(int(se.sum()) - int(se_s.sum())) <= 80
    assert abs(int(be.sum()) - int(be_s.sum())) <= 160
    e64 = [engine.run_mimo_ofdm(noise_var=nv, seed=SEED, first=0, count=200, dtype="f64", method=m,
                                per_realization=True, **args)[1] for m in (_lib.DEMOD_MINDIST, _lib.DEMOD_QAM_SLICER)]
    assert np.array_equal(e64[0], e64[1])
    # (4) round trip: without noise zero forcing recovers every symbol
    clean = engine.run_mimo_ofdm(noise_var=0.0, seed=SEED, first=0, count=2000, dtype="f32", mmse=False, **args)
    assert clean["sym_errors"] == 0 and clean["bit_errors"] == 0 and clean["n_realizations"] + clean["n_skipped"] == 2000
    # (5) sanity of the statistic itself: SER at 25 dB for this chain is ~0.27
    ser = whole["sym_errors"] / (whole["n_realizations"] * whole["n_symbols"])
    assert 0.2 < ser < 0.35 and whole["bit_errors"] >= whole["sym_errors"]
    # (6) a different seed gives different counters, a different first index too
    other = engine.run_mimo_ofdm(noise_var=nv, seed=SEED + 1, first=0, count=n, dtype="f32", **args)
    assert other["sym_errors"] != whole["sym_errors"]


def test_full_size_properties_c2_c3(engine):
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    nv = 1.0 / omodem.dB2Linear(20.0)
    a = engine.run_flat_fading(100000, nv, SEED, 0, 24, dtype="f32")
    b = engine.run_flat_fading(100000, nv, SEED, 0, 10, dtype="f32")
    c = engine.run_flat_fading(100000, nv, SEED, 10, 14, dtype="f32")
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations"):
        assert a[k] == b[k] + c[k]
    ser = a["sym_errors"] / (24 * 100000)
    assert 0.1 < ser < 0.35                              # 0.22 at 20 dB
    clean = engine.run_flat_fading(100000, 0.0, SEED, 0, 4, dtype="f32")
    assert clean["sym_errors"] == 0
    engine.set_constellation(chains.constellation("qpsk", 4))
    p_lin, d_idx = och.discretize_profile(np.array([0.0, -3, -6, -9, -12]), np.arange(5) / (15e3 * 1024),
                                          1.0 / (15e3 * 1024))
    w = engine.run_ofdm_tdl(1024, 16, 1024, 1, nv, p_lin, d_idx, SEED, 0, 30000, dtype="f32")
    w1 = engine.run_ofdm_tdl(1024, 16, 1024, 1, nv, p_lin, d_idx, SEED, 0, 12345, dtype="f32")
    w2 = engine.run_ofdm_tdl(1024, 16, 1024, 1, nv, p_lin, d_idx, SEED, 12345, 30000 - 12345, dtype="f32")
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations"):
        assert w[k] == w1[k] + w2[k]
    assert 0.001 < w["sym_errors"] / (30000 * 1024) < 0.05   # 0.009
    clean = engine.run_ofdm_tdl(1024, 16, 1024, 1, 0.0, p_lin, d_idx, SEED, 0, 500, dtype="f32")
    assert clean["sym_errors"] == 0


def test_pipeline_argument_errors(engine):
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    with pytest.raises(_lib.McleError):
        engine.run_mimo_ofdm(3, 3, 128, 16, 128, 1, 0.1, SEED, 0, 4, dtype="f32")       # (3x3: the planar family only, fft_size >= 256)
    with pytest.raises(_lib.McleError):
        engine.run_mimo_ofdm(4, 3, 1024, 16, 1024, 1, 0.1, SEED, 0, 4)                  # more transmit than receive antennas
    with pytest.raises(_lib.McleError):
        engine.run_mimo_ofdm(4, 4, 1024, 2000, 1024, 1, 0.1, SEED, 0, 4)
    with pytest.raises(_lib.McleError):
        engine.run_awgn(0, 0.1, SEED, 0, 4)
    assert engine.run_awgn(100, 0.1, SEED, 0, 0)["n_realizations"] == 0


# ---- config 5: closed-form interference alignment ------------------------------------------------
def test_ia_closed_form_injected(engine):
    """ClosedFormIASolver on the reference's own channels: precoders, filters, SINR, capacity and --
    through the staged operators -- the decisions of apps/ia/simulate_ia.py."""
    from helpers import golden_cases, relerr
    for kw, reals in golden_cases("c5_ia"):
        H = np.stack([g["big_H"] for g in reals])
        nv = float(reals[0]["noise_var"])
        sol = engine.ia_closed_form(H, nv)
        assert not sol["skipped"].any()
        for b, g in enumerate(reals):
            assert relerr(sol["F"][b], g["F"]) <= 1e-9 and relerr(sol["U"][b], g["U"]) <= 1e-9
            assert relerr(sol["sinr"][b], g["sinr"]) <= 1e-8 and abs(sol["capacity"][b] - g["sum_capacity"]) <= 1e-8
            # interference alignment property (reference tests/ia_package_test.py:977-1034)
            for k in range(3):
                for j in range(3):
                    Hkj = g["big_H"][2 * k:2 * k + 2, 2 * j:2 * j + 2]
                    val = sol["U"][b, k] @ Hkj @ sol["F"][b, j]
                    assert abs(val - (1.0 if j == k else 0.0)) < 1e-8
            # staged chain with injected data and noise -> identical decisions
            engine.set_constellation(g["table"], _lib.CONST_QAM)
            sym = engine.modulate(g["idx"].reshape(-1)).reshape(3, -1)
            X = np.vstack([np.outer(sol["F"][b, k], sym[k]) for k in range(3)])
            Y = engine.mimo_channel(g["big_H"][None], X[None], g["noise"][None], nv)[0]
            est = np.vstack([sol["U"][b, k] @ Y[2 * k:2 * k + 2] for k in range(3)])
            assert relerr(est, g["est"]) <= 1e-8
            assert np.array_equal(engine.demodulate(est), g["decisions"])


@pytest.mark.parametrize("dt,exact", [("f64", True), ("f32", False)])
def test_ia_pipeline(engine, dt, exact):
    kw = dict(mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=200, snr_db=20.0)
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    first, count = 9, 40
    want = [chains.chain_ia(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want])
    want_be = np.array([w["bit_errors"] for w in want])
    want_cap = np.array([w["sum_capacity"] for w in want])
    res, se, be, cap, _ = engine.run_ia(200, 1.0 / omodem.dB2Linear(20.0), SEED, first, count, dtype=dt,
                                        per_realization=True)
    check(res, se, be, want_se, want_be, 600, 2400, exact)
    assert np.max(np.abs(cap - want_cap)) <= 1e-7 and abs(res["sum_capacity"] - want_cap.sum()) <= 1e-6
    # shard invariance at the BASELINE size (1e5 realizations of 600 symbols)
    a = engine.run_ia(200, 0.01, SEED, 0, 100000, dtype="f32", method=_lib.DEMOD_QAM_SLICER)
    b = engine.run_ia(200, 0.01, SEED, 0, 33333, dtype="f32", method=_lib.DEMOD_QAM_SLICER)
    c = engine.run_ia(200, 0.01, SEED, 33333, 66667, dtype="f32", method=_lib.DEMOD_QAM_SLICER)
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations", "n_skipped"):
        assert a[k] == b[k] + c[k]
    assert 0.005 < a["sym_errors"] / (a["n_realizations"] * 600) < 0.08        # 0.023
    assert engine.run_ia(200, 0.0, SEED, 0, 2000, dtype="f32")["sym_errors"] == 0


# ---- iterative interference alignment -------------------------------------------------------
def test_ia_iterative_injected(engine):
    """AlternatingMin / MinLeakage / MaxSinr on the reference's own channels and random initial precoders:
    iteration counts, precoders, filters, SINR, capacity and the decisions of apps/ia/simulate_ia.py."""
    from helpers import golden_cases, relerr
    seen = set()
    for kw, reals in golden_cases("f3_ia_iterative"):
        seen.add(kw["algo"])
        H = np.stack([g["big_H"] for g in reals])
        F0 = np.stack([g["F_init"] for g in reals])
        nv = float(reals[0]["noise_var"])
        init = kw.get("initialize_with", "random")
        sol = engine.ia_iterative(kw["algo"], H, F0, nv, kw["max_iterations"], kw["relative_factor"],
                                  "fix" if init == "random" else init)
        assert not sol["skipped"].any()
        for b, g in enumerate(reals):
            assert int(sol["iterations"][b]) == int(g["runned_iterations"]), (kw["algo"], b)
            assert relerr(sol["F"][b], g["F"]) <= 1e-7 and relerr(sol["U"][b], g["U"]) <= 1e-7
            assert relerr(sol["sinr"][b], g["sinr"]) <= 1e-6 and abs(sol["capacity"][b] - g["sum_capacity"]) <= 1e-6
            engine.set_constellation(g["table"], _lib.CONST_QAM)
            sym = engine.modulate(g["idx"].reshape(-1)).reshape(3, -1)
            X = np.vstack([np.outer(sol["F"][b, k], sym[k]) for k in range(3)])
            Y = engine.mimo_channel(g["big_H"][None], X[None], g["noise"][None], nv)[0]
            est = np.vstack([sol["U"][b, k] @ Y[2 * k:2 * k + 2] for k in range(3)])
            assert relerr(est, g["est"]) <= 1e-6
            assert np.array_equal(engine.demodulate(est), g["decisions"])
    assert seen == {"alt_min", "min_leakage", "max_sinr", "mmse"}
    with pytest.raises(ValueError):
        engine.ia_iterative("alt_min", H, F0, nv, initialize_with="alt_min")   # algorithms.py:928-935
    with pytest.raises(ValueError):
        engine.ia_iterative("alt_min", H, F0[:1], nv)
    with pytest.raises(ValueError):
        engine.ia_iterative("alt_min", H, F0, nv, max_iterations=0)


def test_ia_iterative_svd_initialisation(engine):
    """initialize_with='svd' (algorithms.py:503-547): the start is each direct channel's dominant right singular
    vector -- unique up to a phase, LAPACK's in the reference.  Everything that does not depend on that phase
    equals the reference run (tests/golden/f3b_ia_svd_init.npz): iteration counts, SINRs, capacity, the precoders
    and filters up to one phase per user, and the noiseless link U_k H_kk F_k = 1; error statistics equal the
    oracle chain's."""
    from helpers import golden_cases, relerr
    from pyphysim_amd import ia, multiuser
    classes = {"alt_min": ia.AlternatingMinIASolver, "min_leakage": ia.MinLeakageIASolver,
               "max_sinr": ia.MaxSinrIASolver, "mmse": ia.MMSEIASolver}
    for kw, reals in golden_cases("f3b_ia_svd_init"):
        H = np.stack([g["big_H"] for g in reals])
        nv = float(reals[0]["noise_var"])
        sol = engine.ia_iterative(kw["algo"], H, np.zeros((len(reals), 3, 2), dtype=complex), nv, kw["max_iterations"],
                                  kw["relative_factor"], "svd")
        for b, g in enumerate(reals):
            assert int(sol["iterations"][b]) == int(g["runned_iterations"]), kw
            assert relerr(sol["sinr"][b], g["sinr"]) <= 1e-6 and abs(sol["capacity"][b] - g["sum_capacity"]) <= 1e-6
            for k in range(3):
                f, fr = sol["F"][b, k], g["F"][k]
                ph = np.vdot(fr, f) / abs(np.vdot(fr, f))               # f = fr * ph
                assert relerr(f, fr * ph) <= 1e-6, kw
                assert relerr(sol["U"][b, k], g["U"][k] * np.conj(ph)) <= 1e-6
                Hkk = g["big_H"][2 * k:2 * k + 2, 2 * k:2 * k + 2]
                assert abs(sol["U"][b, k] @ Hkk @ f - 1.0) <= 1e-9
            # the mirror class takes the same route
            muc = multiuser.MultiUserChannelMatrix(engine=engine)
            muc.init_from_channel_matrix(g["big_H"], 2, 2, 3)
            muc.noise_var = nv
            solver = classes[kw["algo"]](muc)
            solver.max_iterations, solver.relative_factor = kw["max_iterations"], kw["relative_factor"]
            solver.initialize_with = "svd"
            assert solver.solve(1) == int(g["runned_iterations"])
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    kw = dict(algo="max_sinr", mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=100, snr_db=14.0, max_iterations=30,
              initialize_with="svd")
    want = [chains.chain_ia_iterative(chains.PhiloxRng(SEED, r), **kw) for r in range(400)]
    res, se, be, cap, its = engine.run_ia(100, 1.0 / omodem.dB2Linear(14.0), SEED, 0, 400, dtype="f64",
                                          per_realization=True, solver="max_sinr", max_iterations=30,
                                          initialize_with="svd")
    assert np.array_equal(its, [w["runned_iterations"] for w in want])
    assert np.allclose(cap, [w["sum_capacity"] for w in want], rtol=0, atol=1e-6)
    ser, ser_ref = se.sum() / (400.0 * 300), np.sum([w["symbol_errors"] for w in want]) / (400.0 * 300)
    assert abs(ser - ser_ref) <= 0.25 * ser_ref + 2e-3


@pytest.mark.parametrize("algo,init", [("alt_min", "random"), ("min_leakage", "random"), ("max_sinr", "random"),
                                       ("max_sinr", "alt_min"), ("min_leakage", "closed_form"),
                                       ("max_sinr", "closed_form"), ("alt_min", "closed_form"), ("mmse", "random"),
                                       ("mmse", "alt_min"), ("mmse", "closed_form")])
@pytest.mark.parametrize("dt,exact", [("f64", True), ("f32", False)])
def test_ia_iterative_pipeline(engine, algo, init, dt, exact):
    kw = dict(algo=algo, mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=100, snr_db=18.0, max_iterations=40,
              initialize_with=init)
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    first, count = 3, 24
    want = [chains.chain_ia_iterative(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want])
    want_be = np.array([w["bit_errors"] for w in want])
    want_cap = np.array([w["sum_capacity"] for w in want])
    want_it = np.array([w["runned_iterations"] for w in want])
    res, se, be, cap, its = engine.run_ia(100, 1.0 / omodem.dB2Linear(18.0), SEED, first, count, dtype=dt,
                                          per_realization=True, solver=algo, max_iterations=40, initialize_with=init)
    check(res, se, be, want_se, want_be, 300, 1200, exact)
    assert np.array_equal(its, want_it) and res["ia_runned_iterations"] == int(want_it.sum())
    assert np.max(np.abs(cap - want_cap)) <= 1e-6
    # shard invariance and sanity at scale
    a = engine.run_ia(100, 0.01, SEED, 0, 20000, dtype="f32", solver=algo, max_iterations=20, initialize_with=init)
    b = engine.run_ia(100, 0.01, SEED, 0, 7777, dtype="f32", solver=algo, max_iterations=20, initialize_with=init)
    c = engine.run_ia(100, 0.01, SEED, 7777, 12223, dtype="f32", solver=algo, max_iterations=20, initialize_with=init)
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "n_realizations", "ia_runned_iterations"):
        assert a[k] == b[k] + c[k]
    assert a["sym_errors"] / (a["n_realizations"] * 300) < 0.15


def test_edge_cases_and_large_indices(engine):
    """64-bit realization indices and seeds, single realizations, odd lengths, big batches."""
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    big_first, big_seed = (1 << 40) + 12345, 0xFEDCBA9876543210
    want = chains.chain_awgn(chains.PhiloxRng(big_seed, big_first), "qam", 16, 1001, 9.0)
    res, se, be = engine.run_awgn(1001, 1.0 / omodem.dB2Linear(9.0), big_seed, big_first, 1, dtype="f64",
                                  per_realization=True)
    assert int(se[0]) == want["symbol_errors"] and int(be[0]) == want["bit_errors"] and res["n_realizations"] == 1
    # one symbol per realization, many realizations (more work items than the persistent grid)
    res = engine.run_awgn(1, 0.5, SEED, 0, 300000, dtype="f32")
    assert res["n_realizations"] == 300000 and 0 < res["sym_errors"] < 300000 and res["n_symbols"] == 1
    assert res["sym_errors_sq"] == res["sym_errors"]                  # e_r in {0, 1}
    # MIMO-OFDM: a single realization and a count that is not a multiple of the grid size
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    a = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, 0.01, SEED, 7, 1, dtype="f32")
    b = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, 0.01, SEED, 0, 769, dtype="f32", per_realization=True)
    assert a["n_realizations"] == 1 and a["sym_errors"] == int(b[1][7]) and b[0]["n_realizations"] == 769
    assert b[0]["sym_errors"] == int(b[1].astype(np.int64).sum())
    # a rank-deficient "channel" cannot occur with Gaussian draws; zero forcing at huge SNR must not skip
    z = engine.run_mimo_ofdm(2, 2, 64, 8, 48, 2, 0.0, SEED, 0, 5000, mmse=False, dtype="f64")
    assert z["n_skipped"] == 0 and z["sym_errors"] == 0


# ---- the reference's MIMO application: six schemes over a flat channel -----------------------------------------
FLAT_EXACT = [("blast", 2, 2, 18.0), ("blast", 3, 4, 14.0), ("blast", 4, 4, 22.0), ("blast", 1, 1, 12.0),
              ("mrc", 1, 3, 6.0), ("mrt", 3, 1, 8.0), ("mrt", 4, 1, 6.0), ("alamouti", 2, 1, 10.0),
              ("alamouti", 2, 3, 4.0), ("alamouti", 2, 4, 2.0)]


@pytest.mark.parametrize("dt,exact", [("f64", True), ("f32", False)])
@pytest.mark.parametrize("scheme,nt,nr,snr", FLAT_EXACT)
def test_mimo_flat_pipeline(engine, dt, exact, scheme, nt, nr, snr):
    kw = dict(scheme=scheme, mod="qam", M=16, nt=nt, nr=nr, NSymbs=128, snr_db=snr)
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    first, count = 3, 70                      # more than one 64-realization chunk
    want_se, want_be, nsym, nbits = oracle_counts(chains.chain_mimo_scheme, first, count, **kw)
    res, se, be = engine.run_mimo_flat(scheme, nt, nr, 128, 1.0 / omodem.dB2Linear(snr), SEED, first, count, dtype=dt,
                                       per_realization=True)
    check(res, se, be, want_se, want_be, nsym, nbits, exact)


@pytest.mark.parametrize("NS", [1, 2, 51, 126, 130, 257, 258])
def test_chunked_pipelines_ragged_symbol_counts(engine, NS):
    """The one-wave pipelines take two columns per lane and pass when the symbol count is even (whole Philox
    blocks, wave-shared data blocks: csrc/wave_draws.hpp) and one column per lane otherwise; both paths, with
    partly filled passes and at every alignment of the rows in the streams, reproduce the oracle's counts."""
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    first, count = 61, 9
    kw = dict(mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=NS, snr_db=12.0)
    want = [chains.chain_ia(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + count)]
    res, se, be, _, _ = engine.run_ia(NS, 1.0 / omodem.dB2Linear(12.0), SEED, first, count, dtype="f64",
                                      per_realization=True)
    assert np.array_equal(se, [w["symbol_errors"] for w in want]) and np.array_equal(be, [w["bit_errors"] for w in want])
    for scheme, nt, nr in (("blast", 3, 4), ("blast", 2, 2), ("mrt", 4, 1), ("mrc", 1, 2)) + (
            (("alamouti", 2, 2),) if NS % 2 == 0 else ()):
        kw = dict(scheme=scheme, mod="qam", M=16, nt=nt, nr=nr, NSymbs=NS, snr_db=10.0)
        want_se, want_be, nsym, nbits = oracle_counts(chains.chain_mimo_scheme, first, count, **kw)
        res, se, be = engine.run_mimo_flat(scheme, nt, nr, NS, 1.0 / omodem.dB2Linear(10.0), SEED, first, count,
                                           dtype="f64", per_realization=True)
        check(res, se, be, want_se, want_be, nsym, nbits, True)


@pytest.mark.parametrize("scheme,n", [("svd", 2), ("svd", 4), ("gmd", 2), ("gmd", 3), ("gmd", 4)])
def test_mimo_flat_svd_gmd(engine, scheme, n):
    """Singular-vector phases are implementation-defined (LAPACK vs Jacobi), so per-realization decisions are
    compared with the staged operator classes (the same device SVD / GMD, validated against the reference on
    injected data) driven by the same Philox draws; statistics are compared with the oracle chain."""
    from oracle import philox as P
    from pyphysim_amd import mimo as mmimo
    snr, NS, count = 16.0, 64, 12
    nv = 1.0 / omodem.dB2Linear(snr)
    table = chains.constellation("qam", 16)
    engine.set_constellation(table, _lib.CONST_QAM)
    res, se, be = engine.run_mimo_flat(scheme, n, n, NS, nv, SEED, 0, count, dtype="f64", per_realization=True)
    cls = mmimo.SVDMimo if scheme == "svd" else mmimo.GMDMimo
    for r in range(count):
        H = P.cnormal(SEED, r, n * n, P.STREAM_CHAN).reshape(n, n)
        idx = P.symbols(SEED, r, n * NS, 16)
        noise = P.cnormal(SEED, r, n * NS, P.STREAM_NOISE).reshape(n, NS)
        obj = cls(H, engine=engine, dtype="f64")
        X = obj.encode(table[idx])
        est = obj.decode(H @ X + np.sqrt(nv) * noise)
        dec = np.argmin(np.abs(est[:, None] - table[None, :]), axis=1)
        assert int(se[r]) == int(np.count_nonzero(dec != idx)), (scheme, n, r)
    # statistics against the oracle chain (numpy.linalg.svd inside): same SER within Monte Carlo noise
    big = engine.run_mimo_flat(scheme, n, n, NS, nv, SEED, 0, 20000, dtype="f32", method=_lib.DEMOD_QAM_SLICER)
    want = [chains.chain_mimo_scheme(chains.PhiloxRng(SEED, r), scheme, "qam", 16, n, n, NS, snr)["symbol_errors"]
            for r in range(400)]
    ser_gpu = big["sym_errors"] / (20000.0 * n * NS)
    ser_ref = np.sum(want) / (400.0 * n * NS)
    assert abs(ser_gpu - ser_ref) < 0.25 * ser_ref + 2e-3
    assert engine.run_mimo_flat(scheme, n, n, NS, 0.0, SEED, 0, 500, dtype="f32")["sym_errors"] == 0


@pytest.mark.parametrize("case", [8, 9, 10, 11, 12])
def test_svd_gmd_with_the_references_basis(engine, case):
    """VERDICT r02 (parity soft spot a13): SVDMimo / GMDMimo decisions depend on the phase LAPACK gave every singular vector.
    With the reference's OWN precoder and receive filter injected (minted into f5_mimo_schemes.npz from
    SVDMimo._calc_precoder / _calc_receive_filter, mimo.py:846-890, 965-1011) the device encode -> channel -> decode chain
    reproduces the reference's estimates to rounding and its DECISIONS and error counts exactly; the device's own
    decomposition gives the same singular values / a consistent pair (G H W = I) but other phases."""
    from pyphysim_amd import mimo as mmimo
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f5_mimo_schemes.npz"), allow_pickle=True)
    kw = json.loads(str(g["case%d_kwargs" % case]))
    cls = mmimo.SVDMimo if kw["scheme"] == "svd" else mmimo.GMDMimo
    for r in range(2):
        pre = "case%d_r%d_" % (case, r)
        table, H, idx, noise = g[pre + "table"], g[pre + "H"], g[pre + "idx"], g[pre + "noise"]
        nv = float(g[pre + "noise_var"])
        engine.set_constellation(table, _lib.CONST_QAM)
        obj = cls(H, engine=engine, dtype="f64")
        obj.set_filters(g[pre + "W"], g[pre + "G_H"])
        X = obj.encode(table[idx])
        est = obj.decode(H @ X + np.sqrt(nv) * noise)
        assert np.max(np.abs(est - g[pre + "est"])) <= 1e-10
        dec = engine.demodulate(est, dtype="f64")
        assert np.array_equal(dec, g[pre + "decisions"])
        assert int(np.count_nonzero(dec != idx)) == int(g[pre + "symbol_errors"])
        # the device decomposition: same link up to the per-stream phase -- an identity end to end without noise
        obj.set_filters(None, None)
        back = obj.decode(H @ obj.encode(table[idx]))
        assert np.max(np.abs(back - table[idx])) <= 1e-9


def test_mimo_flat_errors_and_shards(engine):
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    for bad in (("blast", 4, 2), ("mrc", 2, 2), ("mrt", 2, 2), ("alamouti", 3, 2), ("svd", 2, 3), ("gmd", 1, 1),
                ("blast", 5, 5)):
        with pytest.raises(_lib.McleError):
            engine.run_mimo_flat(bad[0], bad[1], bad[2], 64, 0.1, SEED, 0, 4)
    with pytest.raises(_lib.McleError):
        engine.run_mimo_flat("alamouti", 2, 2, 63, 0.1, SEED, 0, 4)
    with pytest.raises(KeyError):
        engine.run_mimo_flat("vblast", 2, 2, 64, 0.1, SEED, 0, 4)
    a = engine.run_mimo_flat("blast", 4, 4, 200, 0.01, SEED, 0, 100000, dtype="f32", method=_lib.DEMOD_QAM_SLICER)
    b = engine.run_mimo_flat("blast", 4, 4, 200, 0.01, SEED, 0, 33333, dtype="f32", method=_lib.DEMOD_QAM_SLICER)
    c = engine.run_mimo_flat("blast", 4, 4, 200, 0.01, SEED, 33333, 66667, dtype="f32", method=_lib.DEMOD_QAM_SLICER)
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations", "n_skipped"):
        assert a[k] == b[k] + c[k]
    # MMSE beats zero forcing at low SNR
    zf = engine.run_mimo_flat("blast", 4, 4, 100, 1.0, SEED, 0, 20000, dtype="f32")
    mm = engine.run_mimo_flat("blast", 4, 4, 100, 1.0, SEED, 0, 20000, dtype="f32", mmse=True)
    assert mm["sym_errors"] < zf["sym_errors"]
