"""GPU: the frequency-selective MIMO-OFDM link with one receive antenna per WAVEFRONT
(csrc/mimo_tdl_wave.hpp: k_run_mimo_ofdm_tdl_wave -- the default of mcle_run_mimo_ofdm_tdl since round 5, since round 6 also with
tap delays beyond the cyclic prefix; option mimo_tdl_kernel = 1 selects the workgroup-cooperative kernel of rounds 1-4, = 2 the
run-time-order form of the wavefront kernel) against the oracle chain (oracle/chains.py::chain_mimo_ofdm_tdl, pinned to the
reference by tests/golden/f1_mimo_ofdm_tdl.npz) under the same Philox keying, and against the kernel it replaces.
complex128: per-realization symbol AND bit counts exact; complex64: |dSER| <= 1e-4, boundary ties only.
Reference: channels/fading.py:1092-1118, :513-536; modulators/ofdm.py:545-547; mimo/mimo.py:287-309, :577-607."""
import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 27182818
TS15 = 1.0 / (15e3 * 1024)


def _run(engine, first, count, dtype, kernel=0, **kw):
    from pyphysim_amd.channels import discretize_profile
    with engine.options(mimo_tdl_kernel=kernel):
        Ts = kw.get("Ts", TS15)
        p_lin, d_idx = discretize_profile(np.asarray(kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)), dtype=float),
                                          np.asarray(kw.get("tap_delays_samples", (0, 1, 2, 3, 4)), dtype=float) * Ts, Ts)
        nv = 1.0 / omodem.dB2Linear(kw.get("snr_db", 25.0))
        fft = kw.get("fft_size", 1024)
        return engine.run_mimo_ofdm_tdl(kw.get("nt", 4), kw.get("nr", 4), fft, kw.get("cp_size", 16), kw.get("num_used") or fft,
                                        kw.get("n_ofdm_sym", 1), nv, p_lin, d_idx, SEED, first, count, Fd=kw.get("Fd", 10.0), Ts=Ts,
                                        L=kw.get("L", 8), mmse=kw.get("mmse", True), method=kw.get("method", _lib.DEMOD_MINDIST),
                                        dtype=dtype, per_realization=True)


def _oracle(first, count, mod, M, linear_mean=True, **kw):
    okw = dict(mod=mod, M=M, nt=kw.get("nt", 4), nr=kw.get("nr", 4), fft_size=kw.get("fft_size", 1024), cp_size=kw.get("cp_size", 16),
               num_used=kw.get("num_used"), n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw.get("snr_db", 25.0), Fd=kw.get("Fd", 10.0),
               Ts=kw.get("Ts", TS15), L=kw.get("L", 8), tap_powers_dB=kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)),
               tap_delays_samples=kw.get("tap_delays_samples", (0, 1, 2, 3, 4)), mmse=kw.get("mmse", True), linear_mean=linear_mean)
    out = [chains.chain_mimo_ofdm_tdl(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    return (np.array([o["symbol_errors"] for o in out], dtype=np.int64), np.array([o["bit_errors"] for o in out], dtype=np.int64),
            out[0]["num_symbols"], out[0]["num_bits"])


def _set(engine, mod, M):
    engine.set_constellation(chains.constellation(mod, M), _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)


# (count, case): the benchmark geometry first; then what the envelope adds -- rectangular channels, every size, partial bands, odd
# prefixes (unpaired noise draws), several symbols, more taps, other polynomial orders, ZF, constellations without a certificate
CASES = [
    (6, dict(mod="qam", M=64)),                                                                     # bench.py --config f1
    (6, dict(mod="qam", M=64, method=_lib.DEMOD_QAM_SLICER, n_ofdm_sym=2)),
    (6, dict(mod="qam", M=16, nt=2, nr=4, snr_db=16.0)),
    (6, dict(mod="qam", M=16, nt=3, nr=4, snr_db=20.0, cp_size=9, tap_delays_samples=(0, 2, 5, 9),  # odd prefix: rows 1, 3 unpaired
             tap_powers_dB=(0.0, -2.0, -5.0, -8.0), Fd=200.0)),
    (6, dict(mod="qam", M=64, nt=1, nr=4, snr_db=14.0, num_used=600)),
    (6, dict(mod="qpsk", M=4, nt=3, nr=3, snr_db=12.0, n_ofdm_sym=2, num_used=1000, cp_size=33)),   # odd symbol length
    (8, dict(mod="qam", M=16, nt=2, nr=3, snr_db=15.0, mmse=False)),                                # zero forcing
    (8, dict(mod="qam", M=16, nt=2, nr=2, snr_db=18.0, L=12)),
    (8, dict(mod="psk", M=8, nt=1, nr=2, snr_db=12.0, cp_size=32, tap_delays_samples=(0, 7, 17, 31),  # no certificate: grid in LDS
             tap_powers_dB=(0.0, -1.0, -3.0, -6.0), Fd=900.0, n_ofdm_sym=3)),
    (8, dict(mod="qam", M=16, nt=1, nr=1, snr_db=22.0)),
    (6, dict(mod="qam", M=16, snr_db=24.0, cp_size=208, n_ofdm_sym=2, Fd=50.0,                      # the envelope's corner: eight taps
             tap_delays_samples=(0, 1, 7, 33, 64, 65, 130, 200), tap_powers_dB=(0.0, -1.0, -2.0, -3.0, -4.0, -5.0, -6.0, -7.0))),
    (12, dict(mod="qam", M=64, fft_size=256, snr_db=24.0, num_used=200, Ts=1e-6, Fd=50.0)),
    (12, dict(mod="qam", M=16, fft_size=256, nt=2, nr=3, snr_db=14.0, Ts=1e-6, Fd=70.0, cp_size=7, tap_delays_samples=(0, 3, 6),
              tap_powers_dB=(0.0, -4.0, -9.0), n_ofdm_sym=3)),
    (12, dict(mod="qam", M=16, fft_size=256, nt=1, nr=1, snr_db=20.0, Ts=1e-6)),
    (10, dict(mod="qam", M=64, fft_size=512, snr_db=24.0, num_used=300, cp_size=36, Ts=1e-6, Fd=50.0)),
    (10, dict(mod="qam", M=16, fft_size=512, nt=2, nr=2, snr_db=15.0, Ts=1e-6, n_ofdm_sym=2)),
    (10, dict(mod="qpsk", M=4, fft_size=512, nt=1, nr=3, snr_db=4.0, Ts=1e-6, cp_size=5)),
    (5, dict(mod="qam", M=16, fft_size=2048, nt=2, nr=2, snr_db=14.0, num_used=1200, cp_size=144, Fd=5.0)),
    (5, dict(mod="qam", M=64, fft_size=2048, snr_db=25.0, cp_size=17)),
    (5, dict(mod="qam", M=16, fft_size=2048, nt=2, nr=4, snr_db=12.0, n_ofdm_sym=2)),
    # the decode's delay classes (complex64: host-sorted class positions, mimo_tdl.hpp cls_code): one class empty, one class full
    (6, dict(mod="qam", M=16, snr_db=22.0, tap_delays_samples=(0, 2, 4, 6, 8), tap_powers_dB=(0.0, -2.0, -4.0, -6.0, -8.0))),
    (6, dict(mod="qam", M=16, nt=2, nr=4, snr_db=18.0, tap_delays_samples=(1, 3, 5, 7, 9, 11), cp_size=12,
             tap_powers_dB=(0.0, -1.0, -2.0, -3.0, -4.0, -5.0))),
    (6, dict(mod="qam", M=64, snr_db=26.0, cp_size=20, tap_delays_samples=(1, 2, 4, 6, 8, 10, 12, 14),   # seven even, one odd
             tap_powers_dB=(0.0, -1.0, -2.0, -3.0, -4.0, -5.0, -6.0, -7.0))),
]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_wave_kernel_against_the_oracle(engine, case, dtype):
    count, kw = CASES[case]
    kw = dict(kw)
    mod, M = kw.pop("mod"), kw.pop("M")
    _set(engine, mod, M)
    first = (1 << 33) + 4099
    want_se, want_be, nsym, nbits = _oracle(first, count, mod, M, **kw)
    res, se, be = _run(engine, first, count, dtype, **kw)
    assert res["n_symbols"] == nsym and res["n_bits"] == nbits and res["n_realizations"] == count and res["n_skipped"] == 0
    square = kw.get("nt", 4) == kw.get("nr", 4) and kw.get("nt", 4) in (2, 4)
    if square:                                                # the kernel of rounds 1-4 covers Nt = Nr in {2, 4}
        _, se_o, be_o = _run(engine, first, count, dtype, kernel=1, **kw)
    _, se_r, be_r = _run(engine, first, count, dtype, kernel=2, **kw)        # run-time polynomial order
    if dtype == "f64":
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (case, se, want_se)
        assert np.array_equal(se_r, want_se) and np.array_equal(be_r, want_be)
        if square:
            assert np.array_equal(se, se_o) and np.array_equal(be, be_o)
    else:
        assert abs(int(se.sum()) - int(want_se.sum())) <= 1e-4 * count * nsym + 3
        assert abs(int(be.sum()) - int(want_be.sum())) <= 1e-4 * count * nbits + 4
        assert np.max(np.abs(se.astype(np.int64) - want_se)) <= 3                 # boundary ties only
        assert np.max(np.abs(se.astype(np.int64) - se_r.astype(np.int64))) <= 3
        if square:
            assert np.max(np.abs(se.astype(np.int64) - se_o.astype(np.int64))) <= 3
    assert res["sym_errors"] == int(se.astype(np.int64).sum()) and res["sym_errors_sq"] == int((se.astype(np.int64) ** 2).sum())
    assert res["bit_errors"] == int(be.astype(np.int64).sum()) and res["bit_errors_sq"] == int((be.astype(np.int64) ** 2).sum())
    # bit-identical from run to run and under any split of the realization range
    a = _run(engine, first, 2, dtype, **kw)
    b = _run(engine, first + 2, count - 2, dtype, **kw)
    assert np.array_equal(np.concatenate([a[1], b[1]]), se) and np.array_equal(np.concatenate([a[2], b[2]]), be)


def test_the_literal_and_the_linear_mean_response_give_the_same_counts():
    """The deep tests below run the oracle with the DFT of the mean taps (linearity); here it is held to the literal mean of
    per-sample DFTs the reference computes (channels/fading.py:513-536, modulators/ofdm.py:545-547) on the benchmark geometry."""
    a = _oracle(11, 2, "qam", 64, linear_mean=False)
    b = _oracle(11, 2, "qam", 64, linear_mean=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_benchmark_geometry_against_the_oracle_over_240_realizations(engine, dtype):
    """depth against the ORACLE, not against another kernel -- 240 realizations of bench.py's f1 workload
    (4 x 4, 64-QAM, OFDM(1024, 16), five taps, 25 dB), every per-realization symbol and bit count in complex128."""
    _set(engine, "qam", 64)
    first, count = 5000, 240
    want_se, want_be, nsym, nbits = _oracle(first, count, "qam", 64)
    res, se, be = _run(engine, first, count, dtype)
    if dtype == "f64":
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be)
    else:
        assert abs(int(se.sum()) - int(want_se.sum())) <= 1e-5 * count * nsym + 3
        assert np.max(np.abs(se.astype(np.int64) - want_se)) <= 3
    assert res["n_realizations"] == count and res["sym_errors"] == int(se.astype(np.int64).sum())


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", [(1024, 2, 4), (256, 4, 4), (2048, 2, 2), (512, 3, 3)])
def test_rectangular_and_other_sizes_against_the_oracle_over_120_realizations(engine, dtype, shape):
    fft, nt, nr = shape
    _set(engine, "qam", 16)
    kw = dict(fft_size=fft, nt=nt, nr=nr, snr_db=14.0 + 3.0 * (nt == nr), Fd=30.0)
    first, count = 900, 120 if fft <= 1024 else 60
    want_se, want_be, nsym, nbits = _oracle(first, count, "qam", 16, **kw)
    res, se, be = _run(engine, first, count, dtype, **kw)
    if dtype == "f64":
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be)
    else:
        assert abs(int(se.sum()) - int(want_se.sum())) <= 1e-5 * count * nsym + 3
        assert np.max(np.abs(se.astype(np.int64) - want_se)) <= 3


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_wave_kernel_equals_the_cooperative_kernel_over_3000_realizations(engine, dtype):
    """Kernel against kernel (not parity): 3 001 realizations, more than one grid-stride pass of every workgroup."""
    _set(engine, "qam", 64)
    count = 3001
    res, se, be = _run(engine, 123456, count, dtype)
    old, se_o, be_o = _run(engine, 123456, count, dtype, kernel=1)
    if dtype == "f64":
        assert np.array_equal(se, se_o) and np.array_equal(be, be_o)
        assert res == old
    else:
        assert np.max(np.abs(se.astype(np.int64) - se_o.astype(np.int64))) <= 4
        assert abs(res["sym_errors"] - old["sym_errors"]) <= 1e-5 * count * 4096 + 3


# A tap BEYOND the cyclic prefix (inter-symbol interference: the previous symbol's end reaches into this one; zeros in front of the
# first symbol, channels/fading.py:1092-1118).  Round 6: inside the wavefront kernels' envelope -- every geometry, not only the
# square ones the cooperative kernel takes.  Delays on both sides of the prefix, a prefix of zero, a delay that is a multiple of 16
# and one that is not (the history is P - cp samples, P = the largest delay rounded up to 16), histories longer than 64 samples
# (more than one sample per lane), every transform form (radix-4 at 256 / 512 / 2048, radix-16 passes at 1024).
ISI_CASES = [
    (6, dict(mod="qam", M=16, fft_size=256, nt=2, nr=2, snr_db=16.0, cp_size=4, tap_delays_samples=(0, 7, 19),
             tap_powers_dB=(0.0, -4.0, -9.0), n_ofdm_sym=3, Ts=1e-6, Fd=50.0)),
    (6, dict(mod="qam", M=16, fft_size=256, nt=2, nr=3, snr_db=16.0, cp_size=4, tap_delays_samples=(0, 7, 19),
             tap_powers_dB=(0.0, -4.0, -9.0), n_ofdm_sym=3, Ts=1e-6, Fd=50.0)),
    (6, dict(mod="qam", M=64, fft_size=256, nt=1, nr=4, snr_db=24.0, cp_size=0, tap_delays_samples=(0, 1, 16),
             tap_powers_dB=(0.0, -6.0, -12.0), n_ofdm_sym=4, Ts=1e-6, num_used=200)),
    (5, dict(mod="qam", M=16, fft_size=512, nt=3, nr=4, snr_db=18.0, cp_size=9, tap_delays_samples=(0, 9, 10, 100, 131),
             tap_powers_dB=(0.0, -2.0, -4.0, -6.0, -8.0), n_ofdm_sym=3, Ts=1e-6)),
    (4, dict(mod="qam", M=64, snr_db=25.0, cp_size=16, tap_delays_samples=(0, 1, 2, 17, 40), n_ofdm_sym=3)),
    (4, dict(mod="qam", M=16, nt=2, nr=4, snr_db=15.0, cp_size=33, tap_delays_samples=(0, 33, 34, 250), n_ofdm_sym=2,
             tap_powers_dB=(0.0, -3.0, -5.0, -7.0), method=_lib.DEMOD_QAM_SLICER)),
    (3, dict(mod="qpsk", M=4, fft_size=2048, nt=2, nr=2, snr_db=10.0, cp_size=17, tap_delays_samples=(0, 5, 90), n_ofdm_sym=2,
             tap_powers_dB=(0.0, -3.0, -6.0), Fd=5.0)),
    (3, dict(mod="qam", M=16, fft_size=2048, snr_db=20.0, cp_size=8, tap_delays_samples=(0, 3, 8, 21), n_ofdm_sym=2,
             tap_powers_dB=(0.0, -3.0, -6.0, -9.0))),
]


@pytest.mark.parametrize("case", range(len(ISI_CASES)))
def test_a_delay_beyond_the_prefix_against_the_oracle(engine, case):
    count, kw = ISI_CASES[case]
    kw = dict(kw)
    mod, M = kw.pop("mod"), kw.pop("M")
    _set(engine, mod, M)
    want_se, want_be, nsym, nbits = _oracle(40, count, mod, M, **kw)
    res, se, be = _run(engine, 40, count, "f64", **kw)
    assert res["n_symbols"] == nsym and res["n_bits"] == nbits
    assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (se, want_se)
    assert want_se.sum() > 0
    _, se32, _ = _run(engine, 40, count, "f32", **kw)
    assert np.max(np.abs(se32.astype(np.int64) - want_se)) <= 3 + 2e-3 * nsym
    if kw.get("nt", 4) == kw.get("nr", 4) and kw.get("nt", 4) in (2, 4):
        _, se_o, be_o = _run(engine, 40, count, "f64", kernel=1, **kw)        # the cooperative kernel carries the tail too
        assert np.array_equal(se, se_o) and np.array_equal(be, be_o)
    # several passes of a workgroup's realization loop (the history is reset at every first symbol), run-time-order kernels
    n = 2500 if kw.get("fft_size", 1024) <= 512 else 700
    a = _run(engine, 7, n, "f64", **kw)
    b = _run(engine, 7, n, "f64", kernel=2, **kw)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[1][33:33 + count], se) and np.array_equal(a[2][33:33 + count], be)


def test_interference_changes_the_counts_and_other_envelope_errors(engine):
    """The carried tail matters (the same link with the prefix long enough has fewer errors), and what stays outside the fused
    kernels says so: Nt > Nr has no Blast filter (mimo/mimo.py:264-309 needs full column rank); a delay beyond fft_size / 2 on a
    rectangular channel is a configuration for the staged operator chain."""
    _set(engine, "qam", 16)
    kw = dict(fft_size=256, nt=2, nr=3, snr_db=30.0, tap_delays_samples=(0, 7, 19), tap_powers_dB=(0.0, -4.0, -9.0),
              n_ofdm_sym=3, Ts=1e-6, Fd=50.0)
    short = _run(engine, 40, 400, "f64", cp_size=4, **kw)[0]
    long_ = _run(engine, 40, 400, "f64", cp_size=20, **kw)[0]
    assert short["sym_errors"] > 3 * long_["sym_errors"] + 100
    with pytest.raises(_lib.McleError):
        _run(engine, 40, 6, "f64", cp_size=4, **dict(kw, nt=3, nr=2))
    with pytest.raises(_lib.McleUnsupported):
        _run(engine, 40, 6, "f64", cp_size=4, **dict(kw, tap_delays_samples=(0, 7, 130)))


def test_noise_free_link_against_the_oracle(engine):
    """300 dB: what is left is the inter-carrier interference of a channel that moves inside the symbol (the equaliser sees the
    per-symbol MEAN response, channels/fading.py:513-536) -- a handful of 64-QAM decisions in 65 536, the same ones as the oracle's."""
    _set(engine, "qam", 64)
    for kw in (dict(), dict(nt=2, nr=3, fft_size=512, Ts=1e-6)):
        want_se, want_be, nsym, _ = _oracle(7, 16, "qam", 64, snr_db=300.0, **kw)
        assert want_se.sum() <= 1e-4 * 16 * nsym
        res, se, be = _run(engine, 7, 16, "f64", snr_db=300.0, **kw)
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be)
        res32, se32, _ = _run(engine, 7, 16, "f32", snr_db=300.0, **kw)
        # complex64: a near-singular H(f) (one subcarrier in ~1e5) costs a 64-QAM decision without any noise
        assert res32["sym_errors"] <= want_se.sum() + 1e-4 * 16 * nsym
