"""GPU: config 3 with one realization per WAVEFRONT (csrc/siso_tdl_wave.hpp: k_run_ofdm_tdl_wave -- radix-16 register
passes of fft_r16.hpp, no workgroup barrier in the loop, DPP exchange of the noise halves; the default since round 4, option tdl_kernel = 1 selects the
batched kernels) against the oracle
chain under the same Philox keying and against the batched kernels it stands beside.  complex128: per-realization counts
exact; complex64: |dSER| <= 1e-4, boundary ties only (the criteria of the batched kernels' tests)."""
import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 31415926


def _run(engine, first, count, dtype, wave=1, fft=1024, **kw):
    from pyphysim_amd.channels import discretize_profile
    with engine.options(tdl_kernel=2 if wave else 1):            # 2: the wavefront kernel wherever it exists
        Ts = kw.get("Ts", 1.0 / (15e3 * 1024))
        p_lin, d_idx = discretize_profile(np.asarray(kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)), dtype=float),
                                          np.asarray(kw.get("tap_delays_samples", (0, 1, 2, 3, 4)), dtype=float) * Ts, Ts)
        nv = 1.0 / omodem.dB2Linear(kw.get("snr_db", 20.0))
        return engine.run_ofdm_tdl(fft, kw.get("cp_size", 16), kw.get("num_used") or fft, kw.get("n_ofdm_sym", 1), nv,
                                   p_lin, d_idx, SEED, first, count, Fd=kw.get("Fd", 10.0), Ts=Ts, L=kw.get("L", 8),
                                   method=kw.get("method", _lib.DEMOD_MINDIST), dtype=dtype, per_realization=True)


CASES = [dict(mod="qpsk", M=4, snr_db=20.0),                                            # BASELINE config 3
         dict(mod="qam", M=16, snr_db=24.0, num_used=608, n_ofdm_sym=2, L=12),          # partial band, 2 symbols
         dict(mod="qam", M=64, snr_db=30.0, cp_size=9, tap_delays_samples=(0, 2, 5, 9), # odd CP: unpaired noise draws
              tap_powers_dB=(0.0, -2.0, -5.0, -8.0), Fd=200.0),
         dict(mod="psk", M=8, snr_db=18.0, cp_size=32, tap_delays_samples=(0, 7, 17, 31),
              tap_powers_dB=(0.0, -1.0, -3.0, -6.0), Fd=900.0, n_ofdm_sym=3),           # higher polynomial order
         dict(mod="qam", M=64, snr_db=28.0, num_used=1000, n_ofdm_sym=3, cp_size=33,    # odd symbol length: the noise pairing
              tap_delays_samples=(0, 3), tap_powers_dB=(0.0, -4.0), method=_lib.DEMOD_QAM_SLICER),   # alternates per symbol
         dict(mod="qam", M=256, snr_db=34.0, cp_size=0, tap_delays_samples=(0,), tap_powers_dB=(0.0,)),
         # the envelope's corners: eight taps reaching 200 samples into a 208-sample prefix (the prefix copy spans four 64-sample
         # blocks), four symbols; a two-subcarrier band in PSK(2); a Doppler that takes polynomial order 8 in complex64 (complex128: 13 -> the
         # batched kernel serves it; the 8-tap case takes order 3 / 7)
         dict(mod="qam", M=16, snr_db=26.0, cp_size=208, n_ofdm_sym=4, L=8, Fd=50.0,
              tap_delays_samples=(0, 1, 7, 33, 64, 65, 130, 200), tap_powers_dB=(0.0, -1.0, -2.0, -3.0, -4.0, -5.0, -6.0, -7.0)),
         dict(mod="psk", M=2, snr_db=6.0, num_used=2, cp_size=16, n_ofdm_sym=2),
         dict(mod="qpsk", M=4, snr_db=16.0, cp_size=64, Fd=2000.0, tap_delays_samples=(0, 5, 40), tap_powers_dB=(0.0, -3.0, -6.0))]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_wave_kernel_against_the_oracle_and_the_batched_kernels(engine, case, dtype):
    kw = dict(CASES[case])
    mod, M = kw.pop("mod"), kw.pop("M")
    engine.set_constellation(chains.constellation(mod, M), _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)
    first, count = (1 << 34) + 70001, 23                  # not a multiple of the four wavefronts of a workgroup
    okw = dict(mod=mod, M=M, fft_size=1024, cp_size=kw.get("cp_size", 16), num_used=kw.get("num_used"),
               n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], Fd=kw.get("Fd", 10.0), L=kw.get("L", 8),
               tap_powers_dB=kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)),
               tap_delays_samples=kw.get("tap_delays_samples", (0, 1, 2, 3, 4)))
    want = [chains.chain_ofdm_tdl(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want], dtype=np.int64)
    want_be = np.array([w["bit_errors"] for w in want], dtype=np.int64)
    nsym, nbits = want[0]["num_symbols"], want[0]["num_bits"]
    res, se, be = _run(engine, first, count, dtype, **kw)
    assert res["n_symbols"] == nsym and res["n_bits"] == nbits and res["n_realizations"] == count
    old, se_o, be_o = _run(engine, first, count, dtype, wave=0, **kw)
    if dtype == "f64":
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (case, se, want_se)
        assert np.array_equal(se, se_o) and np.array_equal(be, be_o)
    else:
        assert abs(int(se.sum()) - int(want_se.sum())) <= 1e-4 * count * nsym + 2
        assert abs(int(be.sum()) - int(want_be.sum())) <= 1e-4 * count * nbits + 2
        assert np.max(np.abs(se.astype(np.int64) - want_se)) <= 3                 # boundary ties only
        assert np.max(np.abs(se.astype(np.int64) - se_o.astype(np.int64))) <= 3
    assert res["sym_errors"] == int(se.astype(np.int64).sum()) and res["sym_errors_sq"] == int((se.astype(np.int64) ** 2).sum())
    assert res["bit_errors"] == int(be.astype(np.int64).sum())
    # bit-identical from run to run and under any split of the realization range
    a = _run(engine, first, 9, dtype, **kw)
    b = _run(engine, first + 9, count - 9, dtype, **kw)
    assert np.array_equal(np.concatenate([a[1], b[1]]), se) and np.array_equal(np.concatenate([a[2], b[2]]), be)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_wave_kernel_large_sample_and_zero_noise(engine, dtype):
    """BASELINE config 3 over 30 000 realizations (more than one grid pass of wavefronts): counters equal to the batched kernel's
    to the tolerance (complex128: exactly), splits add up; without noise only the deep fades of the one-tap equaliser err."""
    engine.set_constellation(chains.constellation("qpsk", 4), _lib.CONST_GENERIC)
    n = 30011
    new = _run(engine, 0, n, dtype, snr_db=20.0)[0]
    old = _run(engine, 0, n, dtype, wave=0, snr_db=20.0)[0]
    assert new["n_realizations"] == old["n_realizations"] == n
    if dtype == "f64":
        assert new == old
    else:
        assert abs(new["sym_errors"] - old["sym_errors"]) <= 1e-5 * n * 1024 + 2
    a = _run(engine, 0, 12345, dtype, snr_db=20.0)[0]
    b = _run(engine, 12345, n - 12345, dtype, snr_db=20.0)[0]
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations"):
        assert new[k] == a[k] + b[k], k
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    clean = _run(engine, 11, 4099, dtype, snr_db=300.0, method=_lib.DEMOD_QAM_SLICER)[0]
    ref = _run(engine, 11, 4099, dtype, wave=0, snr_db=300.0, method=_lib.DEMOD_QAM_SLICER)[0]
    assert clean["n_realizations"] == 4099 and clean["sym_errors"] <= 2e-4 * 4099 * 1024
    assert abs(clean["sym_errors"] - ref["sym_errors"]) <= 1e-5 * 4099 * 1024 + 2


# A tap BEYOND the cyclic prefix (inter-symbol interference; zeros in front of the first symbol, channels/fading.py:1092-1118): inside
# the wavefront kernel's envelope since round 6 (the previous symbol's end is carried in LDS; until then such a call fell through to
# the batched kernels, which stay the other side of this comparison)
ISI_CASES = [dict(fft=1024, mod="qam", M=16, snr_db=22.0, cp_size=4, tap_delays_samples=(0, 3, 9), tap_powers_dB=(0.0, -3.0, -6.0), n_ofdm_sym=2),
             dict(fft=1024, mod="qam", M=64, snr_db=30.0, cp_size=16, tap_delays_samples=(0, 1, 17, 40, 200), n_ofdm_sym=3),
             dict(fft=256, mod="qpsk", M=4, snr_db=12.0, cp_size=0, tap_delays_samples=(0, 16, 100), tap_powers_dB=(0.0, -3.0, -6.0),
                  n_ofdm_sym=4, num_used=200),
             dict(fft=512, mod="qam", M=16, snr_db=20.0, cp_size=33, tap_delays_samples=(0, 33, 34, 131), n_ofdm_sym=3,
                  tap_powers_dB=(0.0, -2.0, -4.0, -6.0), method=_lib.DEMOD_QAM_SLICER),
             dict(fft=2048, mod="qam", M=16, snr_db=20.0, cp_size=17, tap_delays_samples=(0, 5, 90), tap_powers_dB=(0.0, -3.0, -6.0),
                  n_ofdm_sym=2)]


@pytest.mark.parametrize("case", range(len(ISI_CASES)))
def test_a_delay_beyond_the_prefix(engine, case):
    kw = dict(ISI_CASES[case])
    mod, M, fft = kw.pop("mod"), kw.pop("M"), kw.pop("fft")
    engine.set_constellation(chains.constellation(mod, M), _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)
    first, count = 5, 11
    okw = dict(mod=mod, M=M, fft_size=fft, cp_size=kw["cp_size"], num_used=kw.get("num_used"), n_ofdm_sym=kw["n_ofdm_sym"],
               snr_db=kw["snr_db"], Fd=10.0, L=8, tap_powers_dB=kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)),
               tap_delays_samples=kw["tap_delays_samples"])
    want = [chains.chain_ofdm_tdl(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want], dtype=np.int64)
    want_be = np.array([w["bit_errors"] for w in want], dtype=np.int64)
    res, se, be = _run(engine, first, count, "f64", fft=fft, **kw)
    assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (se, want_se)
    assert want_se.sum() > 0
    for dtype in ("f64", "f32"):                          # kernel against kernel over several passes of a wavefront's loop
        a = _run(engine, 1, 3000, dtype, fft=fft, **kw)
        b = _run(engine, 1, 3000, dtype, wave=0, fft=fft, **kw)
        if dtype == "f64":
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            assert np.array_equal(a[1][4:4 + count], se)
        else:
            assert np.max(np.abs(a[1].astype(np.int64) - b[1].astype(np.int64))) <= 3
            assert abs(a[0]["sym_errors"] - b[0]["sym_errors"]) <= 1e-4 * a[0]["n_symbols"] * 3000 + 2


# ---- the other sizes of the wavefront kernel: radix-4 stages on the wavefront's planes (256, 512, 2048; 2048 in complex64) ----
SIZE_CASES = [dict(mod="qpsk", M=4, snr_db=20.0),
              dict(mod="qam", M=16, snr_db=24.0, used_frac=0.6, n_ofdm_sym=2, cp_size=9, tap_delays_samples=(0, 2, 5, 9),
                   tap_powers_dB=(0.0, -2.0, -5.0, -8.0), Fd=200.0),                         # partial band, odd prefix, 2 symbols
              dict(mod="qam", M=64, snr_db=30.0, cp_size=64, n_ofdm_sym=3, method=_lib.DEMOD_QAM_SLICER,
                   tap_delays_samples=(0, 1, 7, 20, 33, 50, 63, 64), tap_powers_dB=(0.0, -1.0, -2.0, -3.0, -4.0, -5.0, -6.0, -7.0))]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("fft", [256, 512, 2048])
@pytest.mark.parametrize("case", range(len(SIZE_CASES)))
def test_wave_kernel_other_sizes(engine, case, fft, dtype):
    kw = dict(SIZE_CASES[case])
    mod, M = kw.pop("mod"), kw.pop("M")
    used = fft if "used_frac" not in kw else 2 * int(kw.pop("used_frac") * fft / 2)
    kw["num_used"] = used
    engine.set_constellation(chains.constellation(mod, M), _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)
    first, count = (1 << 33) + 4242, 13
    okw = dict(mod=mod, M=M, fft_size=fft, cp_size=kw.get("cp_size", 16), num_used=used,
               n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], Fd=kw.get("Fd", 10.0), L=kw.get("L", 8),
               tap_powers_dB=kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)),
               tap_delays_samples=kw.get("tap_delays_samples", (0, 1, 2, 3, 4)))
    want = [chains.chain_ofdm_tdl(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want], dtype=np.int64)
    want_be = np.array([w["bit_errors"] for w in want], dtype=np.int64)
    nsym, nbits = want[0]["num_symbols"], want[0]["num_bits"]
    res, se, be = _run(engine, first, count, dtype, fft=fft, **kw)
    old, se_o, be_o = _run(engine, first, count, dtype, wave=0, fft=fft, **kw)
    assert res["n_symbols"] == nsym and res["n_bits"] == nbits and res["n_realizations"] == count
    if dtype == "f64":
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (case, fft, se, want_se)
        assert np.array_equal(se, se_o) and np.array_equal(be, be_o)
    else:
        assert abs(int(se.sum()) - int(want_se.sum())) <= 1e-4 * count * nsym + 2
        assert abs(int(be.sum()) - int(want_be.sum())) <= 1e-4 * count * nbits + 2
        assert np.max(np.abs(se.astype(np.int64) - want_se)) <= 3
        assert np.max(np.abs(se.astype(np.int64) - se_o.astype(np.int64))) <= 3
    a = _run(engine, first, 5, dtype, fft=fft, **kw)
    b = _run(engine, first + 5, count - 5, dtype, fft=fft, **kw)
    assert np.array_equal(np.concatenate([a[1], b[1]]), se) and np.array_equal(np.concatenate([a[2], b[2]]), be)


# ---- fft_size 2048 with TWO WAVEFRONTS PER REALIZATION (csrc/siso_tdl_hw.hpp: k_run_ofdm_tdl_hw, the default at 2048 since round 6;
#      option tdl_kernel = 3: the one-wavefront kernel it replaces there) ----
HW_CASES = [dict(mod="qpsk", M=4, snr_db=20.0),                                                       # config 3's link at 2048
            dict(mod="qam", M=16, snr_db=24.0, num_used=1200, n_ofdm_sym=2, cp_size=9,               # partial band, ODD prefix: the
                 tap_delays_samples=(0, 2, 5, 9), tap_powers_dB=(0.0, -2.0, -5.0, -8.0), Fd=200.0),   # unpaired noise draws
            dict(mod="qam", M=64, snr_db=30.0, cp_size=64, n_ofdm_sym=3, method=_lib.DEMOD_QAM_SLICER,
                 tap_delays_samples=(0, 1, 7, 20, 33, 50, 63, 64), tap_powers_dB=(0.0, -1.0, -2.0, -3.0, -4.0, -5.0, -6.0, -7.0)),
            dict(mod="psk", M=8, snr_db=18.0, cp_size=256, n_ofdm_sym=2, Fd=700.0,                    # the longest reach (256 = the
                 tap_delays_samples=(0, 1, 128, 255, 256), tap_powers_dB=(0.0, -1.0, -2.0, -3.0, -4.0)),   # parity planes' 128), grid search
            dict(mod="qam", M=256, snr_db=36.0, cp_size=33, n_ofdm_sym=3, num_used=2, tap_delays_samples=(0, 3),   # odd symbol length: the
                 tap_powers_dB=(0.0, -4.0)),                                                           # pairing alternates per symbol
            dict(mod="psk", M=2, snr_db=5.0, cp_size=16, num_used=2046, Fd=1500.0)]


@pytest.mark.parametrize("case", range(len(HW_CASES)))
def test_half_wave_kernel_at_2048_against_the_oracle_and_the_one_wavefront_kernel(engine, case):
    kw = dict(HW_CASES[case])
    mod, M = kw.pop("mod"), kw.pop("M")
    engine.set_constellation(chains.constellation(mod, M), _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)
    first, count = (1 << 35) + 99, 9
    okw = dict(mod=mod, M=M, fft_size=2048, cp_size=kw.get("cp_size", 16), num_used=kw.get("num_used"), n_ofdm_sym=kw.get("n_ofdm_sym", 1),
               snr_db=kw["snr_db"], Fd=kw.get("Fd", 10.0), L=8, tap_powers_dB=kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)),
               tap_delays_samples=kw.get("tap_delays_samples", (0, 1, 2, 3, 4)))
    want = [chains.chain_ofdm_tdl(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want], dtype=np.int64)
    want_be = np.array([w["bit_errors"] for w in want], dtype=np.int64)
    res, se, be = _run(engine, first, count, "f64", fft=2048, **kw)
    assert res["n_symbols"] == want[0]["num_symbols"] and res["n_realizations"] == count
    assert np.array_equal(se, want_se) and np.array_equal(be, want_be), (se, want_se)
    # kernel against kernel over several passes of a workgroup's loop, an ODD count (the last pair's second slot is not a realization),
    # one realization, and a split of the range
    for dtype in ("f64", "f32"):
        n = 3001
        a = _run(engine, first - 50, n, dtype, fft=2048, **kw)
        with engine.options(tdl_kernel=3):
            b = engine_run_other(engine, first - 50, n, dtype, **kw)
        if dtype == "f64":
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            assert np.array_equal(a[1][50:50 + count], se)
        else:
            assert np.max(np.abs(a[1].astype(np.int64) - b[1].astype(np.int64))) <= 3
            assert abs(a[0]["sym_errors"] - b[0]["sym_errors"]) <= 1e-4 * a[0]["n_symbols"] * n + 2
        one = _run(engine, first - 50 + 7, 1, dtype, fft=2048, **kw)
        assert one[1][0] == a[1][7] and one[2][0] == a[2][7] and one[0]["n_realizations"] == 1
        lo, hi = _run(engine, first - 50, 1234, dtype, fft=2048, **kw), _run(engine, first - 50 + 1234, n - 1234, dtype, fft=2048, **kw)
        assert np.array_equal(np.concatenate([lo[1], hi[1]]), a[1]) and np.array_equal(np.concatenate([lo[2], hi[2]]), a[2])


def engine_run_other(engine, first, count, dtype, **kw):
    """_run without its own tdl_kernel option (the caller chose one)."""
    from pyphysim_amd.channels import discretize_profile
    Ts = kw.get("Ts", 1.0 / (15e3 * 1024))
    p_lin, d_idx = discretize_profile(np.asarray(kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)), dtype=float),
                                      np.asarray(kw.get("tap_delays_samples", (0, 1, 2, 3, 4)), dtype=float) * Ts, Ts)
    nv = 1.0 / omodem.dB2Linear(kw.get("snr_db", 20.0))
    return engine.run_ofdm_tdl(2048, kw.get("cp_size", 16), kw.get("num_used") or 2048, kw.get("n_ofdm_sym", 1), nv, p_lin, d_idx, SEED,
                               first, count, Fd=kw.get("Fd", 10.0), Ts=Ts, L=kw.get("L", 8), method=kw.get("method", _lib.DEMOD_MINDIST),
                               dtype=dtype, per_realization=True)
