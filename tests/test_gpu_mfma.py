"""GPU: the matrix-core config-4 kernel (csrc/pipeline_mimo_mfma.hip: f32, FFT 1024, 4x4) against the oracle chain under
the same Philox keying and against the VALU kernel it replaces (engine option no_mfma = mcle_ctx_set_option(MCLE_OPT_NO_MFMA) selects that one).

The MFMA kernel evaluates the same link with a different (equally f32) association of the sums, so per-realization
counts may differ from the VALU kernel's by a rounding-level tie now and then, never systematically; against the f64
oracle both sit inside the same |dSER| <= 1e-4 band."""
import os

import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 424242


def _run(engine, first, count, mfma=True, variant=None, **kw):
    with engine.options(no_mfma=0 if mfma else 1, mfma_variant=int(variant or 0), f32_mfma=1 if mfma else 0):
        nv = 1.0 / omodem.dB2Linear(kw.get("snr_db", 25.0))
        return engine.run_mimo_ofdm(4, 4, 1024, kw.get("cp_size", 16), kw.get("num_used") or 1024,
                                    kw.get("n_ofdm_sym", 1), nv, SEED, first, count, mmse=kw.get("mmse", True),
                                    method=kw.get("method", _lib.DEMOD_QAM_SLICER), dtype="f32", per_realization=True)


CASES = [dict(M=64, snr_db=25.0),
         dict(M=64, snr_db=25.0, num_used=600, n_ofdm_sym=2),          # partial band, two OFDM symbols
         dict(M=16, snr_db=18.0, cp_size=7, mmse=False),               # odd CP: unpaired noise draws; zero forcing
         dict(M=256, snr_db=32.0, cp_size=0),
         dict(M=4, snr_db=8.0, num_used=1022)]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("variant", [36, 32, 30, 21])
def test_mfma_kernel_against_the_oracle_and_the_valu_kernel(engine, case, variant):
    kw = dict(CASES[case])
    M = kw.pop("M")
    engine.set_constellation(chains.constellation("qam", M), _lib.CONST_QAM)
    first, count = 31337, 24
    okw = dict(mod="qam", M=M, nt=4, nr=4, fft_size=1024, cp_size=kw.get("cp_size", 16), num_used=kw.get("num_used"),
               n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], mmse=kw.get("mmse", True))
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want], dtype=np.int64)
    want_be = np.array([w["bit_errors"] for w in want], dtype=np.int64)
    nsym, nbits = want[0]["num_symbols"], want[0]["num_bits"]
    res, se, be = _run(engine, first, count, variant=variant, **kw)
    assert res["n_symbols"] == nsym and res["n_bits"] == nbits and res["n_realizations"] + res["n_skipped"] == count
    ok = se != 0xFFFFFFFF
    assert abs(int(se[ok].sum()) - int(want_se[ok].sum())) <= 1e-4 * count * nsym + 2
    assert abs(int(be[ok].sum()) - int(want_be[ok].sum())) <= 1e-4 * count * nbits + 2
    assert np.max(np.abs(se[ok].astype(np.int64) - want_se[ok])) <= 3          # boundary ties only
    res_v, se_v, be_v = _run(engine, first, count, mfma=False, **kw)
    assert np.array_equal(se == 0xFFFFFFFF, se_v == 0xFFFFFFFF)
    assert np.max(np.abs(se[ok].astype(np.int64) - se_v[ok].astype(np.int64))) <= 3
    assert res["sym_errors_sq"] == int((se[ok].astype(np.int64) ** 2).sum())
    assert res["bit_errors"] == int(be[ok].astype(np.int64).sum())


def test_mfma_kernel_min_distance_equals_slicer_and_is_split_invariant(engine):
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    first, count = 1 << 33, 700            # more realizations than one workgroup's 64-record filter batch
    res_s, se_s, be_s = _run(engine, first, count)
    res_m, se_m, be_m = _run(engine, first, count, method=_lib.DEMOD_MINDIST)
    assert np.max(np.abs(se_s.astype(np.int64) - se_m.astype(np.int64))) <= 2
    assert abs(int(se_s.astype(np.int64).sum()) - int(se_m.astype(np.int64).sum())) <= 1e-5 * count * 4096 + 2
    # bit-identical from run to run and under any split of the realization range
    res_2, se_2, be_2 = _run(engine, first, count)
    assert np.array_equal(se_s, se_2) and np.array_equal(be_s, be_2)
    a = _run(engine, first, 123)
    b = _run(engine, first + 123, count - 123)
    assert np.array_equal(np.concatenate([a[1], b[1]]), se_s) and np.array_equal(np.concatenate([a[2], b[2]]), be_s)
    assert a[0]["sym_errors"] + b[0]["sym_errors"] == res_s["sym_errors"]


def test_zero_noise_round_trip(engine):
    """No noise, zero forcing: every symbol of every stream comes back (transforms, channel and decode are exact to
    f32 rounding, far inside the decision regions)."""
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    res = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, 0.0, SEED, 5, 4096, mmse=False, method=_lib.DEMOD_QAM_SLICER,
                               dtype="f32")
    assert res["n_realizations"] > 4000
    assert res["sym_errors"] <= 1e-6 * res["n_realizations"] * 4096       # ill-conditioned H now and then


def test_mfma_kernel_aggregate_ser_against_the_oracle(engine):
    """VERDICT r02 (parity soft spot): the complex64 matrix-core kernel against the ORACLE (not against the VALU kernel)
    over a large sample -- 2 048 realizations = 8.4e6 symbols of BASELINE config 4 on the same Philox draws: |dSER| and
    |dBER| <= 1e-5 (the north star allows 1e-4), no systematic sign, both demodulators.  This is bench.py's
    ser_abs_err_vs_oracle made a test."""
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    first, n = 50000, 2048
    kw = dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=25.0, mmse=True)
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + n)]
    want_se = np.array([w["symbol_errors"] for w in want], dtype=np.int64)
    want_be = np.array([w["bit_errors"] for w in want], dtype=np.int64)
    for method in (_lib.DEMOD_QAM_SLICER, _lib.DEMOD_MINDIST):
        res, se, be = _run(engine, first, n, method=method)
        d = se.astype(np.int64) - want_se
        assert abs(int(d.sum())) <= 1e-5 * n * 4096, (method, int(d.sum()))
        assert abs(int((be.astype(np.int64) - want_be).sum())) <= 1e-5 * n * 4096 * 6
        assert np.max(np.abs(d)) <= 3 and np.count_nonzero(d) <= 0.05 * n       # rounding-level ties, rare and unsigned
        assert res["n_realizations"] == n and res["n_skipped"] == 0
    # the complex128 kernel on the same realizations: exact
    nv = 1.0 / omodem.dB2Linear(25.0)
    _, se64, be64 = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, first, n, dtype="f64", per_realization=True)
    assert np.array_equal(se64, want_se) and np.array_equal(be64, want_be)


# ---- config 3 on the matrix cores (csrc/pipeline_siso_tdl.hip: k_run_ofdm_tdl_mfma) ------------------------------------
def _run_tdl(engine, first, count, mfma=True, waves=None, **kw):
    from pyphysim_amd.channels import discretize_profile
    with engine.options(no_mfma=0 if mfma else 1, tdl_mfma_waves=int(waves or 0), tdl_kernel=1):
        Ts = kw.get("Ts", 1.0 / (15e3 * 1024))
        p_lin, d_idx = discretize_profile(np.asarray(kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)), dtype=float),
                                          np.asarray(kw.get("tap_delays_samples", (0, 1, 2, 3, 4)), dtype=float) * Ts, Ts)
        nv = 1.0 / omodem.dB2Linear(kw.get("snr_db", 20.0))
        return engine.run_ofdm_tdl(1024, kw.get("cp_size", 16), kw.get("num_used") or 1024, kw.get("n_ofdm_sym", 1), nv,
                                   p_lin, d_idx, SEED, first, count, Fd=kw.get("Fd", 10.0), Ts=Ts, L=kw.get("L", 8),
                                   method=kw.get("method", _lib.DEMOD_MINDIST), dtype="f32", per_realization=True)


TDL_CASES = [dict(mod="qpsk", M=4, snr_db=20.0),                                            # BASELINE config 3
             dict(mod="qam", M=16, snr_db=24.0, num_used=608, n_ofdm_sym=2, L=12),          # partial band, 2 symbols
             dict(mod="qam", M=64, snr_db=30.0, cp_size=9, tap_delays_samples=(0, 2, 5, 9), # odd CP: unpaired noise
                  tap_powers_dB=(0.0, -2.0, -5.0, -8.0), Fd=200.0),
             dict(mod="psk", M=8, snr_db=18.0, cp_size=32, tap_delays_samples=(0, 7, 17, 31),
                  tap_powers_dB=(0.0, -1.0, -3.0, -6.0), Fd=900.0, n_ofdm_sym=3)]           # higher polynomial order


@pytest.mark.parametrize("case", range(len(TDL_CASES)))
@pytest.mark.parametrize("waves", [2, 3, 32])     # 32: three waves per SIMD with two realizations per pass
def test_tdl_mfma_kernel_against_the_oracle_and_the_valu_kernel(engine, case, waves):
    kw = dict(TDL_CASES[case])
    mod, M = kw.pop("mod"), kw.pop("M")
    kind = _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC
    engine.set_constellation(chains.constellation(mod, M), kind)
    first, count = 70001, 22                  # not a multiple of the four slots of a pass
    okw = dict(mod=mod, M=M, fft_size=1024, cp_size=kw.get("cp_size", 16), num_used=kw.get("num_used"),
               n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], Fd=kw.get("Fd", 10.0), L=kw.get("L", 8),
               tap_powers_dB=kw.get("tap_powers_dB", (0.0, -3.0, -6.0, -9.0, -12.0)),
               tap_delays_samples=kw.get("tap_delays_samples", (0, 1, 2, 3, 4)))
    want = [chains.chain_ofdm_tdl(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want], dtype=np.int64)
    want_be = np.array([w["bit_errors"] for w in want], dtype=np.int64)
    nsym, nbits = want[0]["num_symbols"], want[0]["num_bits"]
    res, se, be = _run_tdl(engine, first, count, waves=waves, **kw)
    assert res["n_symbols"] == nsym and res["n_bits"] == nbits and res["n_realizations"] == count
    assert abs(int(se.sum()) - int(want_se.sum())) <= 1e-4 * count * nsym + 2
    assert abs(int(be.sum()) - int(want_be.sum())) <= 1e-4 * count * nbits + 2
    assert np.max(np.abs(se.astype(np.int64) - want_se)) <= 3                 # boundary ties only
    res_v, se_v, be_v = _run_tdl(engine, first, count, mfma=False, **kw)
    assert np.max(np.abs(se.astype(np.int64) - se_v.astype(np.int64))) <= 3
    assert res["sym_errors_sq"] == int((se.astype(np.int64) ** 2).sum())
    assert res["bit_errors"] == int(be.astype(np.int64).sum())
    # bit-identical from run to run and under any split of the realization range
    a = _run_tdl(engine, first, 9, waves=waves, **kw)
    b = _run_tdl(engine, first + 9, count - 9, waves=waves, **kw)
    assert np.array_equal(np.concatenate([a[1], b[1]]), se) and np.array_equal(np.concatenate([a[2], b[2]]), be)


def test_tdl_mfma_zero_noise_round_trip(engine):
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    res, se, be = _run_tdl(engine, 11, 4099, snr_db=300.0, method=_lib.DEMOD_QAM_SLICER)
    assert res["n_realizations"] == 4099
    assert res["sym_errors"] <= 2e-4 * 4099 * 1024                            # deep fades of the one-tap channel only
    ref, se_v, be_v = _run_tdl(engine, 11, 4099, mfma=False, snr_db=300.0, method=_lib.DEMOD_QAM_SLICER)
    assert abs(res["sym_errors"] - ref["sym_errors"]) <= 1e-5 * 4099 * 1024 + 2


# ---- config 2 on the matrix cores (csrc/pipelines.hip: k_run_flat_mfma) -----------------------------------------------
def _run_flat(engine, first, count, mfma=True, **kw):
    with engine.options(no_mfma=0 if mfma else 1):
        nv = 0.0 if kw.get("snr_db") is None else 1.0 / omodem.dB2Linear(kw["snr_db"])
        return engine.run_flat_fading(kw["N"], nv, SEED, first, count, Fd=kw.get("Fd", 100.0), Ts=kw.get("Ts", 1e-3),
                                      L=kw.get("L", 8), method=kw.get("method", _lib.DEMOD_MINDIST), dtype="f32",
                                      per_realization=True)


FLAT_CASES = [dict(mod="qam", M=64, N=20000, snr_db=20.0, method=_lib.DEMOD_QAM_SLICER),        # BASELINE config 2, shortened
              dict(mod="qam", M=64, N=16384 + 1029, snr_db=24.0),                                # candidate grid, ragged tail
              dict(mod="qam", M=16, N=4099, snr_db=14.0, L=16, Fd=250.0, method=_lib.DEMOD_QAM_SLICER),
              dict(mod="psk", M=8, N=1000, snr_db=12.0, L=16, Ts=1e-4),                          # direct search, 16 rays
              dict(mod="bpsk", M=2, N=17, snr_db=3.0),
              dict(mod="qam", M=256, N=3000, snr_db=30.0, Fd=5.0, Ts=5e-3)]


@pytest.mark.parametrize("case", range(len(FLAT_CASES)))
def test_flat_mfma_kernel_against_the_oracle_and_the_valu_kernel(engine, case):
    kw = dict(FLAT_CASES[case])
    mod, M = kw.pop("mod"), kw.pop("M")
    engine.set_constellation(chains.constellation(mod, M), _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)
    first, count = (1 << 33) + 5, 7
    okw = dict(mod=mod, M=M, N=kw["N"], snr_db=kw["snr_db"], Fd=kw.get("Fd", 100.0), Ts=kw.get("Ts", 1e-3), L=kw.get("L", 8))
    want = [chains.chain_flat_jakes(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want], dtype=np.int64)
    want_be = np.array([w["bit_errors"] for w in want], dtype=np.int64)
    nsym, nbits = want[0]["num_symbols"], want[0]["num_bits"]
    res, se, be = _run_flat(engine, first, count, **kw)
    assert res["n_symbols"] == nsym and res["n_bits"] == nbits and res["n_realizations"] == count
    assert abs(int(se.sum()) - int(want_se.sum())) <= 1e-4 * count * nsym + 2
    assert abs(int(be.sum()) - int(want_be.sum())) <= 1e-4 * count * nbits + 2
    assert np.max(np.abs(se.astype(np.int64) - want_se)) <= 3                 # boundary ties only
    res_v, se_v, be_v = _run_flat(engine, first, count, mfma=False, **kw)
    assert np.max(np.abs(se.astype(np.int64) - se_v.astype(np.int64))) <= 3
    assert np.max(np.abs(be.astype(np.int64) - be_v.astype(np.int64))) <= 4
    assert res["sym_errors_sq"] == int((se.astype(np.int64) ** 2).sum())
    assert res["bit_errors"] == int(be.astype(np.int64).sum())
    a = _run_flat(engine, first, 3, **kw)                                     # any split of the realization range
    b = _run_flat(engine, first + 3, count - 3, **kw)
    assert np.array_equal(np.concatenate([a[1], b[1]]), se) and np.array_equal(np.concatenate([a[2], b[2]]), be)


def test_flat_mfma_zero_noise_and_full_size(engine):
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    for L in (8, 16):
        for method in (_lib.DEMOD_QAM_SLICER, _lib.DEMOD_MINDIST):
            res, se, be = _run_flat(engine, 3, 5, N=100000, snr_db=None, L=L, method=method)
            assert res["n_realizations"] == 5 and res["sym_errors"] == 0      # h s / h = s whatever h is
    nv = dict(N=100000, snr_db=20.0, method=_lib.DEMOD_QAM_SLICER)
    res, se, be = _run_flat(engine, 0, 48, **nv)
    ref, se_v, be_v = _run_flat(engine, 0, 48, mfma=False, **nv)
    assert abs(res["sym_errors"] - ref["sym_errors"]) <= 1e-5 * 48 * 100000
    assert np.max(np.abs(se.astype(np.int64) - se_v.astype(np.int64))) <= 6
