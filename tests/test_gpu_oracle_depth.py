"""GPU: DEPTH against the oracle for every default-dispatched fused kernel.  The per-shape parity tests compare
a handful of realizations with the oracle and thousands kernel-against-kernel; these compare thousands with the ORACLE
(oracle/chains.py under the same Philox keying): complex128 per-realization symbol and bit counts exact, complex64 aggregate
SER within 1e-5 and per-realization differences of boundary ties only.  Oracle budget ~60 s on the box's host (the two TDL
chains use the linear form of the mean frequency response; the LITERAL mean of per-sample DFTs is held at depth by the two
`..._against_the_literal_mean_of_dfts_...` tests below, 512 / 48 realizations; f1's 240 realizations live in tests/test_gpu_mimo_tdl_wave.py)."""
import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 16180339


def _oracle(fn, first, count, **kw):
    out = [fn(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + count)]
    return (np.array([o["symbol_errors"] for o in out], dtype=np.int64), np.array([o["bit_errors"] for o in out], dtype=np.int64),
            out[0]["num_symbols"], out[0]["num_bits"])


def _hold(dtype, res, se, be, want_se, want_be, nsym, nbits, ties=3):
    n = len(want_se)
    assert res["n_realizations"] == n and res["n_skipped"] == 0 and res["n_symbols"] == nsym and res["n_bits"] == nbits
    assert res["sym_errors"] == int(se.astype(np.int64).sum()) and res["bit_errors"] == int(be.astype(np.int64).sum())
    if dtype == "f64":
        assert np.array_equal(se, want_se), np.flatnonzero(se != want_se)[:8]
        assert np.array_equal(be, want_be)
    else:
        assert abs(int(se.sum()) - int(want_se.sum())) <= 1e-5 * n * nsym + 3
        assert abs(int(be.sum()) - int(want_be.sum())) <= 1e-5 * n * nbits + 4
        assert np.max(np.abs(se.astype(np.int64) - want_se)) <= ties


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_config3_wave_kernel_over_2048_realizations(engine, dtype):
    """BASELINE config 3 (QPSK, OFDM(1024, 16), five Jakes taps, 20 dB) on k_run_ofdm_tdl_wave, the default since round 4."""
    from pyphysim_amd.channels import discretize_profile
    engine.set_constellation(chains.constellation("qpsk", 4), _lib.CONST_GENERIC)
    Ts = 1.0 / (15e3 * 1024)
    kw = dict(mod="qpsk", M=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=20.0, Fd=10.0, Ts=Ts, L=8)
    first, count = 40000, 2048
    want_se, want_be, nsym, nbits = _oracle(chains.chain_ofdm_tdl, first, count, linear_mean=True, **kw)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    res, se, be = engine.run_ofdm_tdl(1024, 16, 1024, 1, 1.0 / omodem.dB2Linear(20.0), p_lin, d_idx, SEED, first, count, Fd=10.0, Ts=Ts,
                                      L=8, dtype=dtype, per_realization=True)
    _hold(dtype, res, se, be, want_se, want_be, nsym, nbits)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_config3_at_2048_points_two_wavefronts_per_realization_over_1024_realizations(engine, dtype):
    """Config 3's link at fft_size 2048 on k_run_ofdm_tdl_hw (csrc/siso_tdl_hw.hpp, the default there since the last day of round 6):
    1 025 realizations (an odd count: the last pair's second slot is not a realization) against the oracle, every per-realization
    count in complex128."""
    from pyphysim_amd.channels import discretize_profile
    engine.set_constellation(chains.constellation("qpsk", 4), _lib.CONST_GENERIC)
    Ts = 1.0 / (15e3 * 2048)
    kw = dict(mod="qpsk", M=4, fft_size=2048, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=20.0, Fd=10.0, Ts=Ts, L=8)
    first, count = 50000, 1025
    want_se, want_be, nsym, nbits = _oracle(chains.chain_ofdm_tdl, first, count, linear_mean=True, **kw)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    res, se, be = engine.run_ofdm_tdl(2048, 16, 2048, 1, 1.0 / omodem.dB2Linear(20.0), p_lin, d_idx, SEED, first, count, Fd=10.0, Ts=Ts,
                                      L=8, dtype=dtype, per_realization=True)
    _hold(dtype, res, se, be, want_se, want_be, nsym, nbits)


def test_config3_against_the_literal_mean_of_dfts_over_512_realizations(engine):
    """the 2 048-realization test above runs the oracle with the DFT of the MEAN taps (linearity); the
    reference takes the mean of the per-sample DFTs (channels/fading.py:513-536, modulators/ofdm.py:545-547) -- equal in exact
    arithmetic, rounded differently.  Here the LITERAL form, complex128, every per-realization count."""
    from pyphysim_amd.channels import discretize_profile
    engine.set_constellation(chains.constellation("qpsk", 4), _lib.CONST_GENERIC)
    Ts = 1.0 / (15e3 * 1024)
    kw = dict(mod="qpsk", M=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=20.0, Fd=10.0, Ts=Ts, L=8)
    first, count = 90000, 512
    want_se, want_be, nsym, nbits = _oracle(chains.chain_ofdm_tdl, first, count, linear_mean=False, **kw)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    res, se, be = engine.run_ofdm_tdl(1024, 16, 1024, 1, 1.0 / omodem.dB2Linear(20.0), p_lin, d_idx, SEED, first, count, Fd=10.0, Ts=Ts,
                                      L=8, dtype="f64", per_realization=True)
    _hold("f64", res, se, be, want_se, want_be, nsym, nbits)


def test_f1_against_the_literal_mean_of_dfts_over_48_realizations(engine):
    """The same for the frequency-selective 4 x 4 link at bench.py's f1 geometry (64-QAM, OFDM(1024, 16), five taps, 25 dB):
    48 realizations of the literal oracle (~0.8 s each), complex128, every per-realization count."""
    from pyphysim_amd.channels import discretize_profile
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    Ts = 1.0 / (15e3 * 1024)
    kw = dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=25.0, Fd=10.0, Ts=Ts, L=8,
              tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0), tap_delays_samples=(0, 1, 2, 3, 4), mmse=True)
    first, count = 123000, 48
    want_se, want_be, nsym, nbits = _oracle(chains.chain_mimo_ofdm_tdl, first, count, linear_mean=False, **kw)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    res, se, be = engine.run_mimo_ofdm_tdl(4, 4, 1024, 16, 1024, 1, 1.0 / omodem.dB2Linear(25.0), p_lin, d_idx, SEED, first, count,
                                           Fd=10.0, Ts=Ts, L=8, mmse=True, method=_lib.DEMOD_MINDIST, dtype="f64",
                                           per_realization=True)
    _hold("f64", res, se, be, want_se, want_be, nsym, nbits)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", [(256, 4, 4, 256), (512, 2, 3, 160), (2048, 2, 2, 64), (2048, 4, 4, 24)])
def test_f1_off_the_benchmark_size_on_the_parked_coefficient_kernels(engine, dtype, shape):
    """The frequency-selective link at 256 / 512 / 2048 points with the benchmark's Doppler per symbol -- the polynomial order at which
    the parked-coefficient kernels (pipeline_mimo_tdl_wave_*_{256,512,2048}k.hip, the default there since the last day of round 6) are
    picked, and in complex64 the record kernel's float evaluation of the rays' frequencies: per-realization counts against the oracle."""
    from pyphysim_amd.channels import discretize_profile
    fft, nt, nr, count = shape
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    Ts = 1.0 / (15e3 * fft)
    cp = 16 if fft < 2048 else 32
    kw = dict(mod="qam", M=64, nt=nt, nr=nr, fft_size=fft, cp_size=cp, num_used=None, n_ofdm_sym=1, snr_db=25.0, Fd=10.0, Ts=Ts, L=8,
              tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0), tap_delays_samples=(0, 1, 2, 3, 4), mmse=True)
    first = 7000 + fft
    want_se, want_be, nsym, nbits = _oracle(chains.chain_mimo_ofdm_tdl, first, count, linear_mean=True, **kw)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    res, se, be = engine.run_mimo_ofdm_tdl(nt, nr, fft, cp, fft, 1, 1.0 / omodem.dB2Linear(25.0), p_lin, d_idx, SEED, first, count,
                                           Fd=10.0, Ts=Ts, L=8, mmse=True, method=_lib.DEMOD_MINDIST, dtype=dtype, per_realization=True)
    _hold(dtype, res, se, be, want_se, want_be, nsym, nbits)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_config5_link_kernels_over_5000_realizations(engine, dtype):
    """BASELINE config 5 (K = 3, 2 x 2, closed-form IA, 16-QAM, 200 symbols per stream, 20 dB): k_ia_solve_links + k_ia_link."""
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    kw = dict(mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=200, snr_db=20.0)
    first, count = 7000, 5000
    out = [chains.chain_ia(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + count)]
    want_se = np.array([o["symbol_errors"] for o in out], dtype=np.int64)
    want_be = np.array([o["bit_errors"] for o in out], dtype=np.int64)
    res, se, be, cap, _ = engine.run_ia(200, 1.0 / omodem.dB2Linear(20.0), SEED, first, count, dtype=dtype, per_realization=True)
    _hold(dtype, res, se, be, want_se, want_be, 600, 2400)
    assert np.max(np.abs(cap - np.array([o["sum_capacity"] for o in out]))) <= (1e-7 if dtype == "f64" else 1e-2)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_block_diagonalisation_link_kernels_over_2000_realizations(engine, dtype):
    """f6 (apps/comp_BD/simulate_comp_simple.py: K = 3 cells of 2 x 2, 4-PSK, 500 symbols per stream, 15 dB): k_bd_solve_links +
    k_bd_link against the oracle chain in its canonical-phase form."""
    engine.set_constellation(chains.constellation("psk", 4), _lib.CONST_GENERIC)
    nv = 1.0 / (10.0 ** 1.5)
    kw = dict(mod="psk", M=4, K=3, nr=2, NSymbs=500, iPu=1.0, noise_var=nv, canonical=True)
    first, count = 300, 2000
    want_se, want_be, nsym, nbits = _oracle(chains.chain_bd, first, count, **kw)
    res, se, be = engine.run_bd(3, 2, 500, 1.0, nv, SEED, first, count, method=_lib.DEMOD_MINDIST, dtype=dtype, per_realization=True)
    _hold(dtype, res, se, be, want_se, want_be, nsym, nbits)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_config2_at_its_full_size_over_32_realizations(engine, dtype):
    """BASELINE config 2 (64-QAM, 1e5 symbols per realization, Jakes Fd 100 Hz / Ts 1 ms / L 8, 20 dB): k_run_flat<double> /
    k_run_flat_mfma, every one of the seven 16 384-symbol chunks of a realization."""
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    kw = dict(mod="qam", M=64, N=100000, snr_db=20.0, Fd=100.0, Ts=1e-3, L=8)
    first, count = 123, 32
    want_se, want_be, nsym, nbits = _oracle(chains.chain_flat_jakes, first, count, **kw)
    res, se, be = engine.run_flat_fading(100000, 1.0 / omodem.dB2Linear(20.0), SEED, first, count, Fd=100.0, Ts=1e-3, L=8,
                                         dtype=dtype, per_realization=True)
    _hold(dtype, res, se, be, want_se, want_be, nsym, nbits, ties=6)      # a handful of rounding-level ties in 1e5 symbols


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", [(256, 4, 4), (1024, 2, 4), (2048, 2, 2)])
def test_planar_family_off_the_headline_over_512_realizations(engine, dtype, shape):
    """Config 4's kernel family at three geometries that are not the benchmark's (k_run_mimo_ofdm_planar: radix-4 stages at 256 and
    2048, the rectangular radix-16 form at 1024 2 x 4), 64-QAM, 25 dB."""
    fft, nt, nr = shape
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    kw = dict(mod="qam", M=64, nt=nt, nr=nr, fft_size=fft, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=25.0, mmse=True)
    first, count = 90000, 512
    want_se, want_be, nsym, nbits = _oracle(chains.chain_mimo_ofdm, first, count, **kw)
    res, se, be = engine.run_mimo_ofdm(nt, nr, fft, 16, fft, 1, 1.0 / omodem.dB2Linear(25.0), SEED, first, count, mmse=True,
                                       dtype=dtype, per_realization=True)
    _hold(dtype, res, se, be, want_se, want_be, nsym, nbits)
