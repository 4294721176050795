"""Host-side logic of the reference-interface mirror (no GPU): the object state, closed forms and
errors that never reach a kernel.  Known answers are the ones the reference's own tests hold
(tests/modulators_package_test.py:45-135, 169-187, 211-330, 366-553;
tests/channels_package_test.py:314-498), restated as data."""
import math

import numpy as np
import pytest

from pyphysim_amd import channels, modulators


# ---------------------------------------------------------------------------------------------
# modulators: constellations and theory
# ---------------------------------------------------------------------------------------------
def test_psk_tables_follow_the_reference_gray_order():
    p4, p8 = modulators.PSK(4), modulators.PSK(8)
    assert (p4.M, p4.K, p8.M, p8.K) == (4, 2, 8, 3)
    np.testing.assert_array_almost_equal(p4.symbols, [1, 1j, -1j, -1])
    r = math.sqrt(0.5)
    np.testing.assert_array_almost_equal(
        p8.symbols, [1, r + r * 1j, -r + r * 1j, 1j, r - r * 1j, -1j, -1, -r - r * 1j])
    # setPhaseOffset rebuilds the table in natural (non-Gray) order, as the reference does
    p4.setPhaseOffset(np.pi / 4)
    np.testing.assert_array_almost_equal(p4.symbols, [r + r * 1j, -r + r * 1j, -r - r * 1j, r - r * 1j])
    assert p4.name == "4-PSK" and modulators.BPSK().name == "BPSK"


def test_bpsk_and_qam_tables():
    b = modulators.BPSK()
    assert (b.M, b.K) == (2, 1)
    np.testing.assert_array_equal(b.symbols, [1, -1])
    q4, q16, q64 = modulators.QAM(4), modulators.QAM(16), modulators.QAM(64)
    assert (q4.K, q16.K, q64.K) == (2, 4, 6)
    r = math.sqrt(0.5)
    np.testing.assert_array_almost_equal(q4.symbols, [-r + r * 1j, r + r * 1j, -r - r * 1j, r - r * 1j])
    a, c = 0.94868330, 0.31622777
    np.testing.assert_array_almost_equal(
        q16.symbols[:8], [-a + a * 1j, -c + a * 1j, a + a * 1j, c + a * 1j,
                          -a + c * 1j, -c + c * 1j, a + c * 1j, c + c * 1j])
    np.testing.assert_array_almost_equal(q16.symbols[8:], np.conj(q16.symbols[:8]))
    # 64-QAM: first row of the reference's table and its unit average energy
    l4, l3, l2, l1 = 1.08012345, 0.77151675, 0.46291005, 0.15430335
    np.testing.assert_array_almost_equal(q64.symbols[:8].real, [-l4, -l3, -l1, -l2, l3, l4, l2, l1])
    np.testing.assert_array_almost_equal(q64.symbols[:8].imag, np.full(8, l4))
    np.testing.assert_array_almost_equal(q64.symbols[8:16].imag, np.full(8, l3))
    np.testing.assert_array_almost_equal(q64.symbols[16:24].imag, np.full(8, l1))
    assert np.mean(np.abs(q64.symbols) ** 2) == pytest.approx(1.0)
    for bad in (32, 63):
        with pytest.raises(ValueError):
            modulators.QAM(bad)


def test_theoretical_error_rates_known_answers():
    snr = np.array([-5, 0, 5, 10])
    p4, p8, b = modulators.PSK(4), modulators.PSK(8), modulators.BPSK()
    ser4 = np.array([0.57388349, 0.31731051, 0.07535798, 0.0015654])
    ser8 = np.array([0.76087121, 0.58837243, 0.33584978, 0.08700502])
    np.testing.assert_array_almost_equal(p4.calcTheoreticalSER(snr), ser4)
    np.testing.assert_array_almost_equal(p4.calcTheoreticalBER(snr), ser4 / 2)
    np.testing.assert_array_almost_equal(p8.calcTheoreticalSER(snr), ser8)
    np.testing.assert_array_almost_equal(p8.calcTheoreticalBER(snr), ser8 / 3)
    serb = np.array([2.13228018e-01, 7.86496035e-02, 5.95386715e-03, 3.87210822e-06])
    np.testing.assert_array_almost_equal(b.calcTheoreticalSER(snr), serb)
    np.testing.assert_array_almost_equal(b.calcTheoreticalBER(snr), serb)

    snr = np.array([0, 5, 10, 15, 20])
    known = {
        4: ([2.92139018e-01, 7.39382701e-02, 1.56478964e-03, 1.87220798e-08, 0],
            [1.58655254e-01, 3.76789881e-02, 7.82701129e-04, 9.36103999e-09, 7.61985302e-24]),
        16: ([7.40960364e-01, 5.37385132e-01, 2.22030850e-01, 1.77818422e-02, 1.16162909e-05],
             [2.45520317e-01, 1.59921014e-01, 5.89872026e-02, 4.46540036e-03, 2.90408116e-06]),
        64: ([0.92374224, 0.84846895, 0.67382633, 0.3476243, 0.05027041],
             [0.24128398, 0.2035767, 0.14296128, 0.06410074, 0.00848643]),
    }
    for M, (ser, ber) in known.items():
        q = modulators.QAM(M)
        np.testing.assert_array_almost_equal(q.calcTheoreticalSER(snr), ser)
        np.testing.assert_array_almost_equal(q.calcTheoreticalBER(snr), ber)


def test_packet_error_rate_and_spectral_efficiency():
    p4 = modulators.PSK(4)
    snr = np.array([10, 13])
    ber = p4.calcTheoreticalBER(snr)
    np.testing.assert_array_almost_equal(ber, [7.82701129e-04, 3.96924840e-06])
    for L in (1, 50, 120):
        per = p4.calcTheoreticalPER(snr, L)
        np.testing.assert_array_almost_equal(per, 1 - (1 - ber) ** L)
        np.testing.assert_array_almost_equal(p4.calcTheoreticalSpectralEfficiency(snr, L), 2 * (1 - per))
    np.testing.assert_array_almost_equal(p4.calcTheoreticalSpectralEfficiency(snr), 2 * (1 - ber))


def test_bpsk_rejects_non_binary_input_before_touching_the_device():
    with pytest.raises(ValueError):
        modulators.BPSK().modulate(2)


# ---------------------------------------------------------------------------------------------
# OFDM parameter logic
# ---------------------------------------------------------------------------------------------
def test_ofdm_parameters_and_errors():
    o = modulators.OFDM(64, 16, 52)
    assert (o.fft_size, o.cp_size, o.num_used_subcarriers) == (64, 16, 52)
    o.set_parameters(128, 32, 100)
    assert (o.fft_size, o.cp_size, o.num_used_subcarriers) == (128, 32, 100)
    for args in ((64, 16, 70), (64, 70, 52), (64, -2, 52), (64, 16, 51)):
        with pytest.raises(ValueError):
            o.set_parameters(*args)
    o.set_parameters(64, 16)
    assert o.num_used_subcarriers == 64


def test_ofdm_zero_padding_subcarrier_map_and_power_scale():
    o = modulators.OFDM(64, 16, 52)
    assert o._calc_zeropad(52) == (0, 1)
    assert o._calc_zeropad(104) == (0, 2)
    assert o._calc_zeropad(44) == (8, 1)
    assert o._calc_zeropad(109) == (47, 3)
    o.set_parameters(16, 4, 10)
    np.testing.assert_array_equal(o.get_used_subcarrier_indexes(), [11, 12, 13, 14, 15, 1, 2, 3, 4, 5])
    o.set_parameters(16, 4)
    np.testing.assert_array_equal(o.get_used_subcarrier_indexes(), np.r_[8:16, 0:8])
    o.set_parameters(64, 16, 52)
    assert o._calculate_power_scale() == pytest.approx(64.0 * 64.0 / (52 + 16))
    o.fft_size, o.cp_size, o.num_used_subcarriers = 1024., 100., 900.
    assert o._calculate_power_scale() == pytest.approx(1024. * 1024. / 1000.)


# ---------------------------------------------------------------------------------------------
# tapped-delay-line profiles
# ---------------------------------------------------------------------------------------------
def test_cost259_profiles_known_answers():
    tu, ra, ht = channels.COST259_TUx, channels.COST259_RAx, channels.COST259_HTx
    assert (tu.num_taps, ra.num_taps, ht.num_taps) == (20, 10, 20)
    assert (tu.name, ra.name, ht.name) == ("COST259_TU", "COST259_RA", "COST259_HT")
    assert tu.mean_excess_delay == pytest.approx(5.00428208169e-07, abs=1e-12)
    assert ra.mean_excess_delay == pytest.approx(8.85375638731e-08, abs=1e-12)
    assert ht.mean_excess_delay == pytest.approx(8.93899719191e-07, abs=1e-12)
    assert tu.rms_delay_spread == pytest.approx(5.000561653134637e-07, abs=1e-12)
    assert ra.rms_delay_spread == pytest.approx(1.0000823342626581e-07, abs=1e-12)
    assert ht.rms_delay_spread == pytest.approx(3.039829880190327e-06, abs=1e-12)
    for arr in (tu.tap_powers_dB, tu.tap_delays, tu.tap_powers_linear):
        with pytest.raises(ValueError):
            arr[0] = 30
    assert channels.TdlChannelProfile(np.array([0, -3, -10]), np.array([0, 1e-3, 5e-4])).name == "custom"
    assert channels.TdlChannelProfile(np.zeros(2), np.array([0, 1e-3]), name="some name").name == "some name"


def test_profile_discretisation_known_answers():
    tu, ra, ht = channels.COST259_TUx, channels.COST259_RAx, channels.COST259_HTx
    for p in (tu, ra, ht):
        assert not p.is_discretized
        with pytest.raises(RuntimeError):
            _ = p.num_taps_with_padding
    Ts = 3.255e-08
    tu_d, ra_d, ht_d = (p.get_discretize_profile(Ts) for p in (tu, ra, ht))
    assert (tu_d.num_taps, tu_d.num_taps_with_padding) == (15, 67)
    assert (ra_d.num_taps, ra_d.num_taps_with_padding) == (10, 17)
    assert (ht_d.num_taps, ht_d.num_taps_with_padding) == (18, 554)
    assert tu_d.Ts == Ts and tu_d.is_discretized
    assert tu_d.name == tu.name + " (discretized)"
    with pytest.raises(RuntimeError):
        tu_d.get_discretize_profile(Ts)

    # the 2048-subcarrier / 15 kHz grid of the reference's test: merged taps add their powers
    Ts = 1.0 / (15e3 * 2048)
    d = tu.get_discretize_profile(Ts)
    np.testing.assert_array_equal(d.tap_delays, [0, 7, 16, 21, 27, 38, 40, 41, 47, 50, 56, 58, 60, 63, 66])
    lin = tu.tap_powers_linear / tu.tap_powers_linear.sum()
    groups = [[0], [1], [2, 3, 4], [5], [6], [7], [8, 9], [10], [11, 12], [13], [14, 15], [16], [17], [18], [19]]
    np.testing.assert_array_almost_equal(d.tap_powers_linear, [lin[g].sum() for g in groups])
    assert d.tap_powers_linear.sum() == pytest.approx(1.0)

    # ten times the TU delays: nothing merges any more (channels_package_test.py:745-757)
    far = channels.TdlChannelProfile(tu.tap_powers_dB, 10 * tu.tap_delays).get_discretize_profile(3.255e-08)
    assert (far.num_taps, far.num_taps_with_padding) == (20, 658)


# ---------------------------------------------------------------------------------------------
# comm mirror: argument checks that precede any device work
# ---------------------------------------------------------------------------------------------
def test_block_diagonalizer_argument_errors():
    from pyphysim_amd.comm import blockdiagonalization as mbd
    bd = mbd.BlockDiagonalizer(3, 1.0, 0.1)
    with pytest.raises(AssertionError):        # rows not a multiple of the number of users (the reference's assert)
        bd.block_diagonalize(np.ones((5, 5), dtype=complex))
    with pytest.raises(NotImplementedError):   # non-square channels are outside this build
        bd.block_diagonalize_no_waterfilling(np.ones((6, 9), dtype=complex))
    assert (bd.num_users, bd.iPu, bd.noise_var) == (3, 1.0, 0.1)


def test_small_util_helpers_known_answers():
    """util/conversion.py and util/misc.py helpers; the expected values are the reference's own outputs."""
    from pyphysim_amd import util as u
    assert u.dBm2Linear(30.0) == 1.0 and abs(u.linear2dBm(0.5) - 26.989700043360187) < 1e-12
    assert abs(u.SNR_dB_to_EbN0_dB(10.0, 4) - 3.979400086720376) < 1e-12
    assert abs(u.EbN0_dB_to_SNR_dB(3.0, 6) - 10.781512503836437) < 1e-12
    assert list(u.binary2gray(np.arange(8))) == [0, 1, 3, 2, 6, 7, 5, 4]
    assert list(u.gray2binary(np.arange(8))) == [0, 1, 3, 2, 7, 6, 4, 5]
    assert u.gray2binary(200) == 143 and u.binary2gray(200) == 172 and u.xor(5, 3) == 6
    assert (u.int2bits(0), u.int2bits(5), u.int2bits(8)) == (1, 3, 4)
    with pytest.raises(ValueError):
        u.int2bits(-1)
    assert (u.pretty_time(3725.4), u.pretty_time(65.2), u.pretty_time(2.345)) == ("1h:02m:05s", "1m:05s", "2.35s")
    assert u.equal_dicts({"a": 1, "b": 2}, {"a": 1, "b": 3}, ["b"]) and not u.equal_dicts({"a": 1}, {"a": 2}, [])
    assert u.calc_shannon_sum_capacity(np.array([1.0, 3.0])) == 3.0
    x = np.array([1.0, 2.0, 0.5, -1.0, 3.0])
    assert np.allclose(u.calc_unorm_autocorr(x), [15.25, -0.5, 0.0, 5.0, 3.0])
    assert np.allclose(u.calc_autocorr(x), [1.0, -0.36521739, -0.32282609, 0.20869565, -0.02065217])
    lo, hi = u.calc_confidence_interval(1.0, 0.5, 100, 95)
    assert abs(lo - 0.902) < 1e-12 and abs(hi - 1.098) < 1e-12
    m = np.arange(24).reshape(4, 6)
    o = u.single_matrix_to_matrix_of_matrices(m, np.array([1, 3]), np.array([2, 4]))
    assert o.shape == (2, 2) and np.array_equal(o[1, 0], [[6, 7], [12, 13], [18, 19]])
    assert u.single_matrix_to_matrix_of_matrices(m, None, np.array([2, 4]))[1].shape == (4, 4)
    assert u.single_matrix_to_matrix_of_matrices(m, np.array([1, 3]))[1].shape == (3, 6)
    rs1, rs2 = np.random.RandomState(3), np.random.RandomState(3)
    want = (rs2.randn(2, 3) + 1j * rs2.randn(2, 3)) / np.sqrt(2.0)
    assert np.array_equal(u.randn_c_RS(rs1, 2, 3), want)


def test_mimo_base_surface():
    from pyphysim_amd import mimo
    with pytest.raises(ValueError, match="single receive antenna"):
        mimo.MisoBase.set_channel_matrix(mimo.MisoBase.__new__(mimo.MisoBase), np.ones((2, 3)))
    b = mimo.MisoBase.__new__(mimo.MisoBase)
    b.set_channel_matrix(np.ones(3))
    assert b._channel.shape == (1, 3)
    r = mimo.MRC.__new__(mimo.MRC)
    r.set_channel_matrix(np.ones(3))
    assert r._channel.shape == (3, 1)
    for name in ("encode", "decode"):
        with pytest.raises(NotImplementedError):
            getattr(mimo.MimoBase, name)(b, None)


def test_multiuser_pathloss_block_matrix_known_answer():
    """MultiUserChannelMatrix._from_small_matrix_to_big_matrix: the reference's own docstring example
    (channels/multiuser.py:902-919)."""
    from pyphysim_amd.multiuser import MultiUserChannelMatrix
    big = MultiUserChannelMatrix._from_small_matrix_to_big_matrix(np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]]),
                                                                  np.array([2, 4, 6]), np.array([2, 3, 5]), 3)
    assert big.shape == (12, 10)
    assert list(big[0]) == [1, 1, 2, 2, 2, 3, 3, 3, 3, 3] and list(big[2]) == [4, 4, 5, 5, 5, 6, 6, 6, 6, 6]
    assert list(big[-1]) == [7, 7, 8, 8, 8, 9, 9, 9, 9, 9] and np.array_equal(big[6], big[11])
    # K x (K + external sources): the ExtInt form
    big = MultiUserChannelMatrix._from_small_matrix_to_big_matrix(np.arange(6).reshape(2, 3), [1, 2], [2, 1, 3], 2, 3)
    assert big.shape == (3, 6) and list(big[0]) == [0, 0, 1, 2, 2, 2] and list(big[2]) == [3, 3, 4, 5, 5, 5]
