"""Oracle: TDL profile discretisation, Jakes fading, TDL convolution, AWGN.  TEST INFRASTRUCTURE.

NumPy float64 / complex128 restatement of (paths relative to the reference):
  pyphysim/channels/fading.py:63-87,272-304   TdlChannelProfile / discretisation
  pyphysim/channels/fading.py:327-353         COST259 TU / RA / HT profiles
  pyphysim/channels/fading_generators.py:319-351,403-425  Jakes ctor, phi/psi draw
  pyphysim/channels/fading_generators.py:427-523  time axis + sum of sinusoids
  pyphysim/channels/fading.py:908-959         taps = fading * sqrt(tap power)
  pyphysim/channels/fading.py:1046-1124       corrupt_data (SISO and MIMO)
  pyphysim/channels/singleuser.py:130-151     SuChannel path-loss scale
  pyphysim/util/misc.py:327-355               randn_c
"""
import math

import numpy as np

COST259_TU = (np.array([-5.7, -7.6, -10.1, -10.2, -10.2, -11.5, -13.4, -16.3, -16.9, -17.1,
                        -17.4, -19, -19, -19.8, -21.5, -21.6, -22.1, -22.6, -23.5, -24.3]),
              np.array([0, 217, 512, 514, 517, 674, 882, 1230, 1287, 1311, 1349, 1533, 1535,
                        1622, 1818, 1836, 1884, 1943, 2048, 2140]) * 1e-9)
COST259_RA = (np.array([-5.2, -6.4, -8.4, -9.3, -10.0, -13.1, -15.3, -18.5, -20.4, -22.4]),
              np.array([0., 42., 101., 129., 149., 245., 312., 410., 469., 528]) * 1e-9)
COST259_HT = (np.array([-3.6, -8.9, -10.2, -11.5, -11.8, -12.7, -13.0, -16.2, -17.3, -17.7,
                        -17.6, -22.7, -24.1, -25.8, -25.8, -26.2, -29.0, -29.9, -30.0, -30.7]),
              np.array([0., 356., 441., 528., 546., 609., 625., 842., 916., 941., 15000.,
                        16172., 16492., 16876., 16882., 16978., 17615., 17827., 17849.,
                        18016.]) * 1e-9)


def discretize_profile(tap_powers_dB, tap_delays, Ts):
    """fading.py:272-304 + the dB round trip of get_discretize_profile/ctor (:77-78).

    Returns (tap_powers_linear, delay_indexes): delays rounded to sample indexes and
    merged, powers summed per index, normalised to sum 1, converted to dB and back."""
    tap_powers_dB = np.asarray(tap_powers_dB, dtype=float)
    tap_delays = np.asarray(tap_delays, dtype=float)
    idx, inverse = np.unique(np.round(tap_delays / Ts).astype(int).flatten(), return_inverse=True)
    lin = np.zeros(idx.size)
    for i, v in enumerate(10.0 ** (tap_powers_dB / 10.0)):
        lin[inverse[i]] += v
    lin /= np.sum(lin)
    dB = 10.0 * np.log10(lin)
    return np.power(10, dB / 10.0), idx


def jakes_time_axis(t0, Ts, n):
    """fading_generators.py:459-467.  np.arange(t0, n*Ts + t0, Ts*1.0000000001) evaluates
    t_k = t0 + k*delta with delta = fl(fl(t0 + step) - t0); returns (t, next_t0)."""
    t = np.arange(t0, n * Ts + t0, Ts * 1.0000000001)
    return t, t[-1] + Ts


def jakes_samples(phi, psi, Fd, t):
    """fading_generators.py:519-522.  phi, psi: [L, *shape, 1]; t: [n] -> h [*shape, n]."""
    L = phi.shape[0]
    tt = t.reshape((1,) * (phi.ndim - 1) + (-1,))
    return math.sqrt(1.0 / L) * np.sum(
        np.exp(1j * (2 * np.pi * Fd * np.cos(phi) * tt + psi)), axis=0)


def tdl_taps(fading, tap_powers_linear):
    """fading.py:949-956: fading [taps, ..., n] * sqrt(p_i)."""
    shp = (len(tap_powers_linear),) + (1,) * (fading.ndim - 1)
    return fading * np.sqrt(np.reshape(np.asarray(tap_powers_linear), shp))


def tdl_apply(signal, taps, delay_indexes):
    """fading.py:1080-1090 (SISO): y[d_i + n] += g_i[n] * x[n]; length N + max_delay."""
    signal = np.asarray(signal)
    n = signal.shape[-1]
    out = np.zeros(n + int(delay_indexes[-1]), dtype=complex)
    for i, d in enumerate(delay_indexes):
        out[d:d + n] += taps[i] * signal
    return out


def tdl_apply_mimo(signal, taps, delay_indexes):
    """fading.py:1107-1117 (MIMO, forward direction): taps [taps, Nr, Nt, n], signal [Nt, n]."""
    n = signal.shape[-1]
    nr, nt = taps.shape[1], taps.shape[2]
    out = np.zeros((nr, n + int(delay_indexes[-1])), dtype=complex)
    for i, d in enumerate(delay_indexes):
        for tx in range(nt):
            out[:, d:d + n] += taps[i, :, tx, :] * signal[tx]
    return out


def mean_freq_response(taps, delay_indexes, fft_size, cp_size, n_sym):
    """Per-OFDM-symbol mean of TdlImpulseResponse.get_freq_response (fading.py:513-536; the averaging of
    ofdm.py:545-547 generalised to MIMO).  taps [S, ..., n_sym*(fft+cp)] -> [n_sym, fft, ...]."""
    n_pad = int(delay_indexes[-1]) + 1
    dense = np.zeros((n_pad,) + taps.shape[1:], dtype=complex)
    dense[np.asarray(delay_indexes)] = taps
    fr = np.fft.fft(dense, fft_size, axis=0)                                   # [fft, ..., n]
    fr = fr.reshape(fr.shape[:-1] + (n_sym, fft_size + cp_size)).mean(axis=-1)  # [fft, ..., n_sym]
    return np.moveaxis(fr, -1, 0)                                               # [n_sym, fft, ...]


def mean_freq_response_linear(taps, delay_indexes, fft_size, cp_size, n_sym):
    """The same quantity with the two linear steps swapped: mean over the symbol's samples FIRST, then one DFT per
    symbol -- mean_n FFT(g[:, n]) == FFT(mean_n g[:, n]).  The reference computes the literal
    form above (one FFT per time sample: 1 040 transforms of 1 024 x 16 at config-4 size, 1.3 s of the oracle's 1.4 s
    per realization); the deep GPU parity tests (hundreds of realizations) use this one, which tests/test_oracle_golden.py
    holds to the literal form at 1e-13 on the golden case."""
    m = taps.reshape(taps.shape[:-1] + (n_sym, fft_size + cp_size)).mean(axis=-1)   # [S, ..., n_sym]
    n_pad = int(delay_indexes[-1]) + 1
    dense = np.zeros((n_pad,) + m.shape[1:], dtype=complex)
    dense[np.asarray(delay_indexes)] = m
    return np.moveaxis(np.fft.fft(dense, fft_size, axis=0), -1, 0)              # [n_sym, fft, ...]


def corrupt_data_in_freq_domain(signal, taps_per_block, delay_indexes, fft_size, carrier_indexes=None):
    """fading.py:1126-1287: block i uses the frequency response of impulse response i (taps_per_block
    [S, ..., n_blocks]) on `carrier_indexes` (None = all bins, natural order)."""
    n_blocks = taps_per_block.shape[-1]
    n_pad = int(delay_indexes[-1]) + 1
    dense = np.zeros((n_pad,) + taps_per_block.shape[1:], dtype=complex)
    dense[np.asarray(delay_indexes)] = taps_per_block
    fr = np.fft.fft(dense, fft_size, axis=0)                     # [fft, ..., n_blocks]
    pick = slice(None) if carrier_indexes is None else carrier_indexes
    signal = np.asarray(signal)
    if taps_per_block.ndim == 2:
        bs = signal.size // n_blocks
        out = np.empty(signal.size, dtype=complex)
        for i in range(n_blocks):
            out[i * bs:(i + 1) * bs] = fr[pick, i] * signal[i * bs:(i + 1) * bs]
        return out
    nr, nt = taps_per_block.shape[1:3]
    bs = signal.shape[-1] // n_blocks
    out = np.zeros((signal.shape[-1], nr), dtype=complex)
    for i in range(n_blocks):
        f = fr[pick, :, :, i]
        for tx in range(nt):
            out[i * bs:(i + 1) * bs, :] += f[:, :, tx] * signal[tx, i * bs:(i + 1) * bs, np.newaxis]
    return out.T
