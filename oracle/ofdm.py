"""Oracle: OFDM modulate / demodulate and the one-tap equaliser.  TEST INFRASTRUCTURE.

NumPy complex128 restatement of (paths relative to the reference):
  pyphysim/modulators/ofdm.py:52-94    set_parameters (argument checks)
  pyphysim/modulators/ofdm.py:96-123   _calc_zeropad
  pyphysim/modulators/ofdm.py:125-224  subcarrier numbering / get_used_subcarrier_indexes
  pyphysim/modulators/ofdm.py:226-429  _prepare_input_signal, _add_CP, power scale, modulate
  pyphysim/modulators/ofdm.py:343-368,431-466  _remove_CP, demodulate
  pyphysim/modulators/ofdm.py:515-552  OfdmOneTapEqualizer.equalize_data
  pyphysim/channels/fading.py:482-536  dense tap expansion + get_freq_response
"""
import math

import numpy as np


def check_params(fft_size, cp_size, num_used=None):
    """ofdm.py:75-94; returns the resolved num_used."""
    if cp_size < 0 or cp_size > fft_size:
        raise ValueError("cp_size must be nonnegative and cannot be greater than fft_size")
    if num_used is None:
        num_used = fft_size
    if num_used > fft_size:
        raise ValueError("Number of used subcarriers cannot be greater than the fft_size")
    if num_used % 2 != 0 or num_used < 2:
        raise ValueError("Number of used subcarriers must be a multiple of 2")
    return num_used


def used_subcarrier_indexes(fft_size, num_used):
    """ofdm.py:188-224.  Full band: data k -> bin (k + N/2) mod N.  Otherwise the
    first half of the data rides the negative bins [N-h, N-1], the second half the
    positive bins [1, h]; DC and the band edges stay empty."""
    h = num_used // 2
    if num_used == fft_size:
        return np.concatenate([np.arange(fft_size // 2, fft_size), np.arange(0, fft_size // 2)])
    return np.concatenate([np.arange(fft_size - h, fft_size), np.arange(1, h + 1)])


def power_scale(fft_size, cp_size, num_used):
    """ofdm.py:370-392."""
    return float(fft_size) ** 2 / (float(num_used) + cp_size)


def num_ofdm_symbols(n, num_used):
    """ofdm.py:96-123."""
    return int(np.ceil(float(n) / num_used))


def modulate(x, fft_size, cp_size, num_used=None):
    """ofdm.py:394-429: zero-pad, map, sqrt(scale)*ifft (numpy's 1/N inside), prepend CP, flatten."""
    num_used = check_params(fft_size, cp_size, num_used)
    x = np.asarray(x, dtype=complex).reshape(-1)
    nsym = num_ofdm_symbols(x.size, num_used)
    padded = np.zeros(nsym * num_used, dtype=complex)
    padded[:x.size] = x
    bins = np.zeros((nsym, fft_size), dtype=complex)
    bins[:, used_subcarrier_indexes(fft_size, num_used)] = padded.reshape(nsym, num_used)
    t = math.sqrt(power_scale(fft_size, cp_size, num_used)) * np.fft.ifft(bins, fft_size, 1)
    if cp_size:
        t = np.hstack([t[:, -cp_size:], t])
    return t.reshape(-1)


def demodulate(y, fft_size, cp_size, num_used=None):
    """ofdm.py:431-466: strip CP, fft / sqrt(scale), gather the used bins, flatten."""
    num_used = check_params(fft_size, cp_size, num_used)
    y = np.asarray(y, dtype=complex).reshape(-1)
    nsym = y.size // (fft_size + cp_size)
    blocks = y[:nsym * (fft_size + cp_size)].reshape(nsym, fft_size + cp_size)[:, cp_size:]
    f = np.fft.fft(blocks, fft_size, 1) / math.sqrt(power_scale(fft_size, cp_size, num_used))
    return f[:, used_subcarrier_indexes(fft_size, num_used)].reshape(-1)


def dense_taps(tap_values_sparse, tap_indexes_sparse):
    """fading.py:482-511: scatter sparse taps [taps, ..., N] into [max_delay+1, ..., N]."""
    n_pad = int(tap_indexes_sparse[-1]) + 1
    out = np.zeros((n_pad,) + tap_values_sparse.shape[1:], dtype=complex)
    out[np.asarray(tap_indexes_sparse)] = tap_values_sparse
    return out


def freq_response(tap_values_sparse, tap_indexes_sparse, fft_size):
    """fading.py:513-536: FFT over the delay axis, one per time sample."""
    return np.fft.fft(dense_taps(tap_values_sparse, tap_indexes_sparse), fft_size, axis=0)


def onetap_equalize(data, tap_values_sparse, tap_indexes_sparse, fft_size, cp_size, num_used=None):
    """ofdm.py:515-552 literally: per-sample frequency response, mean over each
    OFDM symbol's samples (CP included), divide on the used bins."""
    num_used = check_params(fft_size, cp_size, num_used)
    data = np.asarray(data, dtype=complex)
    nsym = data.size // num_used
    fr = freq_response(tap_values_sparse, tap_indexes_sparse, fft_size)
    mean_fr = np.mean(fr.reshape(fft_size, nsym, -1), axis=2).T
    eq = data.reshape(-1, num_used) / mean_fr[:, used_subcarrier_indexes(fft_size, num_used)]
    return eq.reshape(-1)


def onetap_equalize_fast(data, tap_values_sparse, tap_indexes_sparse, fft_size, cp_size, num_used=None):
    """Same result via linearity (FFT of the per-symbol mean taps); this is the form
    the HIP kernel uses.  Checked equal to `onetap_equalize` to 1e-12 in the tests."""
    num_used = check_params(fft_size, cp_size, num_used)
    data = np.asarray(data, dtype=complex)
    nsym = data.size // num_used
    taps = np.asarray(tap_values_sparse)
    mean_taps = taps.reshape(taps.shape[0], nsym, -1).mean(axis=2)        # [taps, nsym]
    k = np.arange(fft_size).reshape(-1, 1, 1)
    d = np.asarray(tap_indexes_sparse).reshape(1, -1, 1)
    tw = np.exp(-2j * np.pi * k * d / fft_size)                           # [fft, taps, 1]
    mean_fr = (tw * mean_taps[None, :, :]).sum(axis=1).T                  # [nsym, fft]
    eq = data.reshape(-1, num_used) / mean_fr[:, used_subcarrier_indexes(fft_size, num_used)]
    return eq.reshape(-1)
