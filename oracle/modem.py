"""Oracle: constellations, modulate / demodulate, error counting.  TEST INFRASTRUCTURE.

NumPy complex128 / int64 restatement of (paths relative to the reference):
  pyphysim/modulators/fundamental.py:175-248  Modulator.modulate / demodulate
  pyphysim/modulators/fundamental.py:396-448  PSK (Gray permutation, 1e-15 snap)
  pyphysim/modulators/fundamental.py:534-541,605-647  BPSK
  pyphysim/modulators/fundamental.py:659-777  QAM grid + Gray index matrix
  pyphysim/util/conversion.py:229-279         binary2gray / gray2binary
  pyphysim/util/misc.py:392-414,449-476,519-566  level2bits, count_bits, count_bit_errors
"""
import math

import numpy as np


def binary2gray(num):
    """conversion.py:229-249: g = b ^ (b >> 1)."""
    num = np.asarray(num)
    return (num >> 1) ^ num


def gray2binary(num):
    """conversion.py:252-279: prefix-xor fold over 16 bits."""
    t = np.asarray(num)
    t = t ^ (t >> 8)
    t = t ^ (t >> 4)
    t = t ^ (t >> 2)
    t = t ^ (t >> 1)
    return t


def level2bits(n):
    """misc.py:392-414: bits needed for n levels (n >= 1)."""
    if n < 1:
        raise ValueError("level2bits: n must be greater then one")
    return max(int(n - 1).bit_length(), 1)


def qam_constellation(M):
    """fundamental.py:659-777.  Label (r << k/2 | c) sits at grid row gray(r), col gray(c)."""
    power = math.log(M, 2)
    if (power % 2 != 0) or (2 ** power != M):
        raise ValueError("M must be a square power of 2")
    L = int(round(math.sqrt(M)))
    grid = np.empty(M, dtype=complex)
    for ii in range(L):          # row: imaginary part from +(L-1) downwards
        for jj in range(L):      # column: real part from -(L-1) upwards
            grid[ii * L + jj] = complex(-(L - 1) + 2 * jj, (L - 1) - 2 * ii)
    grid = grid / math.sqrt((M - 1) * 2.0 / 3.0)
    g = binary2gray(np.arange(L, dtype=int))
    half_bits = level2bits(L * L) // 2
    index_matrix = (g.reshape(L, 1) << half_bits) + g.reshape(1, L)
    return grid[index_matrix.reshape(-1)]


def psk_constellation(M, phase_offset=0.0):
    """fundamental.py:396-448 (Gray mapped; |re|,|im| < 1e-15 snapped to 0)."""
    assert 2 ** math.log(M, 2) == M
    ph = 2.0 * math.pi / M * np.arange(0, M) + phase_offset
    re = np.cos(ph)
    im = np.sin(ph)
    re[np.abs(re) < 1e-15] = 0
    im[np.abs(im) < 1e-15] = 0
    return (re + 1j * im)[gray2binary(np.arange(0, M))]


def bpsk_constellation():
    """fundamental.py:537-541: 0 -> +1, 1 -> -1."""
    return np.array([1.0 + 0j, -1.0 + 0j])


def modulate(symbols, idx):
    """fundamental.py:196-199: table gather; IndexError -> ValueError."""
    try:
        return symbols[idx]
    except IndexError:
        raise ValueError("Input data must be between 0 and 2^M")


def demodulate(symbols, rx):
    """fundamental.py:241-246: argmin_m |c_m - r| (first minimum), shape preserved."""
    rx = np.asarray(rx)
    flat = rx.reshape(-1)
    out = np.empty(flat.size, dtype=np.int64)
    step = max(1, (1 << 22) // symbols.size)      # bound the M x N temporary
    col = symbols.reshape(-1, 1)
    for s in range(0, flat.size, step):
        out[s:s + step] = np.abs(col - flat[s:s + step]).argmin(axis=0)
    return out.reshape(rx.shape)


def qam_slicer(M, rx):
    """Per-axis rounding demapper, decision-identical to `demodulate` for square QAM
    away from exact ties; cross-check for the HIP fast path."""
    L = int(round(math.sqrt(M)))
    half_bits = level2bits(M) // 2
    scale = math.sqrt((M - 1) * 2.0 / 3.0)
    rx = np.asarray(rx)
    jj = np.clip(np.floor((rx.real * scale + (L - 1)) * 0.5 + 0.5), 0, L - 1).astype(np.int64)
    ii = np.clip(np.floor(((L - 1) - rx.imag * scale) * 0.5 + 0.5), 0, L - 1).astype(np.int64)
    return (gray2binary(ii) << half_bits) | gray2binary(jj)


def count_bits(n):
    """misc.py:449-476 popcount (non-negative ints)."""
    n = np.asarray(n).astype(np.uint64)
    c = np.zeros(n.shape, dtype=np.int64)
    while np.any(n):
        c += (n & np.uint64(1)).astype(np.int64)
        n = n >> np.uint64(1)
    return c


def count_bit_errors(first, second, axis=None):
    """misc.py:519-566: sum popcount(first xor second)."""
    return np.sum(count_bits(np.bitwise_xor(first, second)), axis)


def count_symbol_errors(first, second):
    """User code of the reference (apps/awgn_modulators/simulate_psk.py:83-86)."""
    return int(np.sum(np.asarray(first) != np.asarray(second)))


def dB2Linear(dB):
    """util/conversion.py:139-158."""
    return 10.0 ** (np.asarray(dB, dtype=float) / 10.0)
