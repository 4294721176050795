"""Oracle: Blast (V-BLAST spatial multiplexing) with ZF / MMSE receive filters.  TEST INFRASTRUCTURE.

NumPy complex128 restatement of (paths relative to the reference):
  pyphysim/mimo/mimo.py:264-309   _calcZeroForceFilter (pinv), _calcMMSEFilter (solve)
  pyphysim/mimo/mimo.py:577-607   Blast._calc_receive_filter (x sqrt(Nt); MMSE iff noise_var > 0)
  pyphysim/mimo/mimo.py:609-660   Blast.encode / decode (Fortran-order (de)interleave)
Quirk kept on purpose: the MMSE regulariser is sigma^2, not Nt*sigma^2 (mimo.py:307).
"""
import math

import numpy as np


def zf_filter(H):
    return np.linalg.pinv(H)


def mmse_filter(H, noise_var):
    Hh = H.conj().T
    return np.linalg.solve(Hh @ H + noise_var * np.eye(H.shape[1]), Hh)


def blast_receive_filter(H, noise_var):
    nt = H.shape[1]
    g = mmse_filter(H, noise_var) if noise_var > 0 else zf_filter(H)
    return g * math.sqrt(nt)


def blast_encode(x, nt):
    x = np.asarray(x)
    if x.size % nt != 0:
        raise ValueError("Input array number of elements must be a multiple of the number "
                         "of transmit antennas.")
    return x.reshape((nt, -1), order='F') / math.sqrt(nt)


def blast_decode(y, H, noise_var):
    return (blast_receive_filter(H, noise_var) @ y).reshape(-1, order='F')
