"""Oracle: Blast (V-BLAST spatial multiplexing) with ZF / MMSE receive filters.  TEST INFRASTRUCTURE.

NumPy complex128 restatement of (paths relative to the reference):
  pyphysim/mimo/mimo.py:264-309   _calcZeroForceFilter (pinv), _calcMMSEFilter (solve)
  pyphysim/mimo/mimo.py:577-607   Blast._calc_receive_filter (x sqrt(Nt); MMSE iff noise_var > 0)
  pyphysim/mimo/mimo.py:609-660   Blast.encode / decode (Fortran-order (de)interleave)
  pyphysim/mimo/mimo.py:666-783   MRT (phase-only precoder, MISO)
  pyphysim/mimo/mimo.py:833-946   SVDMimo (C-order reshape; V / sqrt(Nt), diag(1/S) U^H sqrt(Nt))
  pyphysim/mimo/mimo.py:1168-1269 Alamouti encode / decode
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


def alamouti_encode(x):
    x = np.asarray(x).reshape(-1)
    out = np.empty((2, x.size), dtype=complex)
    out[0, 0::2] = x[0::2]
    out[0, 1::2] = -x[1::2].conjugate()
    out[1, 0::2] = x[1::2]
    out[1, 1::2] = x[0::2].conjugate()
    return out / math.sqrt(2)


def alamouti_decode(Y, H):
    h0, h1 = H[:, 0], H[:, 1]
    out = np.empty(Y.shape[1], dtype=complex)
    out[0::2] = h0.conjugate() @ Y[:, 0::2] + h1 @ Y[:, 1::2].conjugate()
    out[1::2] = h1.conjugate() @ Y[:, 0::2] + (-h0) @ Y[:, 1::2].conjugate()
    out /= np.linalg.norm(H, 'fro') ** 2
    return out * math.sqrt(2)


def mrt_encode(x, h):
    h = np.asarray(h).reshape(1, -1)
    W = np.exp(-1j * np.angle(h)).T / math.sqrt(h.shape[1])
    return W * np.asarray(x).reshape(1, -1)


def mrt_decode(y, h):
    h = np.asarray(h).reshape(1, -1)
    return (math.sqrt(h.shape[1]) / np.sum(np.abs(h)) * np.asarray(y)).reshape(-1)


def svd_encode(x, H):
    nt = H.shape[1]
    _, _, Vh = np.linalg.svd(H)
    return (Vh.conj().T / math.sqrt(nt)) @ np.asarray(x).reshape(nt, -1)


def svd_decode(Y, H):
    nt = H.shape[1]
    U, S, _ = np.linalg.svd(H)
    return ((np.diag(1.0 / S) @ U.conj().T * math.sqrt(nt)) @ Y).reshape(-1)
