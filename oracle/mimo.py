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


# ---- GMDMimo (pyphysim/mimo/mimo.py:952-1067) with util.misc.gmd (pyphysim/util/misc.py:18-159) ----------
def gmd(U, S, V_H):
    """Geometric mean decomposition H = Q R P^H from an SVD: R upper triangular with the geometric mean of the
    singular values on its diagonal (Jiang, Hager, Li).  Planar rotations applied pair by pair."""
    m, n = U.shape[0], V_H.shape[0]
    p = int(np.sum(S >= 0.0))
    R = np.zeros((m, n))
    P = V_H.conj().T.copy()
    Q = U.copy()
    d = np.array(S, dtype=float)
    if p < 2:
        R[0, 0] = d[0]
    z = np.zeros(p - 1)
    large, small = 1, p - 1
    perm = np.arange(p)
    invperm = np.arange(p)
    sigma_bar = float(np.prod(S[0:p]) ** (1.0 / p))
    for k in range(p - 1):
        flag = False
        if d[k] >= sigma_bar:
            i = perm[small]
            small -= 1
            flag = d[i] >= sigma_bar
        else:
            i = perm[large]
            large += 1
            flag = d[i] <= sigma_bar
        k1 = k + 1
        if i != k1:
            d[k1], d[i] = d[i], d[k1]
            j = invperm[k1]
            perm[j] = i
            invperm[i] = j
            Q[:, [k1, i]] = Q[:, [i, k1]]
            P[:, [k1, i]] = P[:, [i, k1]]
        d1, d2 = d[k], d[k1]
        if flag:
            c, s = 1.0, 0.0
        else:
            c = math.sqrt((sigma_bar ** 2 - d2 ** 2) / (d1 ** 2 - d2 ** 2))
            s = math.sqrt(1 - c ** 2)
        d[k1] = d1 * d2 / sigma_bar
        z[k] = s * c * (d2 ** 2 - d1 ** 2) / sigma_bar
        R[k, k] = sigma_bar
        if k > 0:
            R[0:k, k] = z[0:k] * c
            z[0:k] = -z[0:k] * s
        P[:, [k, k1]] = P[:, [k, k1]] @ np.array([[c, -s], [s, c]])
        Q[:, [k, k1]] = Q[:, [k, k1]] @ ((1.0 / sigma_bar) * np.array([[c * d1, -s * d2], [s * d2, c * d1]]))
    R[p - 1, p - 1] = sigma_bar
    R[0:p - 1, p - 1] = z
    return Q, R, P


def scheme_filters(scheme, H):
    """(precoder W, receive filter G_H) of SVDMimo / GMDMimo for channel H with NumPy's (LAPACK's) singular vectors:
    mimo.py:846-890 (W = V / sqrt(Nt), G_H = diag(1/S) U^H sqrt(Nt)) and :965-1011 (W = P / sqrt(Nt), Blast's zero-forcing
    filter of Q R)."""
    nt = H.shape[1]
    U, S, V_H = np.linalg.svd(H)
    if scheme == 'svd':
        return V_H.conj().T / math.sqrt(nt), np.diag(1.0 / S) @ U.conj().T * math.sqrt(nt)
    Q, R, P = gmd(U, S, V_H)
    return P / math.sqrt(nt), blast_receive_filter(Q @ R, 0.0)


def gmd_encode(x, H):
    nt = H.shape[1]
    _, _, P = gmd(*np.linalg.svd(H))
    return (P / math.sqrt(nt)) @ np.asarray(x).reshape(nt, -1)


def gmd_decode(Y, H, noise_var=0.0):
    Q, R, _ = gmd(*np.linalg.svd(H))
    return (blast_receive_filter(Q @ R, noise_var) @ Y).reshape(-1)


def post_processing_linear_sinrs(channel, W, G_H, noise_var=0.0):
    """mimo/mimo.py:62-118 calc_post_processing_linear_SINRs: |E_ii|^2 / (|sum_{j != i} E_ij|^2 + nv ||G_H row i||^2),
    E = G_H channel W (note: the modulus of the SUMMED off-diagonal row entries, as the reference computes it)."""
    G_H = np.atleast_2d(np.asarray(G_H, dtype=complex))
    E = G_H @ (np.asarray(channel, dtype=complex) @ np.asarray(W, dtype=complex))
    s = np.diag(E)
    i = np.sum(E, axis=1) - s
    N = (noise_var or 0.0) * np.linalg.norm(G_H, axis=1) ** 2
    return np.abs(s) ** 2 / (np.abs(i) ** 2 + N)
