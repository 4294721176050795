"""CPU restatement (test infrastructure only) of the K-user channel's covariance matrices and SINRs.

Follows the reference, pyphysim/channels/multiuser.py:
    _from_small_matrix_to_big_matrix   :860-933
    big_H with path loss               :780-805
    calc_Q / calc_JP_Q                 :1314-1450   (external interference: :2530-2634)
    _calc_Bkl_cov_matrix_*             :1452-1826   (:2676-2742)
    _calc_SINR_k / calc_SINR / JP      :1828-2008   (:2636-2807)
    calc_cov_matrix_extint_*           :2469-2520
Pinned to the reference by oracle/make_golden.py (tests/golden/a14b_multiuser_stats.npz).
"""
import numpy as np


def pathloss_big(small, Nr, Nt):
    """small [Kr, Kt] -> [sum Nr, sum Nt]: entry (k, l) over the Nr[k] x Nt[l] block."""
    small = np.asarray(small, dtype=float)
    return np.vstack([np.hstack([np.full((int(Nr[k]), int(Nt[l])), small[k, l]) for l in range(small.shape[1])])
                      for k in range(small.shape[0])])


def effective_big_H(big_H, pl_big=None):
    return big_H if pl_big is None else big_H * np.sqrt(pl_big)


def _blocks(H, Nr, Nt, k):
    cr, ct = np.hstack([0, np.cumsum(Nr)]), np.hstack([0, np.cumsum(Nt)])
    n_tx = int(ct[-1])
    Hk = H[cr[k]:cr[k + 1], :]
    return Hk, [Hk[:, ct[l]:ct[l + 1]] for l in range(len(Nt))], Hk[:, :n_tx], Hk[:, n_tx:]


def cov_ext_plus_noise(H, Nr, Nt, k, pe=1.0, noise_var=None):
    _, _, _, ext = _blocks(H, Nr, Nt, k)
    R = pe * ext @ ext.conj().T
    if noise_var is not None:
        R = R + np.eye(int(Nr[k])) * noise_var
    return R


def calc_Q(H, Nr, Nt, k, F, noise_var=None, pe=1.0, joint=False):
    """Interference (other users) + external interference + noise covariance at receiver k."""
    _, Hkl, Hk_users, _ = _blocks(H, Nr, Nt, k)
    Q = np.zeros((int(Nr[k]), int(Nr[k])), dtype=complex)
    for l in range(len(Nt)):
        if l != k:
            G = (Hk_users if joint else Hkl[l]) @ F[l]
            Q = Q + G @ G.conj().T
    return Q + cov_ext_plus_noise(H, Nr, Nt, k, pe, noise_var)


def bkl_all_l(H, Nr, Nt, k, F, Rek, joint=False):
    """[first_part - second_part(l) for every stream l of user k]; Rek: the noise(+ext) covariance."""
    _, Hkl, Hk_users, _ = _blocks(H, Nr, Nt, k)
    first = np.zeros((int(Nr[k]), int(Nr[k])), dtype=complex)
    for j in range(len(Nt)):
        Hj = Hk_users if joint else Hkl[j]
        first = first + Hj @ (F[j] @ F[j].conj().T) @ Hj.conj().T
    first = first + Rek
    Hkk = Hk_users if joint else Hkl[k]
    out = []
    for l in range(F[k].shape[1]):
        v = F[k][:, l:l + 1]
        out.append(first - Hkk @ (v @ v.conj().T) @ Hkk.conj().T)
    return out


def calc_sinr(H, Nr, Nt, F, U, noise_var=None, pe=1.0, joint=False):
    """-> list over users of per-stream linear SINRs."""
    out = []
    for k in range(len(Nr)):
        _, Hkl, Hk_users, _ = _blocks(H, Nr, Nt, k)
        Rek = cov_ext_plus_noise(H, Nr, Nt, k, pe, noise_var if noise_var is not None else 0.0)
        B = bkl_all_l(H, Nr, Nt, k, F, Rek, joint)
        Hkk = Hk_users if joint else Hkl[k]
        s = np.empty(F[k].shape[1])
        for l in range(F[k].shape[1]):
            u = U[k][:, l:l + 1]
            aux = u.conj().T @ Hkk @ F[k][:, l:l + 1]
            num = (aux @ aux.conj().T).item()
            den = (u.conj().T @ B[l] @ u).item()
            s[l] = abs(num / den)
        out.append(s)
    return out
