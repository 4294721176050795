"""Oracle: closed-form interference alignment for the 3-user MIMO interference channel and the
multi-user channel it runs on.  TEST INFRASTRUCTURE.

NumPy complex128 restatement of (paths relative to the reference):
  pyphysim/channels/multiuser.py:1003-1044   MultiUserChannelMatrix.randomize (big_H = randn_c)
  pyphysim/channels/multiuser.py:1179-1262   corrupt_concatenated_data / corrupt_data
  pyphysim/ia/algorithms.py:73-96            ClosedFormIASolver._calc_E
  pyphysim/ia/algorithms.py:98-191           _calc_all_F_initializations, _updateF, _updateW
  pyphysim/ia/algorithms.py:194-265          solve (best of all eigenvector subsets by sum capacity)
  pyphysim/ia/iabase.py:188-200,299-327      full_F, full_W_H (equivalent channel normalised to 1)
  pyphysim/ia/iabase.py:768-789,897-996      calc_SINR via the B_kl covariance matrices
  pyphysim/util/misc.py:210-255              leig (eigenvectors of the n smallest eigenvalues)
Uses numpy.linalg (LAPACK) exactly where the reference does, so the LAPACK eigenvector
normalisation (unit 2-norm, largest component real and positive) carries over.
"""
import itertools

import numpy as np


def split_blocks(big_H, K, nr, nt):
    return [[big_H[k * nr:(k + 1) * nr, l * nt:(l + 1) * nt] for l in range(K)] for k in range(K)]


def leig(A, n):
    D, V = np.linalg.eig(A)
    order = np.argsort(D.real)
    return V[:, order[0:n]], D[order[0:n]]


def calc_E(H):
    return np.linalg.solve(H[2][0], H[2][1]).dot(np.linalg.solve(H[0][1], H[0][2]).dot(np.linalg.solve(H[1][2], H[1][0])))


def update_F(H, F0):
    F = [F0, np.linalg.pinv(H[2][1]) @ (H[2][0] @ F0), np.linalg.pinv(H[1][2]) @ (H[1][0] @ F0)]
    return [f / np.linalg.norm(f, "fro") for f in F]


def update_W(H, F, Ns):
    A0 = H[0][1] @ F[1]
    A1 = H[1][0] @ F[0]
    A2 = H[2][0] @ F[0]
    return [leig(A @ A.conj().T, Ns)[0] for A in (A0, A1, A2)]


def full_W_H(H, F, W):
    out = []
    for k in range(3):
        Wh = W[k].conj().T
        out.append(np.linalg.solve(Wh @ (H[k][k] @ F[k]), Wh))
    return out


def calc_SINR(H, F, U, noise_var):
    """iabase.py:897-996 with P = 1 (full_F = F)."""
    sinr = []
    for k in range(3):
        first = 0.0
        for j in range(3):
            a = H[k][j] @ F[j]
            first = first + a @ a.conj().T
        s_k = np.empty(F[k].shape[1])
        for l in range(F[k].shape[1]):
            v = F[k][:, l:l + 1]
            a = H[k][k] @ v
            B = first - a @ a.conj().T + noise_var * np.eye(H[k][k].shape[0])
            u_h = U[k][l:l + 1, :]
            num = u_h @ (H[k][k] @ v)
            num = num @ num.conj().T
            den = u_h @ (B @ u_h.conj().T)
            s_k[l] = np.abs(num.item() / den.item())
        sinr.append(s_k)
    return sinr


def closed_form_solve(H, Ns, noise_var, use_best_init=True):
    """-> (F list, U = full_W_H list, sum capacity, SINRs)."""
    vecs = np.linalg.eig(calc_E(H))[1]
    if use_best_init:
        subsets = [vecs[:, c] for c in itertools.combinations(range(vecs.shape[1]), Ns)]
    else:
        subsets = [vecs[:, 0:Ns]]
    best = None
    best_cap = 0
    for F0 in subsets:
        F = update_F(H, np.asarray(F0).reshape(vecs.shape[0], Ns))
        W = update_W(H, F, Ns)
        U = full_W_H(H, F, W)
        sinr = calc_SINR(H, F, U, noise_var)
        cap = float(np.sum([np.sum(np.log2(1 + s)) for s in sinr]))
        if cap > best_cap or not use_best_init:
            best_cap, best = cap, (F, U, cap, sinr)
    return best


def mu_corrupt(big_H, X, noise, noise_var):
    """multiuser.py:1206-1213: big_H @ vstack(X) + sqrt(noise_var) * noise."""
    return big_H @ X + np.sqrt(noise_var) * noise
