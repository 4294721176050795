"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the reference's block diagonalisation (SURVEY.md section 8(f).3 tail):
    pyphysim/comm/waterfilling.py:15-92            doWF
    pyphysim/comm/blockdiagonalization.py:272-363  _calc_BD_matrix_no_power_scaling
                                         :365-401  global water-filling power scaling
                                         :403-464  normalised power scaling
                                         :466-508  block_diagonalize
                                         :510-566  block_diagonalize_no_waterfilling
                                         :568-585  calc_receive_filter (pinv)
    pyphysim/util/misc.py:590-662                  least_right_singular_vectors

Pinned against the reference in this container by oracle/make_golden.py -> tests/golden/f6_block_diag.npz
(tests/test_oracle_golden.py).

`block_diagonalize` follows the reference's own arithmetic (two numpy SVDs per user), so its singular
vectors carry LAPACK's phases.  `block_diagonalize_closed` is the formulation the HIP kernels use (one
inverse + one small Hermitian eigen-decomposition per user, canonical phases); both give the same
precoder up to a unit-modulus factor per column, which `canonical_columns` removes.
"""
import numpy as np


def waterfilling(gains, total_power, noise_var=1.0, Es=1.0):
    """waterfilling.py:15-92.  gains = channel POWER gains; returns (powers, water level)."""
    gains = np.asarray(gains, dtype=float)
    order = np.argsort(gains)[::-1]
    g = gains[order]
    n, removed = g.size, 0
    mu_min = float(noise_var) / (Es * g[n - 1])
    Ps = mu_min - float(noise_var) / (Es * g[:n])
    while sum(Ps) > total_power and removed < n:
        removed += 1
        mu_min = float(noise_var) / (Es * g[n - removed - 1])
        Ps = mu_min - float(noise_var) / (Es * g[:n - removed])
    kept = n - removed
    P_kept = (total_power - np.sum(Ps)) / kept + Ps
    P = np.zeros(n)
    P[order[:kept]] = P_kept
    return P, P_kept[0] + float(noise_var) / g[0]


def least_right_singular_vectors(A, n):
    """misc.py:590-662: (V0 = the n least right singular vectors, V1 = the rest, S of the rest), every
    group in ASCENDING singular-value order (numpy's order reversed)."""
    _, S, V_H = np.linalg.svd(A, full_matrices=True)
    V = V_H.conj().T
    rev = list(reversed(range(V.shape[0])))
    return V[:, rev[:n]], V[:, rev[n:]], S[rev[n:]]


def _user_rows(K, nr_total, user):
    r = nr_total // K
    return list(range(r * user, r * (user + 1)))


def bd_no_power_scaling(H, K):
    """blockdiagonalization.py:272-363 -> (Ms_bad [Nt, sum streams], Sigma [sum streams])."""
    H = np.asarray(H)
    nr = H.shape[0]
    assert nr % K == 0
    r = nr // K
    Ms, Sigma = [], []
    for user in range(K):
        others = [i for u in range(K) if u != user for i in _user_rows(K, nr, u)]
        tilde_H = H[others, :]
        n_streams = nr - np.linalg.matrix_rank(tilde_H)
        tilde_V0 = least_right_singular_vectors(tilde_H, n_streams)[0]
        _, V1, S = least_right_singular_vectors(H[_user_rows(K, nr, user), :] @ tilde_V0, r - n_streams)
        Ms.append(tilde_V0 @ V1)
        Sigma.extend(S)
    return np.hstack(Ms), np.array(Sigma)


def global_waterfilling_scaling(Ms_bad, Sigma, K, iPu, noise_var):
    """:365-401."""
    P = waterfilling(Sigma ** 2, K * iPu, noise_var)[0]
    return Ms_bad @ np.diag(np.sqrt(P))


def normalized_waterfilling_scaling(Ms_bad, Sigma, K, iPu, noise_var):
    """:403-464: global water-filling, then one common factor so the strongest user block meets iPu."""
    n_u = Sigma.size // K
    Ms = global_waterfilling_scaling(Ms_bad, Sigma, K, iPu, noise_var)
    max_sqrt_P = 0
    for user in range(K):
        cur = np.linalg.norm(Ms[:, user * n_u:(user + 1) * n_u], 'fro')
        if cur > max_sqrt_P:
            max_sqrt_P = cur
    return Ms * np.sqrt(iPu) / max_sqrt_P


def block_diagonalize(H, K, iPu, noise_var):
    """:466-508 -> (newH, Ms_good)."""
    Ms_bad, Sigma = bd_no_power_scaling(H, K)
    Ms = normalized_waterfilling_scaling(Ms_bad, Sigma, K, iPu, noise_var)
    return H @ Ms, Ms


def block_diagonalize_no_waterfilling(H, K, iPu):
    """:510-566: each user's block scaled to Frobenius norm sqrt(iPu)."""
    H = np.asarray(H)
    n_u = H.shape[1] // K
    Ms_bad, _ = bd_no_power_scaling(H, K)
    Ms = np.empty(Ms_bad.shape, dtype=complex)
    for user in range(K):
        blk = Ms_bad[:, user * n_u:(user + 1) * n_u]
        Ms[:, user * n_u:(user + 1) * n_u] = blk * np.sqrt(iPu) / np.linalg.norm(blk, 'fro')
    return H @ Ms, Ms


def calc_receive_filter(newH):
    """:568-585."""
    return np.linalg.pinv(newH)


# ---------------------------------------------------------------------------------------------
# the kernels' formulation (square channel: Nt == K * r)
# ---------------------------------------------------------------------------------------------
def canonical_columns(M):
    """Rotate every column so that its largest-magnitude entry is real and positive (first one wins ties):
    the representative of the per-column phase freedom that singular vectors have."""
    M = np.array(M, dtype=complex)
    for j in range(M.shape[1]):
        i = int(np.argmax(np.abs(M[:, j])))
        if abs(M[i, j]) > 0:
            M[:, j] *= np.conj(M[i, j]) / abs(M[i, j])
    return M


def bd_no_power_scaling_closed(H, K):
    """Same (Ms_bad, Sigma) as bd_no_power_scaling up to column phases, for square H, without any SVD:
    with Z = H^-1 and Z_k its columns of user k, the Gram block D_k = Z_k^H Z_k = [(H H^H)^-1]_kk is the
    inverse of the Schur complement H_k P0_k H_k^H (P0_k = projector on the null space of the other users'
    rows), so D_k = U diag(1/sigma^2) U^H gives the singular values of the equivalent channel H_k V0_k and
    its right singular vectors are V_k = Z_k U diag(sigma); columns in ascending sigma like the reference,
    canonical phase per column."""
    H = np.asarray(H, dtype=complex)
    n = H.shape[0]
    assert H.shape[1] == n and n % K == 0
    r = n // K
    Z = np.linalg.inv(H)
    Ms, Sigma = [], []
    for k in range(K):
        Zk = Z[:, k * r:(k + 1) * r]
        lam, U = np.linalg.eigh(Zk.conj().T @ Zk)       # ascending lam == descending sigma
        sig = 1.0 / np.sqrt(lam[::-1])                   # ascending sigma
        V = Zk @ U[:, ::-1] * sig[None, :]
        Ms.append(canonical_columns(V))
        Sigma.extend(sig)
    return np.hstack(Ms), np.array(Sigma)


def block_diagonalize_closed(H, K, iPu, noise_var, waterfill=True):
    """(newH, Ms, W) the way the kernels compute them; W = pinv(newH) built stream by stream."""
    H = np.asarray(H, dtype=complex)
    n = H.shape[0]
    r = n // K
    Ms_bad, Sigma = bd_no_power_scaling_closed(H, K)
    if waterfill:
        Ms = normalized_waterfilling_scaling(Ms_bad, Sigma, K, iPu, noise_var)
    else:
        Ms = np.hstack([Ms_bad[:, k * r:(k + 1) * r] * np.sqrt(iPu) / np.linalg.norm(Ms_bad[:, k * r:(k + 1) * r])
                        for k in range(K)])
    newH = H @ Ms
    # inside a user's block the columns of newH are orthogonal (H_k V_k = U Sigma), so pinv is the scaled
    # conjugate transpose; a switched-off stream (zero column) gets the zero row numpy's pinv returns
    W = np.zeros((n, n), dtype=complex)
    for k in range(K):
        for j in range(k * r, (k + 1) * r):
            b = newH[k * r:(k + 1) * r, j]
            n2 = float(np.sum(np.abs(b) ** 2))
            if n2 > 0.0:
                W[j, k * r:(k + 1) * r] = np.conj(b) / n2
    return newH, Ms, W


# ---------------------------------------------------------------------------------------------
# block diagonalisation with external interference (comm/blockdiagonalization.py:666-1469)
#   BDWithExtIntBase.calc_whitening_matrices :690-720, WhiteningBD :722-836, EnhancedBD :839-1469
#   channels/multiuser.py:2469-2520 calc_cov_matrix_extint_(without|plus)_noise
#   util/misc.py:1167-1200 calc_whitening_matrix, subspace/projections.py:96-130 calcProjectionMatrix
# A channel with external interference is the plain multi-user channel plus extra COLUMNS (the interferers'
# antennas): big_H = [big_H_no_ext_int | H_ext].
# ---------------------------------------------------------------------------------------------
def cov_extint_plus_noise(big_H, K, nr, n_tx_total, pe, noise_var):
    """R_k = pe H_ext,k H_ext,k^H + noise_var I for every user (multiuser.py:2469-2520)."""
    out = []
    for k in range(K):
        ext = big_H[k * nr:(k + 1) * nr, n_tx_total:]
        R = pe * ext @ ext.conj().T
        if noise_var is not None:
            R = R + np.eye(nr) * noise_var
        out.append(R)
    return out


def whitening_matrix(cov):
    """misc.py:1167-1200: W = V diag(L^-1/2) from eig(cov); W^H cov W = I."""
    L, V = np.linalg.eig(cov)
    return V @ np.diag(1.0 / (L ** 0.5))


def whitening_bd(big_H, K, nr, nt, iPu, noise_var, pe):
    """WhiteningBD.block_diagonalize_no_waterfilling (:781-836) -> (Ms per user [K nt, nt], W per user [nt.., nr])."""
    n_tx = K * nt
    H = big_H[:, :n_tx]
    R = cov_extint_plus_noise(big_H, K, nr, n_tx, pe, noise_var)
    from scipy.linalg import block_diag
    big_wf = block_diag(*[whitening_matrix(R[k]).conj().T for k in range(K)])
    newH, Ms = block_diagonalize_no_waterfilling(big_wf @ H, K, iPu)
    big_W = calc_receive_filter(newH) @ big_wf
    Ms_k = [Ms[:, k * nt:(k + 1) * nt] for k in range(K)]
    W_k = [big_W[k * nt:(k + 1) * nt, k * nr:(k + 1) * nr] for k in range(K)]
    return Ms_k, W_k, [nt] * K


def projection_matrix(A):
    Ah = A.conj().T
    return A @ np.linalg.inv(Ah @ A) @ Ah


def stream_reduction_matrix(Re_k, kept):
    """:120-145: the `kept` right singular vectors of Re_k with the SMALLEST singular values."""
    return least_right_singular_vectors(Re_k, kept)[0]


def ebd_receive_filter(Heq_red, P=None):
    """EnhancedBD.calc_receive_filter_user_k (:1056-1099)."""
    if P is None:
        return np.linalg.pinv(Heq_red)
    Pb = projection_matrix(P)
    return np.linalg.pinv(Pb @ Heq_red) @ Pb


def ebd_linear_sinrs(Heq_red, W, Re_k):
    """EnhancedBD._calc_linear_SINRs (:1101-1138)."""
    mtP = W @ Heq_red
    desired = np.abs(np.diagonal(mtP)) ** 2
    internal = np.sum(np.abs(mtP - np.diagflat(np.diagonal(mtP))) ** 2, 1)
    ext = np.diagonal(W @ Re_k @ W.conj().T).real
    return desired / (internal + np.abs(ext))


def enhanced_bd(big_H, K, nr, nt, iPu, noise_var, pe, metric=None, num_streams=None, metric_func=None, Ms_bad=None):
    """EnhancedBD.block_diagonalize_no_waterfilling (:1413-1469) for metric in (None, 'naive', 'fixed', 'capacity',
    'effective_throughput' [metric_func(sinrs) supplied by the caller]).
    -> (MsPk per user [K nt, Ns_k], W per user [Ns_k, nr], Ns per user).
    Ms_bad: the unit-norm BD directions to reduce (default: the reference's own, from numpy's SVDs).  The reduction
    MsPk = Ms_k Pk mixes the COLUMNS of Ms_k, whose phases are whatever the SVD returned -- so a stream-reduced
    solution is only defined relative to a particular Ms_bad (true of the reference itself, which inherits LAPACK's
    phases); tests hand the kernel's Ms_bad in here to pin the reduction arithmetic."""
    n_tx = K * nt
    H = big_H[:, :n_tx]
    if metric is None:
        newH, Ms = block_diagonalize_no_waterfilling(H, K, iPu)
        Ms_k = [Ms[:, k * nt:(k + 1) * nt] for k in range(K)]
        W_k = [np.linalg.pinv(newH[k * nr:(k + 1) * nr, k * nt:(k + 1) * nt]) for k in range(K)]
        return Ms_k, W_k, [nt] * K
    Re = cov_extint_plus_noise(big_H, K, nr, n_tx, pe, noise_var)
    if Ms_bad is None:
        Ms_bad, _ = bd_no_power_scaling(H, K)
    Ms_out, W_out, Ns_out = [], [], []
    for k in range(K):
        Msk = Ms_bad[:, k * nt:(k + 1) * nt]
        Heq = H[k * nr:(k + 1) * nr, :] @ Msk
        if metric in ('naive', 'fixed'):
            cands = [num_streams]
        else:
            cands = list(range(1, nt + 1))
        best = None
        for ns in cands:
            if metric == 'naive':
                Pk = np.eye(nt)[:, 0:ns]
            elif metric == 'fixed' or ns < nt:
                Pk = stream_reduction_matrix(Re[k], ns)
            else:
                Pk = np.eye(nt)
            norm_term = np.linalg.norm(Msk @ Pk, 'fro') / np.sqrt(iPu)
            Heq_red = Heq @ (Pk / norm_term)
            W = ebd_receive_filter(Heq_red, Pk)
            if metric in ('naive', 'fixed'):
                value = 0.0
            else:
                sinrs = ebd_linear_sinrs(Heq_red, W, Re[k])
                value = float(np.sum(np.log2(1 + sinrs))) if metric == 'capacity' else float(metric_func(sinrs))
            if best is None or value > best[0]:                 # np.argmax: first maximum
                best = (value, Msk @ Pk / norm_term, W, Pk.shape[1])
        Ms_out.append(best[1])
        W_out.append(best[2])
        Ns_out.append(best[3])
    return Ms_out, W_out, Ns_out
