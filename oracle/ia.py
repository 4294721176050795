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


def closed_form_solve(H, Ns, noise_var, use_best_init=True, with_W=False):
    """-> (F list, U = full_W_H list, sum capacity, SINRs) [+ W list when with_W]."""
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
            best_cap, best = cap, ((F, U, cap, sinr, W) if with_W else (F, U, cap, sinr))
    return best


def mu_corrupt(big_H, X, noise, noise_var):
    """multiuser.py:1206-1213: big_H @ vstack(X) + sqrt(noise_var) * noise."""
    return big_H @ X + np.sqrt(noise_var) * noise


# ---- iterative solvers -----------------------------------------------------
# pyphysim/ia/algorithms.py:271-883   IterativeIASolverBaseClass (solve loop, _is_diff_significant)
# pyphysim/ia/algorithms.py:885-1129  AlternatingMinIASolver (_updateC / _updateF / _updateW)
# pyphysim/ia/algorithms.py:1132-1240 MinLeakageIASolver
# pyphysim/ia/algorithms.py:1243-1507 MaxSinrIASolver
# pyphysim/ia/iabase.py:600-667       calc_Q / calc_Q_rev;  channels/multiuser.py:1345-1382 (noise term of Q)
# pyphysim/util/misc.py:161-255       peig / leig
# P = 1 for every user (apps/ia/simulate_ia.py:120 calls solve(Ns) without P), so full_F = F.
def peig(A, n):
    D, V = np.linalg.eig(A)
    order = np.argsort(D.real)[::-1]
    return V[:, order[0:n]], D[order[0:n]]


def _calc_Q(H, F, k, noise_var):
    """interference covariance at receiver k (+ noise_var I: multiuser.py:1376-1380)."""
    K = len(F)
    Q = np.zeros((H[k][k].shape[0],) * 2, dtype=complex)
    for l in range(K):
        if l != k:
            a = H[k][l] @ F[l]
            Q = Q + a @ a.conj().T
    if noise_var is not None:
        Q = Q + np.eye(Q.shape[0]) * noise_var
    return Q


def _calc_Q_rev(H, W, k):
    K = len(W)
    Q = np.zeros((H[k][k].shape[1],) * 2, dtype=complex)
    for l in range(K):
        if l != k:
            a = H[l][k].conj().T @ W[l]
            Q = Q + a @ a.conj().T
    return Q


def is_diff_significant(F_old, F_new, relative_factor):
    for fo, fn in zip(F_old, F_new):
        if np.abs(fn - fo).max() > np.abs(fn).min() * relative_factor:
            return True
    return False


def _iterate(F, step, max_iterations, relative_factor, state=None):
    """algorithms.py:857-869; `state` = receive filters handed over by an initialisation other than 'random'."""
    old_F = F
    runned = 0
    for _ in range(max_iterations):
        runned += 1
        F, state = step(F, state)
        if not is_diff_significant(old_F, F, relative_factor):
            break
        old_F = F
    return F, state, runned


def alt_min_solve(H, F_init, noise_var, max_iterations=50, relative_factor=1e-6, W_init=None, full_F_first=None):
    """-> (F, W_H rows, runned_iterations); Ns taken from F_init.  (W_init is ignored: the algorithm derives
    its state C from F in _before_initialize_W_func and W only when it finishes.)"""
    K = len(F_init)
    Ns = [f.shape[1] for f in F_init]
    Nr = [H[k][k].shape[0] for k in range(K)]

    def update_C(F):
        return [peig(_calc_Q(H, F, k, noise_var), Nr[k] - Ns[k])[0] for k in range(K)]

    def step(F, C):
        if C is None:
            # _before_initialize_W_func of the initialisation (calc_Q reads full_F: see general_solve's greedy branch)
            C = update_C(F if full_F_first is None else full_F_first)
        Y = [np.eye(Nr[k], dtype=complex) - C[k] @ C[k].conj().T for k in range(K)]
        newF = [0] * K
        for (l, k) in itertools.permutations(range(K), 2):
            newF[l] = newF[l] + H[k][l].conj().T @ Y[k] @ H[k][l]
        F = []
        for k in range(K):
            f = leig(newF[k], Ns[k])[0]
            F.append(f / np.linalg.norm(f, "fro"))
        return F, update_C(F)

    F0 = [np.asarray(f, dtype=complex) for f in F_init]
    F, C, runned = _iterate(F0, step, max_iterations, relative_factor)
    if C is None:
        C = update_C(F)
    W_H = [np.linalg.inv(np.hstack([H[k][k] @ F[k], C[k]]))[0:Ns[k]] for k in range(K)]
    return F, W_H, runned


def min_leakage_solve(H, F_init, noise_var, max_iterations=50, relative_factor=1e-6, W_init=None, full_F_first=None):
    K = len(F_init)
    Ns = [f.shape[1] for f in F_init]

    def update_W(F):
        return [leig(_calc_Q(H, F, k, noise_var), Ns[k])[0] for k in range(K)]

    def step(F, W):
        if W is None:
            W = update_W(F if full_F_first is None else full_F_first)
        F = [leig(_calc_Q_rev(H, W, k), Ns[k])[0] for k in range(K)]
        return F, update_W(F)

    F0 = [np.asarray(f, dtype=complex) for f in F_init]
    F, W, runned = _iterate(F0, step, max_iterations, relative_factor, W_init)
    if W is None:
        W = update_W(F)
    return F, [w.conj().T for w in W], runned


def max_sinr_solve(H, F_init, noise_var, max_iterations=50, relative_factor=1e-6, W_init=None, full_F_first=None):
    """Ns = 1 per user is what the kernel covers; the restatement keeps the per-stream loop."""
    K = len(F_init)

    def calc_U(Hkk_of, V, chan, per_stream_power, V_cov=None):
        """chan(k, j): channel seen by 'receiver' k from 'transmitter' j in the (possibly reversed) network.
        per_stream_power: the reverse network weighs every transmitter's term by P/Ns (algorithms.py:1290-1345),
        the forward one uses full_F as it is (iabase.py:828-894)."""
        out = []
        for k in range(K):
            first = 0.0
            Vc = V if V_cov is None else V_cov            # the covariances read full_F, the directions F
            for j in range(K):
                a = chan(k, j) @ Vc[j]
                first = first + (a @ a.conj().T) * (1.0 / V[j].shape[1] if per_stream_power else 1.0)
            Hkk = Hkk_of(k)
            U = np.zeros((Hkk.shape[0], V[k].shape[1]), dtype=complex)
            for l in range(V[k].shape[1]):
                v = V[k][:, l:l + 1]
                a = Hkk @ Vc[k][:, l:l + 1]
                second = (a @ a.conj().T) * (1.0 / V[k].shape[1] if per_stream_power else 1.0)
                B = first - second + noise_var * np.eye(Hkk.shape[0])
                u = np.linalg.solve(B, Hkk @ v)
                U[:, l] = (u / np.linalg.norm(u, "fro"))[:, 0]
            out.append(U / np.linalg.norm(U, "fro"))
        return out

    fwd = lambda k, j: H[k][j]
    rev = lambda k, j: H[j][k].conj().T

    def update_W(F, F_cov=None):
        return calc_U(lambda k: H[k][k], F, fwd, False, F_cov)

    def step(F, W):
        if W is None:
            W = update_W(F, full_F_first)
        F = calc_U(lambda k: H[k][k].conj().T, W, rev, True)
        return F, update_W(F)

    F0 = [np.asarray(f, dtype=complex) for f in F_init]
    F, W, runned = _iterate(F0, step, max_iterations, relative_factor, W_init)
    if W is None:
        W = update_W(F)
    return F, [w.conj().T for w in W], runned


ITERATIVE = {"alt_min": alt_min_solve, "min_leakage": min_leakage_solve, "max_sinr": max_sinr_solve}


def iterative_solve(algo, H, F_init, noise_var, max_iterations=50, relative_factor=1e-6, initialize_with='random'):
    """-> (F, U = full_W_H, sum capacity, SINRs, runned_iterations).  initialize_with (algorithms.py:633-663):
    'random' / 'fix': F_init are the starting precoders; 'closed_form' (:572-597): start from the closed-form
    solution's F and W; 'alt_min' (:599-632): run the alternating-minimisation solver first (its own random
    start = F_init, the same max_iterations) and start from its F and its normalised receive filters; 'svd'
    (:503-547): the most significant right singular vector(s) of every user's direct channel."""
    W_init = None
    if initialize_with == 'svd':
        Ns = F_init[0].shape[1]
        F_init = []
        for k in range(len(H)):
            V = np.linalg.svd(H[k][k], full_matrices=True)[2].conj().T
            rev = list(reversed(range(V.shape[0])))          # least_right_singular_vectors, misc.py:647-660
            V1 = V[:, rev[H[k][k].shape[0] - Ns:]]
            F_init.append(V1 / np.linalg.norm(V1, 'fro'))
    if initialize_with == 'closed_form':
        Ns = F_init[0].shape[1]
        F_init, _, _, _, W_init = closed_form_solve(H, Ns, noise_var, True, with_W=True)
    elif initialize_with == 'alt_min':
        if algo == 'alt_min':
            raise RuntimeError("Can't use 'alt_min' initialization with 'AlternatingMinIASolver' class 'alt_min'")
        Fa, WHa, _ = alt_min_solve(H, F_init, noise_var, max_iterations, relative_factor)
        F_init = Fa
        W_init = [w.conj().T / np.linalg.norm(w, 'fro') for w in WHa]
    F, W_H, runned = ITERATIVE[algo](H, F_init, noise_var, max_iterations, relative_factor, W_init)
    U = [np.linalg.solve(W_H[k] @ (H[k][k] @ F[k]), W_H[k]) for k in range(len(F))]
    sinr = calc_SINR(H, F, U, noise_var)
    cap = float(np.sum([np.sum(np.log2(1 + s)) for s in sinr]))
    return F, U, cap, sinr, runned


# ---- MMSEIASolver (pyphysim/ia/algorithms.py:1510-1850) -----------------------------------------------------
def mmse_solve(H, F_init, noise_var, max_iterations=50, relative_factor=1e-6, W_init=None):
    """-> (full_F, W_H rows, runned_iterations).  Precoders V_i = (sum_k H_ki^H U_k U_k^H H_ki + mu_i I)^-1 H_ii^H U_i
    with the Lagrange multiplier mu_i >= 0 found by scipy.optimize.newton (secant) as in the reference
    (:1660-1825); receive filters U_k = (sum_i H_ki V_i V_i^H H_ki^H + sigma^2 I)^-1 H_kk V_k (:1560-1600).
    The iteration test compares the NORMALISED precoders (_F), transmission uses full_F = V."""
    from scipy import optimize
    K = len(F_init)
    P = np.ones(K)

    def calc_U(full_F):
        out = []
        for k in range(K):
            acc = 0
            for i in range(K):
                a = H[k][i] @ full_F[i]
                acc = acc + a @ a.conj().T
            out.append(np.linalg.solve(acc + noise_var * np.eye(H[k][k].shape[0]), H[k][k] @ full_F[k]))
        return out

    def v_for_mu(sum_term, mu, HhU):
        return np.linalg.solve(sum_term + mu * np.eye(sum_term.shape[0]), HhU)

    def calc_V(W, i):
        HhU = H[i][i].conj().T @ W[i]
        sum_term = np.array([0.0])
        for k in range(K):
            a = H[k][i].conj().T @ W[k]
            sum_term = sum_term + a @ a.conj().T
        S = np.linalg.svd(sum_term)[1]
        if S.max() / S.min() > 5e4:
            sum_term = sum_term + np.eye(sum_term.shape[0]) * (S.mean() / 100.0)

        def func(mu, st, hu, p):
            return np.linalg.norm(v_for_mu(st, mu, hu), 'fro') ** 2 - p
        scale = np.linalg.norm(HhU)
        HhU = HhU / scale
        sum_term = sum_term / scale
        if func(0.0, sum_term, HhU, P[i]) <= 0:
            return v_for_mu(sum_term, 0.0, HhU)
        mu = optimize.newton(func, 0.0, args=(sum_term, HhU, P[i]), maxiter=200)
        if abs(mu) > 1e20:
            mu = optimize.newton(func, 0.0, args=(sum_term * 10, HhU * 10, P[i]), maxiter=200) / 10.0
        return v_for_mu(sum_term, mu, HhU)

    full_F = [np.asarray(f, dtype=complex) for f in F_init]
    W = W_init if W_init is not None else calc_U(full_F)
    F = [f / np.linalg.norm(f, 'fro') for f in full_F]
    old_F, runned = F, 0
    for _ in range(max_iterations):
        runned += 1
        full_F = [calc_V(W, i) for i in range(K)]
        F = [v / np.linalg.norm(v, 'fro') for v in full_F]
        W = calc_U(full_F)
        if not is_diff_significant(old_F, F, relative_factor):
            break
        old_F = F
    return full_F, [w.conj().T for w in W], runned


ITERATIVE["mmse"] = mmse_solve


# ---- general geometries: per-user stream counts, 'svd' start, greedy / brute-force stream selection -------------
# pyphysim/ia/algorithms.py:503-547 (_initialize_with_svd), :1853-2075 GreedStreamIASolver, :2057-2260
# BruteForceStreamIASolver.  P = 1 for every user.
def svd_init(H, Ns):
    """Most significant right singular vectors of every direct channel, in the column order of
    util/misc.py:647-660 least_right_singular_vectors' reversed index list."""
    F = []
    for k in range(len(H)):
        V = np.linalg.svd(H[k][k], full_matrices=True)[2].conj().T
        rev = list(reversed(range(V.shape[0])))
        V1 = V[:, rev[H[k][k].shape[0] - Ns[k]:]]
        F.append(V1 / np.linalg.norm(V1, 'fro'))
    return F


def evaluate(H, F, W_H, noise_var):
    """-> (U = full_W_H, per-user SINR arrays, sum capacity) of a precoder / receive-filter set."""
    U = [np.linalg.solve(W_H[k] @ (H[k][k] @ F[k]), W_H[k]) for k in range(len(F))]
    sinr = calc_SINR(H, F, U, noise_var)
    return U, sinr, float(np.sum([np.sum(np.log2(1 + s)) for s in sinr]))


def principal_components(A, n):
    """util/misc.py:870-905 get_principal_component_matrix."""
    U, S, V_H = np.linalg.svd(A)
    newS = np.zeros(U.shape[0], dtype=A.dtype)
    newS[:n] = S[:n]
    newS = np.diag(newS)[:, :V_H.shape[1]]
    return U @ (newS @ V_H[:, :n])


def solve_finalize(F, W_H):
    """algorithms.py:665-735 _solve_finalize: a precoder with more than one stream whose condition number exceeds
    1e4 has dead dimensions (an over-loaded stream allocation collapses onto fewer streams): keep the singular
    directions above max/1e4 in the precoder (renormalised) and in the receive filter.  -> (F, W_H, Ns)"""
    F, W_H = [np.array(f) for f in F], [np.array(w) for w in W_H]
    for k in range(len(F)):
        if F[k].shape[1] > 1:
            S = np.linalg.svd(F[k])[1]
            if S.max() / S.min() > 1e4:
                n = int(np.count_nonzero(S > S.max() / 1.0e4))
                f = principal_components(F[k], n)
                F[k] = f / np.linalg.norm(f, 'fro')
                W_H[k] = principal_components(W_H[k].conj().T, n).conj().T
    return F, W_H, [f.shape[1] for f in F]


def general_solve(algo, H, Ns, noise_var, max_iterations=50, relative_factor=1e-6, F_init=None, select=None):
    """-> dict(F, U, sinr, cap, Ns, runned).  F_init None = 'svd' start; select in (None, 'greedy', 'brute').
    `runned` follows the reference's bookkeeping: the brute-force wrapper clears the solver before every run and
    adds the runs up; the greedy wrapper re-solves with initialize_with = 'fix', which does NOT reset the solver's
    counter, and adds the solver's (cumulative) return values (algorithms.py:1936, 1979)."""
    K = len(H)
    Ns = [int(Ns)] * K if np.isscalar(Ns) else [int(n) for n in Ns]
    run = ITERATIVE[algo]

    def solve(F0, full_F_first=None):
        F, W_H, runned = run(H, F0, noise_var, max_iterations, relative_factor, None, full_F_first)
        F, W_H, ns = solve_finalize(F, W_H)
        U, sinr, cap = evaluate(H, F, W_H, noise_var)
        return dict(F=F, U=U, sinr=sinr, cap=cap, Ns=ns), runned

    if select == 'brute':
        best, total, every, combos = None, 0, [], []
        for comb in itertools.product(*[range(1, n + 1) for n in Ns]):
            sol, runned = solve(svd_init(H, comb))
            total += runned
            every.append(sol["cap"])                # every_sum_capacity / stream_combinations (:2122-2145)
            combos.append(list(comb))
            if best is None or sol["cap"] > best["cap"]:
                best = sol
        best["runned"] = total
        best["every_sum_capacity"], best["stream_combinations"] = every, combos
        return best
    sol, counter = solve(svd_init(H, Ns) if F_init is None else [np.asarray(f, dtype=complex) for f in F_init])
    total = counter
    while select == 'greedy' and any(n > 1 for n in sol["Ns"]):
        old = sol
        sinr, ns = sol["sinr"], sol["Ns"]
        mins = [int(np.argmin(s)) for s in sinr]
        order = [int(i) for i in np.argsort([sinr[i][mins[i]] for i in range(K)]) if ns[i] > 1]
        user, stream = order[0], mins[order[0]]
        # the wrapper deletes the column from F AND from full_F but renormalises only F (algorithms.py:1962-1975);
        # the re-solve's first receive-filter update builds its covariances from that stale full_F
        F = [f.copy() for f in sol["F"]]
        full_F = [f.copy() for f in sol["F"]]
        F[user] = np.delete(F[user], stream, 1)
        full_F[user] = F[user].copy()
        F[user] = F[user] / np.linalg.norm(F[user], 'fro')
        sol, runned = solve(F, full_F)
        counter += runned
        total += counter
        if old["cap"] > sol["cap"]:
            sol = old
            break
    sol = dict(sol)
    sol["runned"] = total
    return sol
