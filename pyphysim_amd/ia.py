"""Mirror of the interference-alignment solvers of pyphysim.ia (reference ia/iabase.py:26-1019 and
ia/algorithms.py:42-2260), unit power per user.

* K = 3 users, 2x2 channels, one stream per user: `solve` runs `ia_closed_form` / `ia_iterative`
  (csrc/kernels_ia.hip, the register-resident kernels of config 5); precoders, receive filters, SINRs and iteration
  counts are the reference's entry by entry (tests/golden/c5_ia.npz, f3_ia_iterative.npz).
* any other geometry with K <= 4 users and Nr, Nt <= 4 (equal across users), per-user stream counts: the iterative
  solvers (alt-min, min-leakage, max-SINR) run `ia_solve_general` (csrc/kernels_ia_general.hip) from a 'random' /
  'fix' / 'svd' start, and `GreedStreamIASolver` / `BruteForceStreamIASolver` wrap them like the reference's
  (tests/golden/f3c_ia_general.npz).  Eigenvector phases are the kernel's, not LAPACK's: SINRs, capacity, F F^H
  and U^H U equal the reference's; the individual entries of F and U differ by one phase per stream.
Anything else raises ValueError with the reason.
"""
import numpy as np

from .engine import get_engine


class IASolverBaseClass:
    """iabase.py:26-1019 (the attributes a simulator reads)."""

    _SOLVER = None

    def __init__(self, multiUserChannel, engine=None):
        self._multiUserChannel = multiUserChannel
        self._engine = engine if engine is not None else getattr(multiUserChannel, "_engine", None)
        self._rs = np.random.RandomState()            # iabase.py:95
        self.clear()

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def clear(self):
        """iabase.py:100-125."""
        self._F = self._full_F = self._W_H = self._full_W_H = None
        self._sinr = self._capacity = None
        self._Ns = None
        self._P = None
        self._runned_iterations = 0

    K = property(lambda self: self._multiUserChannel.K)
    Nr = property(lambda self: self._multiUserChannel.Nr)
    Nt = property(lambda self: self._multiUserChannel.Nt)
    Ns = property(lambda self: self._Ns)
    noise_var = property(lambda self: self._multiUserChannel.noise_var or 0.0)

    @property
    def P(self):
        return np.ones(self.K) if self._P is None else self._P

    @P.setter
    def P(self, value):
        if value is None:
            self._P = None
            return
        value = np.ones(self.K) * value if np.isscalar(value) else np.asarray(value, dtype=float)
        if np.any(value <= 0):
            raise ValueError("P cannot be negative or equal to zero.")
        if not np.allclose(value, 1.0):
            raise ValueError("the GPU solvers run with unit power per user (P = 1)")
        self._P = value

    F = property(lambda self: self._F)
    full_F = property(lambda self: self._full_F)
    W_H = property(lambda self: self._W_H)

    @property
    def W(self):
        return None if self._W_H is None else np.array([w.conj().T for w in self._W_H] + [None], dtype=object)[:-1]

    def _get_channel(self, k, l):
        return self._multiUserChannel.get_Hkl(k, l)

    @staticmethod
    def _objs(items):
        return np.array(list(items) + [None], dtype=object)[:-1]

    def calc_SINR(self):
        """iabase.py:768-789: one array of per-stream SINRs (linear) per user.  After `solve` these are the solver
        kernel's own; after `set_precoders` / `set_receive_filters` they are evaluated by `k_mu_link_stats` from
        full_F and full_W_H (the reference's _calc_Bkl_cov_matrix_all_l / _calc_SINR_k, iabase.py:821-921)."""
        if self._sinr is None:
            if self._full_F is None or self.full_W_H is None:
                raise RuntimeError("no solution yet: call solve, or set the precoders and the receive filters")
            U = [np.asarray(u).conj().T for u in self.full_W_H]
            self._sinr = list(self._multiUserChannel.calc_SINR(self.full_F, U))
        return self._objs(np.atleast_1d(np.asarray(s, dtype=float)) for s in self._sinr)

    def calc_SINR_in_dB(self):
        """iabase.py:791-805."""
        return self._objs(10.0 * np.log10(s) for s in self.calc_SINR())

    def calc_sum_capacity(self):
        """iabase.py:807-819."""
        if self._capacity is None:
            return float(np.sum(np.log2(1.0 + np.hstack(list(self.calc_SINR())))))
        return float(self._capacity)

    def get_cost(self):
        """iabase.py:127-139: the base class has no cost."""
        return -1

    # ---- user-supplied solutions (iabase.py:167-327) ---------------------------------------------------------------
    def set_precoders(self, F=None, full_F=None, P=None):
        """iabase.py:202-250.  Powers other than 1 are accepted here (only the solver kernels need P = 1)."""
        if F is None and full_F is None:
            raise RuntimeError("Either 'F' or 'full_F' must be provided.")
        self._sinr = self._capacity = None
        if P is not None:
            self._P = np.ones(self.K) * P if np.isscalar(P) else np.asarray(P, dtype=float)
        if F is None:
            F = [np.asarray(f) / np.linalg.norm(np.asarray(f), "fro") for f in full_F]
        self._F = self._objs(np.asarray(f) for f in F)
        self._full_F = (self._objs(np.asarray(f) for f in full_F) if full_F is not None
                        else self._objs(f * np.sqrt(p) for f, p in zip(self._F, self.P)))
        self._Ns = np.array([f.shape[1] for f in self._F], dtype=int)

    def set_receive_filters(self, W_H=None, W=None):
        """iabase.py:329-361."""
        if W is None and W_H is None:
            raise RuntimeError("Either 'W' or 'W_H' must be provided.")
        if W is not None and W_H is not None:
            raise RuntimeError("Either 'W' or 'W_H' must be provided (but not both of them.")
        self._sinr = self._capacity = None
        self._full_W_H = None
        self._W_H = self._objs(np.asarray(w) for w in W_H) if W_H is not None else self._objs(np.asarray(w).conj().T for w in W)

    @property
    def full_W_H(self):
        """iabase.py:299-327: W_H scaled so that full_W_H H_kk full_F = I (solve(W_H H_kk full_F, W_H)); the two small
        products and the inverse run on the device (mcle_mimo_channel, mcle_pinv)."""
        if self._full_W_H is None and self._W_H is not None:
            out = []
            for k in range(self.K):
                eq = self.engine.mimo_channel(np.asarray(self._W_H[k])[None],
                                              self.engine.mimo_channel(np.asarray(self._get_channel(k, k))[None],
                                                                       np.asarray(self._full_F[k])[None])[0][None])[0]
                out.append(self.engine.mimo_channel(self.engine.pinv(eq)[None], np.asarray(self._W_H[k])[None])[0])
            self._full_W_H = self._objs(out)
        return self._full_W_H

    @property
    def full_W(self):
        """iabase.py:320-327."""
        fw = self.full_W_H
        return None if fw is None else self._objs(np.asarray(w).conj().T for w in fw)

    def calc_Q(self, k):
        """iabase.py:618-643: interference-plus-noise covariance at receiver k for full_F (device: k_mu_link_stats)."""
        return self._multiUserChannel.calc_Q(k, self.full_F)

    def calc_Q_rev(self, k):
        """iabase.py:645-672: the same in the reverse network (channels H_lk^H, the receive filters as precoders with
        power P_l, no noise term): the forward routine on the conjugate-transposed block matrix."""
        W = self.W
        if W is None:
            raise RuntimeError("the reverse-network covariance needs the receive filters")
        for l in range(self.K):
            if l != k and abs(np.linalg.norm(W[l], "fro") - 1.0) >= 1e-6:
                raise AssertionError("calc_Q_rev wants unit-norm receive filters (iabase.py:663)")
        big = np.asarray(self._multiUserChannel.big_H)
        n_tx = int(np.sum(self.Nt))
        res = self.engine.mu_link_stats(big[:, :n_tx].conj().T, self.Nt, self.Nr,
                                        F=[np.asarray(W[l]) * np.sqrt(self.P[l]) for l in range(self.K)], want=("Q",))
        return res["Q"][k][0]

    def calc_remaining_interference_percentage(self, k, Qk=None):
        """iabase.py:674-719: share of the interference power left in the Ns[k] least-interfered directions."""
        if Qk is None:
            Qk = self.calc_Q(k)
        ev = np.linalg.eigvalsh(np.asarray(Qk))              # ascending, like leig's sort on the real parts
        return float(np.sum(np.abs(ev[:int(self.Ns[k])])) / np.trace(np.abs(np.asarray(Qk))))

    def solve(self, Ns, P=None):
        raise NotImplementedError("solve: Not implemented")

    def _is_special(self, Ns_arr):
        return (self.K == 3 and list(self.Nr) == [2, 2, 2] and list(self.Nt) == [2, 2, 2]
                and list(Ns_arr) == [1, 1, 1])

    def _general_dims(self, Ns):
        """-> (Ns array, nr, nt) for the general kernel, or ValueError."""
        Ns_arr = np.ones(self.K, dtype=int) * Ns if isinstance(Ns, (int, np.integer)) else np.asarray(Ns, dtype=int)
        nr, nt = int(self.Nr[0]), int(self.Nt[0])
        if (self.K < 2 or self.K > 4 or any(int(v) != nr for v in self.Nr) or any(int(v) != nt for v in self.Nt)
                or nr > 6 or nt > 6 or len(Ns_arr) != self.K or np.any(Ns_arr < 1) or np.any(Ns_arr > min(nr, nt))):
            raise ValueError("the GPU interference-alignment kernels cover K <= 4 users with the same Nr, Nt <= 6 for "
                             "every user and 1 <= Ns <= min(Nr, Nt) (got K = %d, Nr = %s, Nt = %s, Ns = %s)"
                             % (self.K, list(self.Nr), list(self.Nt), list(Ns_arr)))
        return Ns_arr, nr, nt

    def _store_general(self, sol, nr, nt):
        ns = [int(n) for n in sol["Ns"][0]]
        self._Ns = np.array(ns, dtype=int)
        obj = lambda items: np.array(list(items) + [None], dtype=object)[:-1]
        self._full_F = obj(np.array(sol["F"][0, k][:nt, :ns[k]]) for k in range(self.K))
        self._F = obj(f / np.linalg.norm(f, "fro") for f in self._full_F)
        self._full_W_H = obj(np.array(sol["U"][0, k][:ns[k], :nr]) for k in range(self.K))
        self._W_H = obj(u / np.linalg.norm(u, "fro") for u in self._full_W_H)
        self._sinr = [np.array(sol["sinr"][0, k][:ns[k]]) for k in range(self.K)]
        self._capacity = sol["capacity"][0]

    def _check_geometry(self, Ns):
        Ns_arr = np.ones(self.K, dtype=int) * Ns if isinstance(Ns, (int, np.integer)) else np.asarray(Ns, dtype=int)
        if self.K != 3 or list(self.Nr) != [2, 2, 2] or list(self.Nt) != [2, 2, 2] or list(Ns_arr) != [1, 1, 1]:
            raise ValueError("the GPU interference-alignment kernels cover K = 3 users, 2x2 channels and one stream "
                             "per user (got K = %d, Nr = %s, Nt = %s, Ns = %s)" % (self.K, list(self.Nr),
                                                                                   list(self.Nt), list(Ns_arr)))
        return Ns_arr

    def _store(self, sol, Ns_arr):
        self._Ns = Ns_arr
        self._full_F = np.array([sol["F"][0, k].reshape(2, 1) for k in range(3)] + [None], dtype=object)[:-1]
        self._F = np.array([f / np.linalg.norm(f, "fro") for f in self._full_F] + [None], dtype=object)[:-1]
        self._full_W_H = np.array([sol["U"][0, k].reshape(1, 2) for k in range(3)] + [None], dtype=object)[:-1]
        self._W_H = np.array([u / np.linalg.norm(u) for u in self._full_W_H] + [None], dtype=object)[:-1]
        self._sinr = sol["sinr"][0]
        self._capacity = sol["capacity"][0]


class ClosedFormIASolver(IASolverBaseClass):
    """algorithms.py:42-265 (use_best_init = True, the only mode of the kernel)."""

    def __init__(self, multiUserChannel, use_best_init=True, engine=None):
        super().__init__(multiUserChannel, engine)
        if not use_best_init:
            raise ValueError("the closed-form kernel always evaluates both initialisations (use_best_init=True)")

    def solve(self, Ns, P=None):
        assert self.K == 3, "The ClosedFormIASolver class only works in a MIMO-IC scenario with 3 users."
        Ns_arr = self._check_geometry(Ns)
        self.P = P
        sol = self.engine.ia_closed_form(np.asarray(self._multiUserChannel.big_H)[np.newaxis], self.noise_var)
        self._store(sol, Ns_arr)


class IterativeIASolverBaseClass(IASolverBaseClass):
    """algorithms.py:271-883."""

    def __init__(self, multiUserChannel, engine=None):
        super().__init__(multiUserChannel, engine)
        self.max_iterations = 50
        self.relative_factor = 1e-6
        self._initialize_with = "random"
        self._alt_min_rs = np.random.RandomState()     # the embedded AlternatingMinIASolver's own RandomState

    @property
    def initialize_with(self):
        return self._initialize_with

    @initialize_with.setter
    def initialize_with(self, value):
        if value not in ("random", "alt_min", "closed_form", "fix", "svd"):
            raise RuntimeError("Invalid initialize_with value: '%s'" % (value,))
        if value == "alt_min" and self._SOLVER == "alt_min":
            raise RuntimeError("Can't use 'alt_min' initialization with '%s' class 'alt_min'" % type(self).__name__)
        self._initialize_with = value

    runned_iterations = property(lambda self: self._runned_iterations)

    def randomizeF(self, Ns, P=None):
        """iabase.py:511-545: F_k = normalized(randn_c_RS(rs, Nt, Ns))."""
        Ns_arr, nr, nt = self._general_dims(Ns)
        self.P = P
        self._runned_iterations = 0
        self._F = np.array([self._draw(self._rs, nt, int(Ns_arr[k])) for k in range(self.K)] + [None],
                           dtype=object)[:-1]
        self._full_F = self._F
        self._Ns = Ns_arr

    @staticmethod
    def _draw(rs, nt=2, ns=1):
        f = (1.0 / np.sqrt(2.0)) * (rs.randn(nt, ns) + 1j * rs.randn(nt, ns))
        return f / np.linalg.norm(f, "fro")

    _SELECT = None      # set by the stream-selection wrappers for the duration of their solve

    def _solve_general(self, Ns, P, select=None):
        """Geometries beyond K = 3, 2x2, one stream (algorithms.py:802-883 on csrc/kernels_ia_general.hip)."""
        Ns_arr, nr, nt = self._general_dims(Ns)
        if self._SOLVER not in ("alt_min", "min_leakage", "max_sinr"):
            raise ValueError("%s runs on the K = 3, 2x2, one-stream kernel only" % type(self).__name__)
        if self._SOLVER == "min_leakage" and np.any(Ns_arr > 1):
            # the reference asserts here too: calc_Q_rev wants unit-norm receive filters (iabase.py:663)
            raise ValueError("MinLeakageIASolver supports one stream per user (the reference's calc_Q_rev asserts "
                             "||W|| = 1, iabase.py:663)")
        self.P = P
        init = self._initialize_with
        F0 = None
        if select == "brute" or init == "svd":
            if nr != nt:
                raise ValueError("the 'svd' start is defined for Nr == Nt")
        elif init == "random":
            self.randomizeF(Ns_arr, P)
            F0 = self._F
        elif init == "fix":
            if self._F is None:
                raise RuntimeError("The precoder must be manually set, since you specified the 'fix' initialize_with "
                                   "option.")
            F0 = self._F
            Ns_arr = np.array([f.shape[1] for f in F0], dtype=int)
        else:
            raise ValueError("initialize_with = %r is available on the K = 3, 2x2, one-stream kernel only" % (init,))
        pad = None
        if F0 is not None:
            D = 4 if max(nr, nt) <= 4 else 6                    # the library's two matrix capacities
            pad = np.zeros((1, 4, D, D), dtype=complex)
            for k in range(self.K):
                f = np.asarray(F0[k])
                pad[0, k, :f.shape[0], :f.shape[1]] = f
        sol = self.engine.ia_solve_general(self._SOLVER, np.asarray(self._multiUserChannel.big_H), self.K, nr, nt,
                                           [int(n) for n in Ns_arr], self.noise_var, self.max_iterations,
                                           self.relative_factor, F_init=pad, select=select)
        self._store_general(sol, nr, nt)
        self._last_every_capacity = sol["every_capacity"][0] if "every_capacity" in sol else []
        self._runned_iterations = int(sol["iterations"][0])
        return self._runned_iterations

    def solve(self, Ns, P=None):
        """algorithms.py:802-883."""
        Ns_probe = np.ones(self.K, dtype=int) * Ns if isinstance(Ns, (int, np.integer)) else np.asarray(Ns, dtype=int)
        if not self._is_special(Ns_probe) or (self._initialize_with == "fix" and self._F is not None
                                               and any(np.asarray(f).shape != (2, 1) for f in self._F)):
            return self._solve_general(Ns, P)
        Ns_arr = self._check_geometry(Ns)
        self.P = P
        init = self._initialize_with
        if init == "random":
            self.randomizeF(Ns, P)
            F0 = self._F
        elif init == "fix":
            if self._F is None:
                raise RuntimeError("The precoder must be manually set, since you specified the 'fix' initialize_with "
                                   "option.")
            F0 = self._F
        elif init == "alt_min":
            F0 = [self._draw(self._alt_min_rs) for _ in range(3)]
        else:
            F0 = [np.zeros((2, 1), dtype=complex)] * 3
        F0 = np.stack([np.asarray(f).reshape(2) for f in F0])[np.newaxis]
        sol = self.engine.ia_iterative(self._SOLVER, np.asarray(self._multiUserChannel.big_H)[np.newaxis], F0,
                                       self.noise_var, self.max_iterations, self.relative_factor,
                                       "fix" if init in ("random", "fix") else init)
        self._store(sol, Ns_arr)
        self._runned_iterations = int(sol["iterations"][0])
        return self._runned_iterations


class AlternatingMinIASolver(IterativeIASolverBaseClass):
    """algorithms.py:885-1129."""
    _SOLVER = "alt_min"

    def get_cost(self):
        """algorithms.py:937-962: interference power outside the (Nr - Ns)-dimensional interference subspace C_k of every
        receiver.  The reference measures it against the C_k of its last iteration; here C_k is the dominant eigenspace
        of the final interference covariance (what that iteration converges to), i.e. the sum of the Ns_k smallest
        eigenvalues of the noise-free Q_k (device: k_mu_link_stats)."""
        big = np.asarray(self._multiUserChannel.big_H)
        Q = self.engine.mu_link_stats(big[:, :int(np.sum(self.Nt))], self.Nr, self.Nt, F=list(self.full_F), want=("Q",))["Q"]
        return float(sum(np.sum(np.abs(np.linalg.eigvalsh(Q[k][0])[:int(self.Ns[k])])) for k in range(self.K)))


class _LeakageCost:
    def get_cost(self):
        """algorithms.py:1146-1171: sum_k trace |W_k^H Q_k W_k| (Q_k from the device, W_k the unit-norm filters)."""
        W = self.W
        return float(sum(np.trace(np.abs(np.asarray(W[k]).conj().T @ self.calc_Q(k) @ np.asarray(W[k]))) for k in range(self.K)))


class MinLeakageIASolver(_LeakageCost, IterativeIASolverBaseClass):
    """algorithms.py:1132-1240."""
    _SOLVER = "min_leakage"


class MaxSinrIASolver(IterativeIASolverBaseClass):
    """algorithms.py:1243-1507."""
    _SOLVER = "max_sinr"


class MMSEIASolver(IterativeIASolverBaseClass):
    """algorithms.py:1510-1850."""
    _SOLVER = "mmse"


class GreedStreamIASolver:
    """algorithms.py:1853-2050: solve for the asked stream counts, then keep removing the stream with the worst SINR
    (re-solving from the remaining precoders) while the sum capacity grows.  The whole procedure runs inside one
    kernel launch (csrc/kernels_ia_general.hip, stream_selection = greedy); the wrapped solver ends up holding the
    best solution, as in the reference."""

    def __init__(self, iasolver_obj):
        if not isinstance(iasolver_obj, IterativeIASolverBaseClass):
            raise TypeError("iasolver_obj must be an iterative IA solver")
        self._iasolver = iasolver_obj
        self._runned_iterations = 0

    runned_iterations = property(lambda self: self._runned_iterations)

    def clear(self):
        self._iasolver.clear()
        self._runned_iterations = 0

    def solve(self, Ns, P=None):
        self._iasolver.clear()
        self._runned_iterations = self._iasolver._solve_general(Ns, P, select="greedy")
        self._iasolver._initialize_with = "fix"          # what the reference leaves behind (algorithms.py:1977)
        return self._runned_iterations

    def __getattr__(self, name):                         # F, W_H, Ns, calc_SINR ... of the wrapped solver
        return getattr(self._iasolver, name)


class BruteForceStreamIASolver:
    """algorithms.py:2057-2260: every combination of 1..Ns[k] streams per user from the 'svd' start, best sum
    capacity wins (one kernel launch, stream_selection = brute)."""

    def __init__(self, iasolver_obj):
        if not isinstance(iasolver_obj, IterativeIASolverBaseClass):
            raise TypeError("iasolver_obj must be an iterative IA solver")
        self._iasolver = iasolver_obj
        self._runned_iterations = 0
        self._stream_combinations, self._every_sum_capacity = [], []

    runned_iterations = property(lambda self: self._runned_iterations)
    stream_combinations = property(lambda self: self._stream_combinations)       # algorithms.py:2122-2132
    every_sum_capacity = property(lambda self: self._every_sum_capacity)         # :2134-2145

    def clear(self):
        self._iasolver.clear()
        self._runned_iterations = 0
        self._stream_combinations, self._every_sum_capacity = [], []

    def solve(self, Ns, P=None):
        import itertools
        self._iasolver.clear()
        self._iasolver._initialize_with = "svd"
        self._runned_iterations = self._iasolver._solve_general(Ns, P, select="brute")
        ns = [int(Ns)] * self._iasolver.K if isinstance(Ns, (int, np.integer)) else [int(n) for n in Ns]
        self._stream_combinations = list(itertools.product(*[range(1, n + 1) for n in ns]))     # :2176-2189
        self._every_sum_capacity = [float(c) for c in self._iasolver._last_every_capacity]
        return self._runned_iterations

    def __getattr__(self, name):
        return getattr(self._iasolver, name)
