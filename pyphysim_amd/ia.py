"""Mirror of the interference-alignment solvers of pyphysim.ia (reference ia/iabase.py:26-1019 and
ia/algorithms.py:42-1850) for the geometry the GPU kernels cover: K = 3 users, 2x2 channels, one stream per user,
unit power.  `solve` runs `ia_closed_form` / `ia_iterative` (csrc/kernels_ia.hip) on the channel of a
`multiuser.MultiUserChannelMatrix`; precoders, receive filters, SINRs and iteration counts are the reference's
(tests/golden/c5_ia.npz, f3_ia_iterative.npz).  Other geometries raise ValueError with the reason.
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
    full_W_H = property(lambda self: self._full_W_H)

    @property
    def W(self):
        return None if self._W_H is None else np.array([w.conj().T for w in self._W_H] + [None], dtype=object)[:-1]

    def _get_channel(self, k, l):
        return self._multiUserChannel.get_Hkl(k, l)

    def calc_SINR(self):
        """iabase.py:768-789: one array of per-stream SINRs (linear) per user."""
        return np.array([np.array([s]) for s in self._sinr] + [None], dtype=object)[:-1]

    def calc_sum_capacity(self):
        return float(self._capacity)

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
        Ns_arr = self._check_geometry(Ns)
        self.P = P
        self._runned_iterations = 0
        self._F = np.array([self._draw(self._rs) for _ in range(3)] + [None], dtype=object)[:-1]
        self._full_F = self._F
        self._Ns = Ns_arr

    @staticmethod
    def _draw(rs):
        f = (1.0 / np.sqrt(2.0)) * (rs.randn(2, 1) + 1j * rs.randn(2, 1))
        return f / np.linalg.norm(f, "fro")

    def solve(self, Ns, P=None):
        """algorithms.py:802-883."""
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


class MinLeakageIASolver(IterativeIASolverBaseClass):
    """algorithms.py:1132-1240."""
    _SOLVER = "min_leakage"


class MaxSinrIASolver(IterativeIASolverBaseClass):
    """algorithms.py:1243-1507."""
    _SOLVER = "max_sinr"


class MMSEIASolver(IterativeIASolverBaseClass):
    """algorithms.py:1510-1850."""
    _SOLVER = "mmse"
