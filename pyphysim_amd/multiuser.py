"""Mirror of pyphysim.channels.multiuser.MultiUserChannelMatrix (reference channels/multiuser.py:586-1262, the part
the interference-alignment path uses; path loss, external interference and the OFDM variants are out of scope).

The channel and noise draws come from the object's own NumPy ``RandomState``s exactly as in the reference
(:670-709, :1036-1038, :1206-1210), so ``set_channel_seed`` / ``set_noise_seed`` reproduce reference runs; the
arithmetic ``big_H @ X + noise`` runs in `k_mimo_channel` on the GPU.
"""
import math

import numpy as np

from .engine import get_engine


def _randn_c_rs(rs, *shape):
    """util/misc.py:327-355 with an explicit RandomState: real block first, then the imaginary block."""
    return (1.0 / math.sqrt(2.0)) * (rs.randn(*shape) + 1j * rs.randn(*shape))


class MultiUserChannelMatrix:
    def __init__(self, engine=None, dtype="f64"):
        self._engine, self.dtype = engine, dtype
        self._big_H = np.zeros((0, 0), dtype=complex)
        self._Nr = np.array([], dtype=int)
        self._Nt = np.array([], dtype=int)
        self._K = 0
        self._RS_channel = np.random.RandomState()
        self._RS_noise = np.random.RandomState()
        self._last_noise = None
        self._noise_var = None
        self._W = None
        self._big_W = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    # ---- seeds (multiuser.py:670-709) ----------------------------------------------------------------
    def set_channel_seed(self, seed=None):
        self._RS_channel.seed(seed=seed)

    def set_noise_seed(self, seed=None):
        self._RS_noise.seed(seed)

    def re_seed(self):
        self.set_channel_seed(None)
        self.set_noise_seed(None)

    # ---- properties -------------------------------------------------------------------------------------
    Nr = property(lambda self: self._Nr)
    Nt = property(lambda self: self._Nt)
    K = property(lambda self: self._K)
    big_H = property(lambda self: self._big_H)
    last_noise = property(lambda self: self._last_noise)
    pathloss = property(lambda self: None)

    @property
    def H(self):
        out = np.zeros((self._K, self._K), dtype=np.ndarray)
        for k in range(self._K):
            for l in range(self._K):
                out[k, l] = self.get_Hkl(k, l)
        return out

    @property
    def noise_var(self):
        return self._noise_var

    @noise_var.setter
    def noise_var(self, value):
        if value is not None and value < 0.0:
            raise ValueError("noise_var must be non-negative or None")
        self._noise_var = value

    # ---- construction -----------------------------------------------------------------------------------
    def init_from_channel_matrix(self, channel_matrix, Nr, Nt, K):
        """multiuser.py:935-995."""
        Nr_array = np.ones(K, dtype=int) * Nr if isinstance(Nr, (int, np.integer)) else np.asarray(Nr, dtype=int)
        Nt_array = np.ones(K, dtype=int) * Nt if isinstance(Nt, (int, np.integer)) else np.asarray(Nt, dtype=int)
        channel_matrix = np.asarray(channel_matrix)
        if channel_matrix.shape != (int(np.sum(Nr_array)), int(np.sum(Nt_array))):
            raise ValueError("Shape of the channel_matrix must be equal to the sum or receive antennas of all users "
                             "times the sum of the receive antennas of all users.")
        if Nt_array.size != K or Nr_array.size != K:
            raise ValueError("K must be equal to the number of elements in Nr and Nt")
        self._K, self._Nr, self._Nt = int(K), Nr_array, Nt_array
        self._big_H = np.array(channel_matrix, dtype=complex)
        self._big_H.setflags(write=False)

    def randomize(self, Nr, Nt, K):
        """multiuser.py:1003-1044: big_H = randn_c_RS(RS_channel, sum(Nr), sum(Nt))."""
        Nr = np.ones(K, dtype=int) * Nr if isinstance(Nr, (int, np.integer)) else np.asarray(Nr)
        Nt = np.ones(K, dtype=int) * Nt if isinstance(Nt, (int, np.integer)) else np.asarray(Nt)
        self._Nr, self._Nt, self._K = Nr.astype(int), Nt.astype(int), int(K)
        self._big_H = _randn_c_rs(self._RS_channel, int(np.sum(self._Nr)), int(np.sum(self._Nt)))
        self._big_H.setflags(write=False)

    def get_Hkl(self, k, l):
        """Channel from transmitter l to receiver k (multiuser.py:1046-1089)."""
        r0, t0 = int(np.sum(self._Nr[:k])), int(np.sum(self._Nt[:l]))
        return self._big_H[r0:r0 + self._Nr[k], t0:t0 + self._Nt[l]]

    def get_Hk(self, k):
        r0 = int(np.sum(self._Nr[:k]))
        return self._big_H[r0:r0 + self._Nr[k], :]

    # ---- post-processing filters (multiuser.py:1135-1177) --------------------------------------------
    def set_post_filter(self, filters):
        self._W = list(filters)
        self._big_W = None

    W = property(lambda self: self._W)

    @property
    def big_W(self):
        if self._big_W is None and self._W is not None:
            rows, cols = sum(w.shape[0] for w in self._W), sum(w.shape[1] for w in self._W)
            big = np.zeros((rows, cols), dtype=complex)
            r = c = 0
            for w in self._W:
                big[r:r + w.shape[0], c:c + w.shape[1]] = w
                r += w.shape[0]
                c += w.shape[1]
            self._big_W = big
        return self._big_W

    # ---- the channel itself -------------------------------------------------------------------------------
    def corrupt_concatenated_data(self, data):
        """multiuser.py:1179-1221: big_H @ data (+ sqrt(noise_var) * randn_c_RS(RS_noise, shape))."""
        data = np.asarray(data)
        if data.ndim == 1:
            data = data.reshape(-1, 1)
        noise = None
        nv = 0.0
        if self._noise_var is not None:
            noise = _randn_c_rs(self._RS_noise, int(np.sum(self._Nr)), data.shape[1])
            nv = float(self._noise_var)
            self._last_noise = noise * math.sqrt(nv)
        else:
            self._last_noise = None
        out = self.engine.mimo_channel(np.asarray(self._big_H)[np.newaxis], data[np.newaxis],
                                       None if noise is None else noise[np.newaxis], nv, dtype=self.dtype)[0]
        if self.big_W is not None:
            out = np.dot(self.big_W.conjugate().T, out)
        return out

    def corrupt_data(self, data):
        """multiuser.py:1223-1262: list of per-transmitter arrays in, array of per-receiver arrays out."""
        out = self.corrupt_concatenated_data(np.vstack(list(data)))
        res = np.zeros(self._K, dtype=np.ndarray)
        cum = np.hstack([0, np.cumsum(self._Nr)])
        for k in range(self._K):
            res[k] = out[cum[k]:cum[k + 1], :]
        return res


class MultiUserChannelMatrixExtInt(MultiUserChannelMatrix):
    """reference channels/multiuser.py:2011-2520 (what the block-diagonalisation variants with external interference
    read): the plain multi-user channel plus the columns of `extIntK` external interferers with `extIntNt` antennas
    each (they have no receive antennas, so big_H is [sum Nr, sum Nt + sum extIntNt])."""

    def __init__(self, engine=None, dtype="f64"):
        super().__init__(engine, dtype)
        self._extIntK = 0
        self._extIntNt = np.array([], dtype=int)

    extIntK = property(lambda self: self._extIntK)
    extIntNt = property(lambda self: self._extIntNt)
    K = property(lambda self: self._K - self._extIntK)
    Nr = property(lambda self: self._Nr[:self._K - self._extIntK])
    Nt = property(lambda self: self._Nt[:self._K - self._extIntK])

    @property
    def big_H_no_ext_int(self):
        """multiuser.py:2097-2122."""
        return self._big_H[:, :int(np.sum(self.Nt))]

    @staticmethod
    def _prepare(Nr, Nt, K, NtE):
        NtE = np.atleast_1d(np.asarray(NtE, dtype=int)) if not isinstance(NtE, (int, np.integer)) else np.array([NtE])
        Nr = np.ones(K, dtype=int) * Nr if isinstance(Nr, (int, np.integer)) else np.asarray(Nr, dtype=int)
        Nt = np.ones(K, dtype=int) * Nt if isinstance(Nt, (int, np.integer)) else np.asarray(Nt, dtype=int)
        return (np.hstack([Nr, np.zeros(NtE.size, dtype=int)]), np.hstack([Nt, NtE]), K + NtE.size, NtE.size, NtE)

    def randomize(self, Nr, Nt, K, NtE):
        """multiuser.py:2379-2420: one randn_c draw of the full [sum Nr, sum Nt + sum NtE] matrix."""
        full_Nr, full_Nt, full_K, self._extIntK, self._extIntNt = self._prepare(Nr, Nt, K, NtE)
        MultiUserChannelMatrix.randomize(self, full_Nr, full_Nt, full_K)

    def init_from_channel_matrix(self, channel_matrix, Nr, Nt, K, NtE):
        """multiuser.py:2335-2377."""
        full_Nr, full_Nt, full_K, self._extIntK, self._extIntNt = self._prepare(Nr, Nt, K, NtE)
        MultiUserChannelMatrix.init_from_channel_matrix(self, channel_matrix, full_Nr, full_Nt, full_K)

    def calc_cov_matrix_extint_without_noise(self, pe=1.0):
        """multiuser.py:2469-2494."""
        n_tx = int(np.sum(self.Nt))
        out = np.empty(self.K, dtype=np.ndarray)
        cum = np.hstack([0, np.cumsum(self.Nr)])
        for k in range(self.K):
            ext = self._big_H[cum[k]:cum[k + 1], n_tx:]
            out[k] = pe * np.dot(ext, ext.conj().T)
        return out

    def calc_cov_matrix_extint_plus_noise(self, pe=1.0):
        """multiuser.py:2496-2520."""
        out = self.calc_cov_matrix_extint_without_noise(pe)
        if self.noise_var is not None:
            for k in range(self.K):
                out[k] = out[k] + np.eye(int(self.Nr[k])) * self.noise_var
        return out
