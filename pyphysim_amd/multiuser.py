"""Mirror of pyphysim.channels.multiuser.MultiUserChannelMatrix / MultiUserChannelMatrixExtInt (reference
channels/multiuser.py:586-2807): the K-user block channel with path loss, its covariance matrices (calc_Q, calc_JP_Q,
_calc_Bkl_cov_matrix_*) and post-filter SINRs (calc_SINR, calc_JP_SINR), with and without external interference.

The channel and noise draws come from the object's own NumPy ``RandomState``s exactly as in the reference
(:670-709, :1036-1038, :1206-1210), so ``set_channel_seed`` / ``set_noise_seed`` reproduce reference runs; the
arithmetic runs on the GPU: ``big_H @ X + noise`` in `k_mimo_channel`, everything covariance- or SINR-shaped in
`k_mu_link_stats` (csrc/kernels_multiuser.hip).
"""
import math

import numpy as np

from . import channels as _channels
from .engine import get_engine


class MuChannel:
    """reference channels/multiuser.py:42-440: a grid of independent single-user links, `su[rx, tx]`, every one a
    SuChannel with its own fading generator cloned from the one given (`get_similar_fading_generator`, rx-major -- the
    order the generators draw in under a seeded NumPy state), one shared channel profile, an optional path loss per
    link; receiver rx hears sum_tx su[rx, tx](signal[tx]).  Every link is a device convolution (`k_tdl_apply` /
    `k_tdl_apply_mimo`), the sum over transmitters is a device add."""

    def __init__(self, N, fading_generator=None, channel_profile=None, tap_powers_dB=None, tap_delays=None, Ts=None,
                 engine=None, dtype=None):
        if fading_generator is None:
            fading_generator = _channels.RayleighSampleGenerator()
        num_rx, num_tx = (N if isinstance(N, (tuple, list)) else (N, N))
        self._engine = engine if engine is not None else getattr(fading_generator, "_engine", None)
        self.dtype = dtype if dtype is not None else getattr(fading_generator, "dtype", None)
        self._su_siso_channels = np.empty((num_rx, num_tx), dtype=object)
        for rx in range(num_rx):
            for tx in range(num_tx):
                gen = fading_generator.get_similar_fading_generator()
                self._su_siso_channels[rx, tx] = _channels.SuChannel(gen, channel_profile=channel_profile,
                                                                     tap_powers_dB=tap_powers_dB, tap_delays=tap_delays,
                                                                     Ts=Ts, engine=self._engine, dtype=self.dtype)
                channel_profile = self._su_siso_channels[rx, tx].channel_profile      # one profile object for every link
        self._pathloss_matrix = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def __repr__(self):
        if self._su_siso_channels.size:
            return "<%s.%s object at %s>\n%s" % (self.__class__.__module__, self.__class__.__name__, hex(id(self)),
                                                 repr(self.channel_profile))
        return "<%s.%s object at %s>" % (self.__class__.__module__, self.__class__.__name__, hex(id(self)))

    @property
    def switched_direction(self):
        return self._su_siso_channels[0, 0].switched_direction

    @switched_direction.setter
    def switched_direction(self, value):
        for link in self._su_siso_channels.flat:
            link.switched_direction = value

    num_tx_antennas = property(lambda self: self._su_siso_channels[0, 0].num_tx_antennas)
    num_rx_antennas = property(lambda self: self._su_siso_channels[0, 0].num_rx_antennas)
    channel_profile = property(lambda self: self._su_siso_channels[0, 0].channel_profile)
    num_taps = property(lambda self: self._su_siso_channels[0, 0].num_taps)
    num_taps_with_padding = property(lambda self: self._su_siso_channels[0, 0].num_taps_with_padding)
    pathloss_matrix = property(lambda self: self._pathloss_matrix)

    def set_pathloss(self, pathloss_matrix):
        """:256-290 (a matrix of LINEAR power ratios, one per link; None switches the path loss off)."""
        num_rx, num_tx = self._su_siso_channels.shape
        self._pathloss_matrix = None if pathloss_matrix is None else np.copy(pathloss_matrix)
        for rx in range(num_rx):
            for tx in range(num_tx):
                self._su_siso_channels[rx, tx].set_pathloss(None if pathloss_matrix is None else pathloss_matrix[rx, tx])

    def _through_every_link(self, signal, send):
        links = self._su_siso_channels.T if self.switched_direction else self._su_siso_channels
        num_rx, num_tx = links.shape
        signal = np.asarray(signal) if not isinstance(signal, np.ndarray) or signal.dtype != object else signal
        if num_tx == 1 and getattr(signal, "ndim", 2) == 1 and signal.dtype != object:
            signal = np.reshape(signal, (1, -1))
        outputs = np.empty(num_rx, dtype=object)
        for rx in range(num_rx):
            acc = send(links[rx, 0], signal[0])
            for tx in range(1, num_tx):
                acc = self.engine.awgn_add(acc, send(links[rx, tx], signal[tx]), 1.0, dtype=self.dtype)   # acc + link output
            outputs[rx] = acc
        return outputs

    def corrupt_data(self, signal):
        """:292-330: signal[tx] = what transmitter tx sends; -> object array, one received stream per receiver."""
        return self._through_every_link(signal, lambda link, x: link.corrupt_data(x))

    def corrupt_data_in_freq_domain(self, signal, fft_size, carrier_indexes=None):
        """:332-394: the block-static per-subcarrier form of every link."""
        return self._through_every_link(signal, lambda link, x: link.corrupt_data_in_freq_domain(x, fft_size,
                                                                                                 carrier_indexes))

    def get_last_impulse_response(self, rx_idx, tx_idx):
        """:396-420."""
        return self._su_siso_channels[rx_idx, tx_idx].get_last_impulse_response()


class MuMimoChannel(MuChannel):
    """reference channels/multiuser.py:521-583: every link carries num_rx_antennas x num_tx_antennas antennas."""

    def __init__(self, N, num_rx_antennas, num_tx_antennas, fading_generator=None, channel_profile=None, tap_powers_dB=None,
                 tap_delays=None, Ts=None, engine=None, dtype=None):
        super().__init__(N, fading_generator, channel_profile, tap_powers_dB, tap_delays, Ts, engine=engine, dtype=dtype)
        for link in self._su_siso_channels.flat:
            link.set_num_antennas(num_rx_antennas, num_tx_antennas)


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
        self._pathloss_matrix = None
        self._pathloss_big_matrix = None

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
    last_noise = property(lambda self: self._last_noise)
    pathloss = property(lambda self: self._pathloss_matrix)

    @property
    def big_H(self):
        """multiuser.py:780-805: the block matrix, times sqrt(path loss) entry by entry when one is set."""
        if self._pathloss_big_matrix is None:
            return self._big_H
        return self._big_H * np.sqrt(self._pathloss_big_matrix)

    @staticmethod
    def _from_small_matrix_to_big_matrix(small_matrix, Nr, Nt, Kr, Kt=None):
        """multiuser.py:860-933: entry (k, l) repeated over the Nr[k] x Nt[l] block."""
        Kt = Kr if Kt is None else Kt
        small_matrix = np.asarray(small_matrix)
        rows = [np.hstack([np.ones((int(Nr[k]), int(Nt[l]))) * small_matrix[k, l] for l in range(Kt)]) for k in range(Kr)]
        return np.vstack(rows)

    def set_pathloss(self, pathloss_matrix=None):
        """multiuser.py:1264-1312: K x K LINEAR power ratios, transmitter l (column) to receiver k (row)."""
        self._pathloss_matrix = None if pathloss_matrix is None else np.array(pathloss_matrix, dtype=float)
        if pathloss_matrix is None:
            self._pathloss_big_matrix = None
        else:
            self._pathloss_big_matrix = self._from_small_matrix_to_big_matrix(self._pathloss_matrix, self._Nr, self._Nt,
                                                                              self._K)

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
        return self.big_H[r0:r0 + self._Nr[k], t0:t0 + self._Nt[l]]

    def get_Hk(self, k):
        r0 = int(np.sum(self._Nr[:k]))
        return self.big_H[r0:r0 + self._Nr[k], :]

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
        out = self.engine.mimo_channel(np.asarray(self.big_H)[np.newaxis], data[np.newaxis],
                                       None if noise is None else noise[np.newaxis], nv, dtype=self.dtype)[0]
        if self.big_W is not None:
            out = np.dot(self.big_W.conjugate().T, out)
        return out

    def corrupt_data(self, data):
        """multiuser.py:1223-1262: list of per-transmitter arrays in, array of per-receiver arrays out."""
        out = self.corrupt_concatenated_data(np.vstack(list(data)))
        n_users = int(self.K)               # not self._K: the ExtInt subclass counts its external sources there
        res = np.zeros(n_users, dtype=np.ndarray)
        cum = np.hstack([0, np.cumsum(self._Nr)])
        for k in range(n_users):
            res[k] = out[cum[k]:cum[k + 1], :]
        return res

    # ---- covariance matrices and SINRs (multiuser.py:1314-2008), evaluated by k_mu_link_stats ------------------------
    def _stats(self, F=None, U=None, joint=False, pe=0.0, want=("Q",)):
        n_users = int(self.K)
        n_ext = int(np.sum(self._Nt)) - int(np.sum(self.Nt))
        return self.engine.mu_link_stats(self._big_H, self.Nr, self.Nt, F=F, U=U, noise_var=self._noise_var or 0.0, pe=pe,
                                         n_ext=n_ext, joint=joint, pathloss_big=self._pathloss_big_matrix, want=want), n_users

    def calc_Q(self, k, F_all_users):
        """Interference-plus-noise covariance at receiver k (:1345-1383)."""
        return self._stats(F_all_users, want=("Q",))[0]["Q"][k][0]

    def calc_JP_Q(self, k, F_all_users):
        """Joint-processing form: every precoder spans all transmit antennas (:1416-1450)."""
        return self._stats(F_all_users, joint=True, want=("Q",))[0]["Q"][k][0]

    def _calc_Bkl_cov_matrix_all_l(self, F_all_users, k, N0_or_Rek=0.0):
        """:1552-1621: a noise power, or (as the ExtInt class passes it, :2670) receiver k's interference-plus-noise
        covariance matrix Rek, which enters every B^{[kl]} as an additive term (:1494-1510)."""
        Rek = None
        if N0_or_Rek is not None and not np.isscalar(N0_or_Rek):
            Rek = np.asarray(N0_or_Rek)
            if Rek.shape != (int(self.Nr[k]), int(self.Nr[k])):
                raise ValueError("Rek must be a [Nr[k], Nr[k]] matrix")
        keep, self._noise_var = self._noise_var, (0.0 if Rek is not None else float(N0_or_Rek or 0.0))
        try:
            B = self._stats(F_all_users, want=("B",))[0]["B"][k][0]
        finally:
            self._noise_var = keep
        out = np.empty(B.shape[0], dtype=np.ndarray)
        for l in range(B.shape[0]):
            out[l] = B[l] if Rek is None else B[l] + Rek
        return out

    def _sinr(self, F, U, joint, pe=0.0):
        res, n_users = self._stats(F, U, joint=joint, pe=pe, want=("sinr",))
        out = np.empty(n_users, dtype=np.ndarray)
        for k in range(n_users):
            out[k] = res["sinr"][k][0]
        return out

    def calc_SINR(self, F, U):
        """Per-stream SINRs of every user for precoders F and receive filters U (:1869-1899)."""
        return self._sinr(F, U, False)

    def calc_JP_SINR(self, F, U):
        """:1978-2008."""
        return self._sinr(F, U, True)


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

    def get_Hk_without_ext_int(self, k):
        """multiuser.py:2191-2239."""
        return self.get_Hk(k)[:, :int(np.sum(self.Nt))]

    def get_Hk_with_ext_int(self, k):
        """multiuser.py:2241-2287."""
        return self.get_Hk(k)

    H_no_ext_int = property(lambda self: self.H[:self.K, :self.K])

    def set_pathloss(self, pathloss_matrix=None, ext_int_pathloss=None):
        """multiuser.py:2415-2467: K x K between the users plus K x extIntK from the external sources."""
        if pathloss_matrix is None:
            self._pathloss_matrix = None
            self._pathloss_big_matrix = None
            return
        full = np.hstack([np.asarray(pathloss_matrix, dtype=float), np.asarray(ext_int_pathloss, dtype=float)])
        self._pathloss_matrix = full
        self._pathloss_big_matrix = self._from_small_matrix_to_big_matrix(full, self._Nr, self._Nt, self.K, self._K)

    def corrupt_data(self, data, ext_int_data):
        """multiuser.py:2130-2160: the users' data followed by the external sources' data."""
        return MultiUserChannelMatrix.corrupt_data(self, list(data) + list(ext_int_data))

    def calc_cov_matrix_extint_without_noise(self, pe=1.0):
        """multiuser.py:2469-2494."""
        keep, self._noise_var = self._noise_var, None
        try:
            Re = self._stats(pe=pe, want=("Re",))[0]["Re"]
        finally:
            self._noise_var = keep
        out = np.empty(self.K, dtype=np.ndarray)
        for k in range(self.K):
            out[k] = Re[k][0]
        return out

    def calc_cov_matrix_extint_plus_noise(self, pe=1.0):
        """multiuser.py:2496-2520."""
        Re = self._stats(pe=pe, want=("Re",))[0]["Re"]
        out = np.empty(self.K, dtype=np.ndarray)
        for k in range(self.K):
            out[k] = Re[k][0]
        return out

    def calc_Q(self, k, F_all_users, pe=1.0):
        """multiuser.py:2530-2567."""
        return self._stats(F_all_users, pe=pe, want=("Q",))[0]["Q"][k][0]

    def calc_JP_Q(self, k, F_all_users, pe=1.0):
        """multiuser.py:2598-2634."""
        return self._stats(F_all_users, joint=True, pe=pe, want=("Q",))[0]["Q"][k][0]

    def calc_SINR(self, F, U, pe=1.0):
        """multiuser.py:2636-2674."""
        return self._sinr(F, U, False, pe)

    def calc_JP_SINR(self, F, U, pe=1.0):
        """multiuser.py:2771-2807."""
        return self._sinr(F, U, True, pe)
