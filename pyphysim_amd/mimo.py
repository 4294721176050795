"""Host mirror of pyphysim.mimo.Blast / MRC (reference mimo/mimo.py:30-660, 789-827): same
methods, argument meaning and errors; encode / filter / decode run in libmcle's HIP kernels."""
import math
import warnings

import numpy as np

from .engine import DeviceArray, get_engine


def calc_post_processing_linear_SINRs(channel, W, G_H, noise_var=None, engine=None):
    """reference mimo.py:62-118: linear SINR of every stream of precoder W / receive filter G_H on `channel`."""
    eng = engine if engine is not None else get_engine()
    G = np.atleast_2d(np.asarray(G_H, dtype=complex))
    Wm = np.asarray(W, dtype=complex)
    if Wm.ndim == 1:
        Wm = Wm[:, np.newaxis]
    H = np.atleast_2d(np.asarray(channel, dtype=complex))
    return eng.post_processing_sinrs(H[np.newaxis], Wm[np.newaxis], G[np.newaxis], noise_var or 0.0)[0]


def calc_post_processing_SINRs(channel, W, G_H, noise_var=None, engine=None):
    """reference mimo.py:33-59: the same in dB."""
    return 10.0 * np.log10(calc_post_processing_linear_SINRs(channel, W, G_H, noise_var, engine))


class MimoBase:
    """reference mimo.py:30-459 (the pieces Blast needs)."""

    def _precoder_and_filter(self, noise_var):
        raise NotImplementedError("this MIMO scheme has no linear precoder / receive filter pair")

    def calc_linear_SINRs(self, noise_var):
        """reference mimo.py:311-326: post-processing SINRs of the scheme's own precoder and receive filter.
        Drop-in includes the reference's quirk: despite its name the method returns calc_post_processing_SINRs,
        i.e. values in dB (mimo.py:325), and calc_SINRs applies linear2dB once more (mimo.py:345)."""
        W, G_H = self._precoder_and_filter(noise_var)
        return calc_post_processing_SINRs(self._channel, W, G_H, noise_var, self.engine)

    def calc_SINRs(self, noise_var):
        """reference mimo.py:328-345."""
        return 10.0 * np.log10(self.calc_linear_SINRs(noise_var))

    def __init__(self, channel=None, engine=None, dtype=None):
        self._channel = None
        self._engine, self.dtype = engine, dtype
        if channel is not None:
            self.set_channel_matrix(channel)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def set_channel_matrix(self, channel):
        channel = np.asarray(channel)
        if channel.ndim == 1:
            channel = channel[:, np.newaxis]      # MRC-style single column (mimo.py:816-820)
        self._channel = channel

    def getNumberOfLayers(self):          # pragma: no cover
        """mimo.py:347-355."""
        raise NotImplementedError("getNumberOfLayers still needs to be implemented in the {0} class".format(
            self.__class__.__name__))

    def encode(self, transmit_data):      # pragma: no cover
        """mimo.py:357-370."""
        raise NotImplementedError("encode still needs to be implemented in the {0} class".format(self.__class__.__name__))

    def decode(self, received_data):      # pragma: no cover
        """mimo.py:372-385."""
        raise NotImplementedError("decode still needs to be implemented in the {0} class".format(self.__class__.__name__))

    @property
    def Nt(self):
        return int(self._channel.shape[1])

    @property
    def Nr(self):
        return int(self._channel.shape[0])


class Blast(MimoBase):
    """reference mimo.py:465-660."""

    def __init__(self, channel=None, engine=None, dtype=None):
        super().__init__(None, engine, dtype)
        self._noise_var = 0.0
        if channel is not None:
            self.set_channel_matrix(channel)

    def set_channel_matrix(self, channel):
        channel = np.asarray(channel)
        if channel.ndim == 2:
            Nr, Nt = channel.shape
            if Nt > Nr:       # the reference only warns (mimo.py:505-515)
                warnings.warn("The number of transmit antennas for {0} should not be greater than the number "
                              "of receive antennas.".format(self.__class__.__name__))
        super().set_channel_matrix(channel)

    def getNumberOfLayers(self):
        return self.Nt

    def _precoder_and_filter(self, noise_var):
        """mimo.py:556-607: W = I / sqrt(Nt); G_H = sqrt(Nt) x (MMSE if noise_var > 0 else zero forcing)."""
        G, _ = self.engine.blast_filter(self._channel[np.newaxis], noise_var or 0.0, dtype="f64")
        return np.eye(self.Nt) / math.sqrt(self.Nt), G[0]

    def set_noise_var(self, noise_var):
        """None / 0 -> zero forcing; > 0 -> MMSE (mimo.py:529-553)."""
        if noise_var is None:
            self._noise_var = 0.0
        elif noise_var >= 0.0:
            self._noise_var = noise_var
        else:
            raise ValueError("Noise variance must be a non-negative value.")

    def encode(self, transmit_data):
        """reshape((Nt, -1), order='F') / sqrt(Nt) (mimo.py:609-641)."""
        x = np.asarray(transmit_data).reshape(-1)
        if x.size % self.Nt != 0:
            raise ValueError("Input array number of elements must be a multiple of the number of transmit "
                             "antennas.")
        return self.engine.blast_encode(x, self.Nt, dtype=self.dtype)[0]

    def receive_filter(self):
        """sqrt(Nt) * (MMSE if noise_var > 0 else zero forcing) (mimo.py:577-607)."""
        G, skipped = self.engine.blast_filter(self._channel[np.newaxis], self._noise_var, dtype=self.dtype)
        return G[0]

    def decode(self, received_data):
        """(G @ received).reshape(-1, order='F') (mimo.py:643-660)."""
        Y = np.asarray(received_data)
        if Y.ndim == 1:
            Y = Y[:, np.newaxis]
        G, _ = self.engine.blast_filter(self._channel[np.newaxis], self._noise_var, dtype=self.dtype)
        return self.engine.blast_decode(G, Y[np.newaxis], dtype=self.dtype)[0]


class MRC(Blast):
    """reference mimo.py:789-827: Blast with a (possibly 1-D) channel."""

    def set_channel_matrix(self, channel):
        """mimo.py:810-827: a 1-D channel is one transmit antenna seen by len(channel) receive antennas."""
        channel = np.asarray(channel)
        super().set_channel_matrix(channel[:, np.newaxis] if channel.ndim == 1 else channel)


class MisoBase(MimoBase):
    """reference mimo.py:388-441: schemes for a single receive antenna (MRT, Alamouti's MISO form)."""

    def set_channel_matrix(self, channel):
        channel = np.asarray(channel)
        if channel.ndim == 1:
            channel = channel[np.newaxis, :]
        elif channel.shape[0] != 1:
            raise ValueError("The MRT scheme is only defined for the scenario with a single receive antenna")
        self._channel = channel


class MRT(MimoBase):
    """reference mimo.py:666-783 (MISO: one receive antenna).  Phase-only transmit precoder."""

    def set_channel_matrix(self, channel):
        channel = np.asarray(channel)
        if channel.ndim == 1:
            channel = channel[np.newaxis, :]
        elif channel.shape[0] != 1:
            raise ValueError("The MRT scheme is only defined for the scenario with a single receive antenna")
        self._channel = channel

    def getNumberOfLayers(self):
        return 1

    def _precoder_and_filter(self, noise_var):
        """mimo.py:687-731: phase-only precoder, scalar receive gain."""
        W = np.exp(-1j * np.angle(self._channel)).T / math.sqrt(self.Nt)
        return W, np.array([[math.sqrt(self.Nt) / np.sum(np.abs(self._channel))]])

    def encode(self, transmit_data):
        x = np.asarray(transmit_data).reshape(1, -1)
        return self.engine.mrt_encode(self._channel.reshape(1, -1), x, dtype=self.dtype)[0]

    def decode(self, received_data):
        y = np.asarray(received_data).reshape(1, -1)
        return self.engine.mrt_decode(self._channel.reshape(1, -1), y, dtype=self.dtype)[0]


class SVDMimo(Blast):
    """reference mimo.py:833-946: precoder V / sqrt(Nt), receive filter diag(1/S) U^H sqrt(Nt); square
    channels up to 4x4 (Jacobi SVD in f64 on the GPU).  W and G are a consistent singular-vector
    pair; LAPACK's particular phase choice is not reproduced (G H W = I either way)."""

    _injected = None

    def set_filters(self, W, G_H):
        """Injected basis: use this precoder / receive filter pair instead of the device decomposition's.  An SVD fixes
        every singular-vector pair only up to a common phase, and the decisions depend on it (the filtered noise is rotated
        by it); with the reference's own pair (LAPACK's phases) encode / decode reproduce the reference's estimates and
        decisions exactly (tests/test_gpu_pipelines.py::test_svd_gmd_with_the_references_basis).  None restores the
        device decomposition."""
        self._injected = None if W is None else (np.asarray(W, dtype=complex), np.asarray(G_H, dtype=complex))

    def _filters(self):
        if self._injected is not None:
            return self._injected[0], self._injected[1], None
        W, G, S = self.engine.svd_filters(self._channel[np.newaxis], dtype=self.dtype)
        return W[0], G[0], S[0]

    def _precoder_and_filter(self, noise_var):
        W, G, _ = self.engine.svd_filters(self._channel[np.newaxis], dtype="f64")
        return W[0], G[0]

    def encode(self, transmit_data):
        x = np.asarray(transmit_data).reshape(-1)
        if x.size % self.Nt != 0:
            raise ValueError("Input array number of elements must be a multiple of the number of transmit antennas")
        W, _, _ = self._filters()
        return self.engine.mimo_channel(W[np.newaxis], x.reshape(1, self.Nt, -1), dtype=self.dtype)[0]

    def decode(self, received_data):
        _, G, _ = self._filters()
        Y = np.asarray(received_data)
        return self.engine.mimo_channel(G[np.newaxis], Y[np.newaxis], dtype=self.dtype)[0].reshape(-1)


class GMDMimo(Blast):
    """reference mimo.py:952-1067: geometric mean decomposition precoder (util.misc.gmd) and Blast's
    ZF / MMSE filter on the equivalent channel Q R; square channels up to 4x4."""

    _injected = None
    set_filters = SVDMimo.set_filters

    def _filters(self):
        if self._injected is not None:
            return self._injected[0], self._injected[1], None
        W, G, R = self.engine.gmd_filters(self._channel[np.newaxis], self._noise_var, dtype=self.dtype)
        return W[0], G[0], R[0]

    def _precoder_and_filter(self, noise_var):
        W, G, _ = self.engine.gmd_filters(self._channel[np.newaxis], noise_var or 0.0, dtype="f64")
        return W[0], G[0]

    def encode(self, transmit_data):
        x = np.asarray(transmit_data).reshape(-1)
        if x.size % self.Nt != 0:
            raise ValueError("Input array number of elements must be a multiple of the number of transmit antennas")
        W, _, _ = self._filters()
        return self.engine.mimo_channel(W[np.newaxis], x.reshape(1, self.Nt, -1), dtype=self.dtype)[0]

    def decode(self, received_data):
        _, G, _ = self._filters()
        Y = np.asarray(received_data)
        return self.engine.mimo_channel(G[np.newaxis], Y[np.newaxis], dtype=self.dtype)[0].reshape(-1)


class Alamouti(MimoBase):
    """reference mimo.py:1073-1287: 2 transmit antennas, any number of receive antennas."""

    def set_channel_matrix(self, channel):
        channel = np.asarray(channel)
        if channel.ndim == 1:
            channel = channel[np.newaxis, :]
        elif channel.shape[1] != 2:
            raise ValueError("The number of transmit antennas must be equal to 2 for the {0} scheme".format(
                self.__class__.__name__))
        self._channel = channel

    def getNumberOfLayers(self):
        return 1

    def calc_linear_SINRs(self, noise_var):
        """mimo.py:1147-1166."""
        return np.linalg.norm(self._channel, "fro") ** 2 / noise_var

    def encode(self, transmit_data):
        x = np.asarray(transmit_data).reshape(-1)
        return self.engine.alamouti_encode(x, dtype=self.dtype)[0]

    def decode(self, received_data):
        Y = np.asarray(received_data)
        if Y.ndim == 1:
            Y = Y[np.newaxis, :]
        return self.engine.alamouti_decode(self._channel[np.newaxis], Y[np.newaxis], dtype=self.dtype)[0]
