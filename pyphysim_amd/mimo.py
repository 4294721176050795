"""Host mirror of pyphysim.mimo.Blast / MRC (reference mimo/mimo.py:30-660, 789-827): same
methods, argument meaning and errors; encode / filter / decode run in libmcle's HIP kernels."""
import math

import numpy as np

from .engine import DeviceArray, get_engine


class MimoBase:
    """reference mimo.py:30-459 (the pieces Blast needs)."""

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
            if Nt > Nr:
                raise ValueError("Blast scheme requires at least as many receive antennas as transmit antennas")
        super().set_channel_matrix(channel)

    def getNumberOfLayers(self):
        return self.Nt

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
