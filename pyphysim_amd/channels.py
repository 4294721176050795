"""Host mirror of pyphysim.channels for the hot path: TdlChannelProfile (+ COST259 profiles),
JakesSampleGenerator, RayleighSampleGenerator, TdlImpulseResponse, TdlChannel (SISO) and
SuChannel, with the reference's names, arguments, state evolution and errors
(reference channels/fading_generators.py, channels/fading.py, channels/singleuser.py).

Host side = object state (phases, time bookkeeping, profile discretisation); the sample
generation (sum of sinusoids) and the time-varying convolution run in libmcle's HIP kernels.
"""
import math

import numpy as np

from . import util
from .engine import DeviceArray, get_engine


def discretize_profile(tap_powers_dB, tap_delays, Ts):
    """reference fading.py:272-304 plus the dB round trip of the discretised profile's ctor
    (:77-78): delays -> rounded sample indexes (merged), powers summed per index and normalised
    to sum 1.  Returns (tap_powers_linear, delay_indexes)."""
    tap_powers_dB = np.asarray(tap_powers_dB, dtype=float)
    tap_delays = np.asarray(tap_delays, dtype=float)
    idx, inv = np.unique(np.round(tap_delays / Ts).astype(int).flatten(), return_inverse=True)
    acc = np.zeros(idx.size)
    np.add.at(acc, inv, util.dB2Linear(tap_powers_dB))
    acc /= np.sum(acc)
    return util.dB2Linear(util.linear2dB(acc)), idx


class TdlChannelProfile:
    """reference fading.py:28-315."""

    def __init__(self, tap_powers_dB=None, tap_delays=None, name="custom"):
        self._name = name
        if tap_powers_dB is None and tap_delays is None:
            tap_powers_dB = np.zeros(1)
            tap_delays = np.zeros(1)
        tap_powers_dB = np.asarray(tap_powers_dB, dtype=float)
        self._tap_powers_dB = tap_powers_dB.copy()
        self._tap_powers_linear = util.dB2Linear(tap_powers_dB)
        self._tap_delays = np.asarray(tap_delays).copy()
        self._num_taps = self._tap_delays.size
        lin = self._tap_powers_linear
        self._mean_excess_delay = np.sum(lin * self._tap_delays) / np.sum(lin)
        second = np.sum(lin * self._tap_delays ** 2) / np.sum(lin)
        self._rms_delay_spread = math.sqrt(max(second - self._mean_excess_delay ** 2, 0.0))
        self._Ts = None
        for a in (self._tap_powers_dB, self._tap_powers_linear, self._tap_delays):
            a.flags["WRITEABLE"] = False

    name = property(lambda self: self._name)
    tap_powers_dB = property(lambda self: self._tap_powers_dB)
    tap_powers_linear = property(lambda self: self._tap_powers_linear)
    tap_delays = property(lambda self: self._tap_delays)
    num_taps = property(lambda self: self._num_taps)
    mean_excess_delay = property(lambda self: self._mean_excess_delay)
    rms_delay_spread = property(lambda self: self._rms_delay_spread)
    Ts = property(lambda self: self._Ts)
    is_discretized = property(lambda self: self._Ts is not None)

    @property
    def num_taps_with_padding(self):
        if not self.is_discretized:
            raise RuntimeError("TdlChannelProfile is not discretized")
        return int(self._tap_delays[-1]) + 1

    def get_discretize_profile(self, Ts):
        if self.is_discretized:
            raise RuntimeError("Trying to discretize a TdlChannelProfile that is already discretized.")
        lin, idx = discretize_profile(self._tap_powers_dB, self._tap_delays, Ts)
        prof = TdlChannelProfile(util.linear2dB(lin), idx, self._name + " (discretized)")
        prof._Ts = Ts
        return prof

    def __repr__(self):
        return "<TdlChannelProfile: {0} taps>".format(self._num_taps)


COST259_TUx = TdlChannelProfile(
    np.array([-5.7, -7.6, -10.1, -10.2, -10.2, -11.5, -13.4, -16.3, -16.9, -17.1, -17.4, -19, -19, -19.8, -21.5,
              -21.6, -22.1, -22.6, -23.5, -24.3]),
    np.array([0, 217, 512, 514, 517, 674, 882, 1230, 1287, 1311, 1349, 1533, 1535, 1622, 1818, 1836, 1884, 1943,
              2048, 2140]) * 1e-9, "COST259_TU")
COST259_RAx = TdlChannelProfile(
    np.array([-5.2, -6.4, -8.4, -9.3, -10.0, -13.1, -15.3, -18.5, -20.4, -22.4]),
    np.array([0., 42., 101., 129., 149., 245., 312., 410., 469., 528]) * 1e-9, "COST259_RA")
COST259_HTx = TdlChannelProfile(
    np.array([-3.6, -8.9, -10.2, -11.5, -11.8, -12.7, -13.0, -16.2, -17.3, -17.7, -17.6, -22.7, -24.1, -25.8,
              -25.8, -26.2, -29.0, -29.9, -30.0, -30.7]),
    np.array([0., 356., 441., 528., 546., 609., 625., 842., 916., 941., 15000., 16172., 16492., 16876., 16882.,
              16978., 17615., 17827., 17849., 18016.]) * 1e-9, "COST259_HT")


def generate_jakes_samples(Fd, Ts=1e-3, NSamples=100, L=8, shape=None, current_time=0, phi_l=None, psi_l=None,
                           engine=None, dtype=None):
    """fading_generators.py:15-98, the function form of the Jakes generator -> (new_current_time, h [shape..., NSamples]).
    phi_l / psi_l default to np.random.rand(L, *shape, 1) as there (NOT times 2 pi: this legacy entry point feeds the
    unit-interval draws to cos() as they are; JakesSampleGenerator scales them).  Sum of sinusoids on the device."""
    t = np.arange(current_time, NSamples * Ts + current_time, Ts * 1.0000000001)
    dims = [L] + ([] if shape is None else list(shape)) + [1]
    if phi_l is None:
        phi_l = np.random.rand(*dims)
    if psi_l is None:
        psi_l = np.random.rand(*dims)
    streams = int(np.prod(dims[1:-1])) if len(dims) > 2 else 1
    eng = engine if engine is not None else get_engine()
    h = eng.jakes_generate(np.asarray(phi_l).reshape(L, streams), np.asarray(psi_l).reshape(L, streams), Fd, 0.0, 0.0,
                           t.size, dtype=dtype, times=t)
    h = np.asarray(h).reshape((tuple(shape) if shape is not None else ()) + (t.size,))
    return t[-1] + Ts, h


class FadingSampleGenerator:
    """reference fading_generators.py:101-205."""

    def __init__(self, shape=None):
        self._shape = None
        self._samples = None
        if shape is not None:
            self._shape = (shape,) if isinstance(shape, int) else tuple(shape)

    @property
    def shape(self):
        return self._shape

    @shape.setter
    def shape(self, new_shape):
        self._shape = None if new_shape is None else ((new_shape,) if isinstance(new_shape, int)
                                                      else tuple(new_shape))

    def get_samples(self):
        return self._samples

    def skip_samples_for_next_generation(self, num_samples):
        raise NotImplementedError

    def generate_more_samples(self, num_samples=None):
        raise NotImplementedError

    def get_similar_fading_generator(self):
        """fading_generators.py:198-205."""
        raise NotImplementedError("Implement in a subclass")


class RayleighSampleGenerator(FadingSampleGenerator):
    """reference fading_generators.py:208-282: i.i.d. CN(0,1) samples (drawn on the GPU)."""

    def __init__(self, shape=None):
        super().__init__(shape)
        self.generate_more_samples()

    def generate_more_samples(self, num_samples=None):
        if self._shape is None:
            shape = () if num_samples is None else (num_samples,)
        else:
            shape = tuple(self._shape) + (() if num_samples is None else (num_samples,))
        self._samples = util.randn_c(*shape) if shape else util.randn_c(1)[0]

    def skip_samples_for_next_generation(self, num_samples):
        pass

    def generate_block_samples(self, num_blocks, skip):
        self.generate_more_samples(num_blocks)
        self._samples = np.asarray(self._samples).reshape((tuple(self._shape) if self._shape else (1,))
                                                          + (num_blocks,))
        return self._samples

    def get_similar_fading_generator(self):
        return RayleighSampleGenerator(self._shape)


class JakesSampleGenerator(FadingSampleGenerator):
    """reference fading_generators.py:289-553.  phi/psi are drawn on the host from ``RS`` exactly
    as the reference does (2*pi*RS.rand(L, *shape, 1), phi first; re-drawn whenever ``shape`` is
    set), time advances the same way, and the L x streams x samples sum of sinusoids runs on the
    GPU with the phase accumulated in f64."""

    def __init__(self, Fd=100, Ts=1e-3, L=8, shape=None, RS=None, engine=None, dtype=None):
        super().__init__(shape)
        self._Fd, self._Ts, self._L = Fd, Ts, L
        self._phi_l = self._psi_l = None
        self.RS = np.random if RS is None else RS
        self._engine, self.dtype = engine, dtype
        self._current_time = 0.0
        self._set_phi_and_psi_according_to_shape()
        self.generate_more_samples()

    L = property(lambda self: self._L)
    Ts = property(lambda self: self._Ts)
    Fd = property(lambda self: self._Fd)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    @property
    def shape(self):
        return self._shape

    @shape.setter
    def shape(self, new_shape):
        FadingSampleGenerator.shape.fset(self, new_shape)
        self._set_phi_and_psi_according_to_shape()

    def _set_phi_and_psi_according_to_shape(self):
        dims = [self._L] + ([] if self._shape is None else list(self._shape)) + [1]
        self._phi_l = 2 * np.pi * self.RS.rand(*dims)
        self._psi_l = 2 * np.pi * self.RS.rand(*dims)

    def _next_time_axis(self, num_samples):
        """(t0, delta) of np.arange(t0, n*Ts + t0, Ts*1.0000000001) and the state update of
        fading_generators.py:459-467 (t_k = t0 + k*delta, delta = fl(fl(t0 + step) - t0))."""
        n = 1 if num_samples is None else int(num_samples)
        t = np.arange(self._current_time, n * self._Ts + self._current_time, self._Ts * 1.0000000001)
        t0 = self._current_time
        delta = float(t[1] - t[0]) if t.size > 1 else self._Ts * 1.0000000001
        self._current_time = t[-1] + self._Ts
        return t0, delta, t.size

    def generate_more_samples(self, num_samples=None):
        t0, delta, n = self._next_time_axis(num_samples)
        streams = int(np.prod(self._shape)) if self._shape else 1
        h = self.engine.jakes_generate(self._phi_l.reshape(self._L, streams), self._psi_l.reshape(self._L, streams),
                                       self._Fd, t0, delta, n, dtype=self.dtype)
        self._samples = h.reshape((tuple(self._shape) if self._shape else (1,)) + (n,))
        if self._shape is None:
            self._samples = self._samples.reshape(1, n)[0] if num_samples is None else self._samples.reshape(n)

    def skip_samples_for_next_generation(self, num_samples):
        self._current_time += num_samples * self._Ts

    def generate_block_samples(self, num_blocks, skip):
        """`num_blocks` x [generate_more_samples(1); skip_samples_for_next_generation(skip)] in one launch:
        the sample times follow the same floating-point recurrence as the two reference calls."""
        times = np.empty(num_blocks)
        for i in range(num_blocks):
            times[i] = self._current_time
            self._current_time = self._current_time + self._Ts
            self._current_time += skip * self._Ts
        streams = int(np.prod(self._shape)) if self._shape else 1
        h = self.engine.jakes_generate(self._phi_l.reshape(self._L, streams), self._psi_l.reshape(self._L, streams),
                                       self._Fd, 0.0, 0.0, num_blocks, dtype=self.dtype, times=times)
        self._samples = h.reshape((tuple(self._shape) if self._shape else (1,)) + (num_blocks,))
        return self._samples

    def get_similar_fading_generator(self):
        return JakesSampleGenerator(self._Fd, self._Ts, self._L, self._shape, engine=self._engine, dtype=self.dtype)


class TdlImpulseResponse:
    """reference fading.py:356-698 (the parts the hot path touches)."""

    def __init__(self, tap_values, channel_profile):
        self._channel_profile = channel_profile
        self._tap_values_sparse = tap_values
        self._tap_values_dense = None

    tap_values_sparse = property(lambda self: self._tap_values_sparse)
    channel_profile = property(lambda self: self._channel_profile)

    @property
    def tap_indexes_sparse(self):
        return self._channel_profile.tap_delays

    @property
    def tap_delays_sparse(self):
        return self._channel_profile.tap_delays

    @property
    def Ts(self):
        return self._channel_profile.Ts

    @property
    def num_samples(self):
        return self._tap_values_sparse.shape[-1]

    @property
    def tap_values(self):
        """dense [max_delay + 1, ..., n] copy with zeros at the unused delays (fading.py:482-511)."""
        if self._tap_values_dense is None:
            sp = np.asarray(self._tap_values_sparse)
            dense = np.zeros((int(self.tap_indexes_sparse[-1]) + 1,) + sp.shape[1:], dtype=complex)
            dense[np.asarray(self.tap_indexes_sparse)] = sp
            dense.flags["WRITEABLE"] = False
            self._tap_values_dense = dense
        return self._tap_values_dense

    @staticmethod
    def concatenate_samples(impulse_responses):
        """fading.py:655-698: one impulse response holding the samples of several, in order (same profile object)."""
        if len(impulse_responses) < 2:
            if len(impulse_responses) == 1:
                return impulse_responses[0]
            raise ValueError("impulse_responses must contain at least two TdlImpulseResponse objects.")
        if impulse_responses[0].channel_profile is not impulse_responses[1].channel_profile:
            raise ValueError("TdlImpulseResponse objects must have the same channel profile object")
        taps = np.concatenate([np.asarray(a.tap_values_sparse) for a in impulse_responses], axis=-1)
        return TdlImpulseResponse(taps, impulse_responses[0].channel_profile)

    def get_freq_response(self, fft_size):
        """fading.py:513-536.  Host FFT of the dense taps: an analysis helper, off the hot path
        (the equaliser kernel never materialises it)."""
        return np.fft.fft(self.tap_values, fft_size, axis=0)


class TdlChannel:
    """reference fading.py:701-1287, SISO branch.  corrupt_data = time-varying sparse convolution
    with one impulse response per sample, executed on the GPU."""

    def __init__(self, fading_generator, channel_profile=None, tap_powers_dB=None, tap_delays=None, Ts=None,
                 engine=None, dtype=None):
        if isinstance(fading_generator, JakesSampleGenerator):
            if Ts is None:
                Ts = fading_generator.Ts
            elif Ts != fading_generator.Ts:
                raise RuntimeError("The provided sampling interval Ts is different from the one in the Jakes "
                                   "sample generator.")
        if channel_profile is None:
            channel_profile = TdlChannelProfile(tap_powers_dB, tap_delays)
        elif not isinstance(channel_profile, TdlChannelProfile):
            raise AssertionError("channel_profile must be an obj of the TdlChannelProfile class")
        if not channel_profile.is_discretized:
            if isinstance(fading_generator, RayleighSampleGenerator) and Ts is None:
                Ts = 1.0
            assert Ts is not None
            channel_profile = channel_profile.get_discretize_profile(Ts)
        elif channel_profile.Ts != Ts and Ts is not None:
            raise RuntimeError("Channel profile is already discretized, but it does not agree with the "
                               "discretized parameter Ts")
        self._channel_profile = channel_profile
        self._fading_generator = fading_generator
        self._engine = engine if engine is not None else getattr(fading_generator, "_engine", None)
        self.dtype = dtype if dtype is not None else getattr(fading_generator, "dtype", None)
        # the generator produces one stream per tap: shape (num_taps,) [+ (Nr, Nt)]; assigning the
        # shape re-draws the Jakes phases (fading.py:796-798, fading_generators.py:381-386)
        base = fading_generator.shape
        self._set_fading_generator_shape(tuple(base) if base else None)
        self._last_impulse_response = None
        self._switched_direction = False

    channel_profile = property(lambda self: self._channel_profile)
    num_taps = property(lambda self: self._channel_profile.num_taps)
    num_taps_with_padding = property(lambda self: self._channel_profile.num_taps_with_padding)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def _set_fading_generator_shape(self, new_shape):
        """(num_taps,) + new_shape (fading.py:854-868)."""
        self._fading_generator.shape = (self.num_taps,) + (tuple(new_shape) if new_shape else ())

    def set_num_antennas(self, num_rx_antennas, num_tx_antennas):
        self._set_fading_generator_shape((num_rx_antennas, num_tx_antennas))

    @property
    def num_tx_antennas(self):
        shp = self._fading_generator.shape
        return -1 if shp is None or len(shp) == 1 else shp[2]

    @property
    def num_rx_antennas(self):
        shp = self._fading_generator.shape
        return -1 if shp is None or len(shp) == 1 else shp[1]

    @property
    def switched_direction(self):
        return self._switched_direction

    @switched_direction.setter
    def switched_direction(self, value):
        if not isinstance(value, bool):
            raise TypeError("switched_direction must be a boolean value")
        self._switched_direction = value

    def generate_impulse_response(self, num_samples=1):
        self._fading_generator.generate_more_samples(num_samples)
        fading = np.asarray(self._fading_generator.get_samples())
        fading = fading.reshape(tuple(self._fading_generator.shape) + (-1,))
        amp = np.sqrt(self._channel_profile.tap_powers_linear).reshape((-1,) + (1,) * (fading.ndim - 1))
        self._last_impulse_response = TdlImpulseResponse(fading * amp, self._channel_profile)

    def get_last_impulse_response(self):
        """fading.py:991-1007: raises until an impulse response has been generated."""
        if self._last_impulse_response is None:
            raise RuntimeError("No impulse response was generated yet")
        return self._last_impulse_response

    def corrupt_data(self, signal):
        """fading.py:1046-1124: SISO (generator shape (taps,)) or MIMO (shape (taps, Nr, Nt))."""
        signal = np.asarray(signal)
        shp = self._fading_generator.shape
        if len(shp) == 1:
            if signal.ndim != 1:
                raise ValueError("a SISO TdlChannel takes a 1-D signal")
            self.generate_impulse_response(signal.shape[-1])
            ir = self._last_impulse_response
            return self.engine.tdl_apply(signal, ir.tap_values_sparse, ir.tap_indexes_sparse, dtype=self.dtype)
        if len(shp) != 3:
            raise RuntimeError("Shape of the fading generator of the TdlChannel class must have either 1 (SISO) "
                               "or 3 (MIMO) dimensions")
        _, nr, nt = shp
        n_in = nr if self._switched_direction else nt
        if n_in == 1 and signal.ndim == 1:
            signal = signal.reshape(1, -1)
        self.generate_impulse_response(signal.shape[-1])
        ir = self._last_impulse_response
        taps = ir.tap_values_sparse
        if self._switched_direction:
            # reverse link (fading.py:1098-1106): the same taps seen from the other side -- antenna axes swapped
            taps = np.ascontiguousarray(np.swapaxes(np.asarray(taps), 1, 2))
        return self.engine.tdl_apply_mimo(signal, taps, ir.tap_indexes_sparse, dtype=self.dtype)


    def corrupt_data_in_freq_domain(self, signal, fft_size, carrier_indexes=None):
        """fading.py:1126-1287: block-static channel applied per subcarrier -- one impulse response per
        block of `fft_size` samples (the generator skips fft_size - 1 samples between blocks)."""
        signal = np.asarray(signal)
        num_symbols = signal.shape[-1]
        if carrier_indexes is None:
            block_size, pick = fft_size, None
        elif isinstance(carrier_indexes, slice):
            lo, hi, st = carrier_indexes.indices(fft_size)
            block_size, pick = (hi - lo) // st, np.arange(fft_size)[carrier_indexes]
        else:
            pick = np.asarray(carrier_indexes)
            block_size = len(pick)
        if num_symbols % block_size != 0:
            raise ValueError("The num of elements in `signal` must be a multiple of number of sent elements per "
                             "`fft_size`.")
        shp = self._fading_generator.shape
        if len(shp) not in (1, 3):
            raise RuntimeError("Shape of the fading generator of the TdlChannel class must have either 1 (SISO) "
                               "or 3 (MIMO) dimensions")
        n_blocks = num_symbols // block_size
        fading = np.asarray(self._fading_generator.generate_block_samples(n_blocks, fft_size - 1))
        fading = fading.reshape(tuple(shp) + (n_blocks,))
        amp = np.sqrt(self._channel_profile.tap_powers_linear).reshape((-1,) + (1,) * (fading.ndim - 1))
        taps = fading * amp
        self._last_impulse_response = TdlImpulseResponse(taps, self._channel_profile)
        H = self.engine.tdl_mean_freq_response(taps, self._channel_profile.tap_delays, n_blocks, fft_size, 0, 0,
                                               dtype=self.dtype, group=1)          # [blocks, fft, (nr, nt)]
        if pick is not None:
            H = H[:, pick]
        if len(shp) == 1:
            return self.engine.cmul(H.reshape(-1), signal.reshape(-1), dtype=self.dtype)
        _, nr, nt = shp
        H = np.asarray(H).reshape(num_symbols, nr, nt)
        if self._switched_direction:          # fading.py:1254-1258: out[:, t] = sum_r H[:, r, t] * signal[r]
            H = np.ascontiguousarray(np.swapaxes(H, 1, 2))
            nr, nt = nt, nr
        if nt == 1 and signal.ndim == 1:
            signal = signal.reshape(1, -1)
        out = self.engine.blast_decode_per_subcarrier(H, signal, dtype=self.dtype)
        return out.reshape(num_symbols, nr).T


class TdlMimoChannel(TdlChannel):
    """reference fading.py:1290-1333: the generator must carry a (Nr, Nt) shape."""

    def __init__(self, fading_generator, channel_profile=None, tap_powers_dB=None, tap_delays=None, Ts=None,
                 engine=None, dtype=None):
        if fading_generator.shape is None or len(fading_generator.shape) != 2:
            raise RuntimeError("The provided fading_generator for the TdlMimoChannel class must have a shape with "
                               "two values")
        super().__init__(fading_generator, channel_profile, tap_powers_dB, tap_delays, Ts, engine=engine,
                         dtype=dtype)


class SuChannel:
    """reference singleuser.py:19-359: TdlChannel + optional path loss; default = flat fading."""

    def __init__(self, fading_generator=None, channel_profile=None, tap_powers_dB=None, tap_delays=None, Ts=None,
                 engine=None, dtype=None):
        if fading_generator is None:
            fading_generator = RayleighSampleGenerator()
            if channel_profile is None and Ts is None:
                Ts = 1.0
        if channel_profile is None and tap_powers_dB is None and tap_delays is None:
            self._tdlchannel = TdlChannel(fading_generator, tap_powers_dB=np.zeros(1), tap_delays=np.zeros(1),
                                          Ts=Ts, engine=engine, dtype=dtype)
        else:
            self._tdlchannel = TdlChannel(fading_generator, channel_profile, tap_powers_dB, tap_delays, Ts,
                                          engine=engine, dtype=dtype)
        self._pathloss_value = None

    def set_pathloss(self, pathloss_value=None):
        if pathloss_value is not None and (pathloss_value < 0 or pathloss_value > 1):
            raise ValueError("Pathloss must be between 0 and 1")
        self._pathloss_value = pathloss_value

    num_taps = property(lambda self: self._tdlchannel.num_taps)
    num_taps_with_padding = property(lambda self: self._tdlchannel.num_taps_with_padding)
    channel_profile = property(lambda self: self._tdlchannel.channel_profile)

    def corrupt_data(self, signal):
        out = self._tdlchannel.corrupt_data(signal)
        if self._pathloss_value is not None:
            out = out * math.sqrt(self._pathloss_value)
        return out

    def corrupt_data_in_freq_domain(self, signal, fft_size, carrier_indexes=None):
        """singleuser.py:153-194."""
        out = self._tdlchannel.corrupt_data_in_freq_domain(signal, fft_size, carrier_indexes)
        if self._pathloss_value is not None:
            out = out * math.sqrt(self._pathloss_value)
        return out

    def set_num_antennas(self, num_rx_antennas, num_tx_antennas):
        """singleuser.py:112-128."""
        self._tdlchannel.set_num_antennas(num_rx_antennas, num_tx_antennas)

    num_tx_antennas = property(lambda self: self._tdlchannel.num_tx_antennas)
    num_rx_antennas = property(lambda self: self._tdlchannel.num_rx_antennas)

    @property
    def switched_direction(self):
        return self._tdlchannel.switched_direction

    @switched_direction.setter
    def switched_direction(self, value):
        self._tdlchannel.switched_direction = value

    def get_last_impulse_response(self):
        ir = self._tdlchannel.get_last_impulse_response()
        if self._pathloss_value is None:
            return ir
        return TdlImpulseResponse(ir.tap_values_sparse * math.sqrt(self._pathloss_value), ir.channel_profile)


class SuMimoChannel(SuChannel):
    """reference singleuser.py:154-359: SuChannel whose generator carries (N, N) antennas."""

    def __init__(self, num_antennas, fading_generator=None, channel_profile=None, tap_powers_dB=None,
                 tap_delays=None, Ts=None, engine=None, dtype=None):
        if fading_generator is None:
            fading_generator = RayleighSampleGenerator()
            if channel_profile is None and Ts is None:
                Ts = 1.0
        fading_generator.shape = (num_antennas, num_antennas)
        super().__init__(fading_generator, channel_profile, tap_powers_dB, tap_delays, Ts, engine=engine,
                         dtype=dtype)
