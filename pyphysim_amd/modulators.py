"""Host mirror of pyphysim.modulators for the hot path: QAM / PSK / QPSK / BPSK, OFDM and
OfdmOneTapEqualizer with the reference's names, arguments and error behaviour
(reference modulators/fundamental.py, modulators/ofdm.py).  Constellation tables and argument
checks are host-side setup; modulate / demodulate / FFTs run in the HIP kernels of libmcle
(no NumPy fallback: without the library or a GPU every compute call raises McleError).

Arrays: pass NumPy arrays for drop-in behaviour (host -> device -> host per call, complex128 /
int64 like the reference), or :class:`pyphysim_amd.engine.DeviceArray` to stay on the GPU.
"""
import math

import numpy as np
from scipy.special import erfc

from . import _lib
from .engine import DeviceArray, get_engine

PI = math.pi


def qfunc(x):
    """Gaussian tail function Q(x) (reference util/misc.py:569-592)."""
    return 0.5 * erfc(np.asarray(x) / math.sqrt(2.0))


def dB2Linear(value_dB):
    """reference util/conversion.py:139-158."""
    return pow(10, np.asarray(value_dB, dtype=float) / 10.0)


def level2bits(n):
    """reference util/misc.py:392-414."""
    if n < 1:
        raise ValueError("level2bits: n must be greater then one")
    return max(int(n - 1).bit_length(), 1)


def _gray(n):
    n = np.asarray(n)
    return n ^ (n >> 1)


def _gray_inverse(g):
    g = np.asarray(g).copy()
    shift = 1
    while shift < 32:
        g ^= g >> shift
        shift <<= 1
    return g


def constellation(kind, M=None, phase_offset=0.0):
    """Build a constellation table the way the reference's classes do.

    'qam': square grid, re = -(L-1)+2*col, im = (L-1)-2*row, / sqrt(2(M-1)/3); the label whose
           high half-bits are r and low half-bits are c sits at row gray(r), column gray(c)
           (fundamental.py:697-777).
    'psk': exp(j(2 pi m / M + offset)) with |re|,|im| < 1e-15 snapped to 0, label m holding the
           point gray^-1(m) (fundamental.py:396-448).   'qpsk' = psk(4, pi/4).   'bpsk' = [1, -1].
    """
    if kind == "qam":
        bits = math.log(M, 2)
        if (bits % 2 != 0) or (2 ** bits != M):
            raise ValueError("M must be a square power of 2")
        L = int(round(math.sqrt(M)))
        half = int(round(bits)) // 2
        labels = np.arange(M)
        row, col = _gray(labels >> half), _gray(labels & (L - 1))
        pts = (-(L - 1) + 2.0 * col) + 1j * ((L - 1) - 2.0 * row)
        return pts / math.sqrt((M - 1) * 2.0 / 3.0)
    if kind == "psk":
        if 2 ** math.log(M, 2) != M:
            raise AssertionError("M must be a power of 2")
        m = _gray_inverse(np.arange(M))
        ang = 2.0 * PI / M * m + phase_offset
        re, im = np.cos(ang), np.sin(ang)
        re[np.abs(re) < 1e-15] = 0
        im[np.abs(im) < 1e-15] = 0
        return re + 1j * im
    if kind == "qpsk":
        return constellation("psk", 4, PI / 4.0)
    if kind == "bpsk":
        return np.array([1.0 + 0j, -1.0 + 0j])
    raise ValueError("unknown constellation kind %r" % (kind,))


class Modulator:
    """Base class (reference fundamental.py:30-390): table look-up modulation and minimum
    distance demodulation, executed by libmcle."""
    _kind = _lib.CONST_GENERIC

    def __init__(self, engine=None, dtype=None):
        self._M = 0
        self._K = 0
        self.symbols = np.array([])
        self._engine = engine
        self.dtype = dtype          # None: engine default (f64, the reference's arithmetic)

    # -- properties of the reference -------------------------------------------------------
    @property
    def name(self):
        return "{0:d}-{1:s}".format(self._M, self.__class__.__name__)

    @property
    def M(self):
        return self._M

    @property
    def K(self):
        return self._K

    def __repr__(self):
        return "{0} object".format(self.name)

    def setConstellation(self, symbols):
        symbols = np.asarray(symbols)
        self._M = symbols.size
        self._K = np.log2(self._M)
        self.symbols = symbols

    # -- engine plumbing -------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def _bind(self):
        eng = self.engine
        eng.set_constellation(np.asarray(self.symbols, dtype=complex), self._kind)
        return eng

    # -- the operators ---------------------------------------------------------------------
    def modulate(self, inputData):
        """symbols[inputData]; ValueError for indices >= M (fundamental.py:196-199)."""
        eng = self._bind()
        if isinstance(inputData, DeviceArray):
            return eng.modulate(inputData, dtype=self.dtype)
        arr = np.asarray(inputData)
        out = eng.modulate(arr.reshape(-1), dtype=self.dtype).reshape(arr.shape)
        return out[()] if arr.ndim == 0 else out

    def demodulate(self, receivedData, method=None):
        """argmin_m |symbols[m] - r| with numpy's first-minimum tie rule (fundamental.py:241-246)."""
        eng = self._bind()
        method = _lib.DEMOD_MINDIST if method is None else method
        if isinstance(receivedData, DeviceArray):
            return eng.demodulate(receivedData, method=method, dtype=self.dtype)
        arr = np.asarray(receivedData)
        return eng.demodulate(arr.reshape(-1), method=method, dtype=self.dtype).reshape(arr.shape)

    # -- theory (host-side closed forms; off the hot path) ----------------------------------
    def calcTheoreticalSER(self, SNR):
        raise NotImplementedError("calcTheoreticalSER")

    def calcTheoreticalBER(self, SNR):
        raise NotImplementedError("calcTheoreticalBER")

    def calcTheoreticalPER(self, SNR, packet_length):
        return 1 - ((1 - self.calcTheoreticalBER(SNR)) ** packet_length)

    def calcTheoreticalSpectralEfficiency(self, SNR, packet_length=None):
        if packet_length is None:
            return self.K * (1 - self.calcTheoreticalBER(SNR))
        return self.K * (1 - self.calcTheoreticalPER(SNR, packet_length))


class PSK(Modulator):
    """reference fundamental.py:393-503."""

    def __init__(self, M, phaseOffset=0, engine=None, dtype=None):
        super().__init__(engine, dtype)
        self.setConstellation(constellation("psk", M, phaseOffset))

    def setPhaseOffset(self, phaseOffset):
        # the reference re-creates the table WITHOUT the Gray permutation here (fundamental.py:459)
        M = self._M
        ang = 2.0 * PI / M * np.arange(0, M) + phaseOffset
        re, im = np.cos(ang), np.sin(ang)
        re[np.abs(re) < 1e-15] = 0
        im[np.abs(im) < 1e-15] = 0
        self.setConstellation(re + 1j * im)

    def calcTheoreticalSER(self, SNR):
        return 2.0 * qfunc(np.sqrt(2.0 * dB2Linear(SNR)) * math.sin(PI / self._M))

    def calcTheoreticalBER(self, SNR):
        return 1.0 / level2bits(self._M) * self.calcTheoreticalSER(SNR)


class QPSK(PSK):
    """reference fundamental.py:510-525."""

    def __init__(self, engine=None, dtype=None):
        super().__init__(4, PI / 4.0, engine, dtype)

    def __repr__(self):
        return "QPSK object"


class BPSK(Modulator):
    """reference fundamental.py:534-647: 0 -> +1, 1 -> -1; demodulate = (re < 0)."""
    _kind = _lib.CONST_BPSK

    def __init__(self, engine=None, dtype=None):
        super().__init__(engine, dtype)
        self.setConstellation(np.array([1, -1]))

    @property
    def name(self):
        return "{0:s}".format(self.__class__.__name__)

    def __repr__(self):
        return "BPSK object"

    def modulate(self, inputData):
        if not isinstance(inputData, DeviceArray) and np.any(np.asarray(inputData) > 1):
            raise ValueError("Input data can only contains '0's and '1's")
        out = super().modulate(inputData)
        if isinstance(out, DeviceArray):
            return out
        return out.real        # the reference returns 1 - 2*idx (real)

    def calcTheoreticalSER(self, SNR):
        return qfunc(np.sqrt(2 * dB2Linear(SNR)))

    def calcTheoreticalBER(self, SNR):
        return self.calcTheoreticalSER(SNR)


class QAM(Modulator):
    """reference fundamental.py:656-857 (square, Gray mapped).  ``demodulate(..., method=
    DEMOD_QAM_SLICER)`` selects the per-axis slicer kernel (same decisions, ~M/4 x less work)."""
    _kind = _lib.CONST_QAM

    def __init__(self, M, engine=None, dtype=None):
        super().__init__(engine, dtype)
        self.setConstellation(constellation("qam", M))

    def _psc(self, SNR):
        return 2.0 * (1.0 - 1.0 / np.sqrt(self._M)) * qfunc(np.sqrt(dB2Linear(SNR) * 3.0 / (self._M - 1.0)))

    def calcTheoreticalSER(self, SNR):
        return 1 - (1 - self._psc(SNR)) ** 2

    def calcTheoreticalBER(self, SNR):
        return (2.0 * self._psc(SNR)) / level2bits(self._M)


class OFDM:
    """reference modulators/ofdm.py:16-466.  IFFT/FFT, subcarrier mapping, power scale and cyclic
    prefix run in one LDS-resident kernel per OFDM symbol."""

    def __init__(self, fft_size, cp_size, num_used_subcarriers=None, engine=None, dtype=None):
        self.fft_size = 0
        self.cp_size = 0
        self.num_used_subcarriers = 0
        self._engine = engine
        self.dtype = dtype
        self.set_parameters(fft_size, cp_size, num_used_subcarriers)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def set_parameters(self, fft_size, cp_size, num_used_subcarriers=None):
        if (cp_size < 0) or cp_size > fft_size:
            raise ValueError("cp_size must be nonnegative and cannot be greater than fft_size")
        if num_used_subcarriers is None:
            num_used_subcarriers = fft_size
        if num_used_subcarriers > fft_size:
            raise ValueError("Number of used subcarriers cannot be greater than the fft_size")
        if (num_used_subcarriers % 2 != 0) or (num_used_subcarriers < 2):
            raise ValueError("Number of used subcarriers must be a multiple of 2")
        self.fft_size, self.cp_size, self.num_used_subcarriers = fft_size, cp_size, num_used_subcarriers

    def _calc_zeropad(self, input_data_size):
        n_sym = int(np.ceil(float(input_data_size) / self.num_used_subcarriers))
        return self.num_used_subcarriers * n_sym - input_data_size, n_sym

    def get_used_subcarrier_indexes(self):
        """ofdm.py:188-224 (host helper; the kernels compute the same map on the fly)."""
        n, u = self.fft_size, self.num_used_subcarriers
        if u == n:
            return np.concatenate([np.arange(n // 2, n), np.arange(0, n // 2)])
        return np.concatenate([np.arange(n - u // 2, n), np.arange(1, u // 2 + 1)])

    def _calculate_power_scale(self):
        return float(self.fft_size) ** 2 / (float(self.num_used_subcarriers) + self.cp_size)

    def modulate(self, input_signal):
        if isinstance(input_signal, DeviceArray):
            return self.engine.ofdm_modulate(input_signal, self.fft_size, self.cp_size, self.num_used_subcarriers,
                                             dtype=self.dtype)
        x = np.asarray(input_signal).reshape(-1)
        return self.engine.ofdm_modulate(x, self.fft_size, self.cp_size, self.num_used_subcarriers,
                                         dtype=self.dtype).reshape(-1)

    def demodulate(self, received_signal):
        if isinstance(received_signal, DeviceArray):
            return self.engine.ofdm_demodulate(received_signal, self.fft_size, self.cp_size,
                                               self.num_used_subcarriers, dtype=self.dtype)
        y = np.asarray(received_signal).reshape(-1)
        n_sym = y.size // (self.fft_size + self.cp_size)
        return self.engine.ofdm_demodulate(y[:n_sym * (self.fft_size + self.cp_size)], self.fft_size, self.cp_size,
                                           self.num_used_subcarriers, dtype=self.dtype).reshape(-1)


class OfdmOneTapEqualizer:
    """reference modulators/ofdm.py:469-552.  ``impulse_response`` is anything with
    ``tap_values_sparse`` [taps, samples] and ``tap_indexes_sparse`` (e.g.
    pyphysim_amd.channels.TdlImpulseResponse)."""

    def __init__(self, ofdm_obj):
        self._ofdm_obj = ofdm_obj

    def equalize_data(self, data, impulse_response):
        o = self._ofdm_obj
        data = np.asarray(data).reshape(-1)
        return o.engine.onetap_equalize(data, impulse_response.tap_values_sparse, impulse_response.tap_indexes_sparse,
                                        o.fft_size, o.cp_size, o.num_used_subcarriers, dtype=o.dtype).reshape(-1)
