"""Host mirror of the pyphysim.util pieces on the hot path (reference util/misc.py,
util/conversion.py): randn_c, count_bit_errors / count_bits, level2bits, dB conversions.

``randn_c`` draws on the GPU under the mcle-philox-v1 contract (see csrc/philox.hpp): the
reference's global MT19937 stream cannot be reproduced on a GPU, so ``seed()`` here selects a
Philox (seed, realization) pair instead and successive calls consume consecutive positions of
the NOISE stream.  Same-seed equality with the reference is obtained through the injected-input
operators (tests/test_gpu_operators.py), not through this generator.
"""
import numpy as np

from . import _lib
from .engine import DeviceArray, get_engine

_state = {"seed": 0, "realization": 0, "pos": 0}


def seed(seed_value, realization=0):
    """Select the Philox key / subsequence used by randn_c and randint_symbols."""
    _state.update(seed=int(seed_value) & 0xFFFFFFFFFFFFFFFF, realization=int(realization), pos=0, dpos=0)


def randn_c(*args, engine=None, dtype=None, device=False):
    """Circularly symmetric complex Gaussian array with unit variance (misc.py:327-355)."""
    eng = engine or get_engine()
    n = int(np.prod(args)) if args else 1
    out = eng.randn_c(n, _state["seed"], _state["realization"], _lib.STREAM_NOISE, first=_state["pos"],
                      dtype=dtype, device=device)
    _state["pos"] += n
    if device:
        return out.reshape(*args) if args else out
    return out.reshape(args) if args else out[0]


def randint_symbols(M, n, engine=None, device=False):
    """n symbol indices uniform in [0, M) (the np.random.randint(0, M, n) of the reference apps)."""
    eng = engine or get_engine()
    out = eng.rand_symbols(int(n), int(M), _state["seed"], _state["realization"], first=_state.get("dpos", 0),
                           device=device)
    _state["dpos"] = _state.get("dpos", 0) + int(n)
    return out


def level2bits(n):
    """misc.py:392-414."""
    if n < 1:
        raise ValueError("level2bits: n must be greater then one")
    return max(int(n - 1).bit_length(), 1)


def count_bit_errors(first, second, axis=None, engine=None):
    """Number of differing bits between two index arrays (misc.py:519-566).  The total (axis=None)
    and per-row (axis=1 of a 2-D input, or axis=-1) sums run on the GPU; other axes transpose on
    the host first."""
    eng = engine or get_engine()
    a, b = np.asarray(first), np.asarray(second)
    a, b = np.broadcast_arrays(a, b)
    if axis is None:
        _, _, be = eng.count_errors(a.reshape(-1), b.reshape(-1), 31, 1)
        return int(be[0])
    a = np.moveaxis(a, axis, -1)
    b = np.moveaxis(b, axis, -1)
    rows = int(np.prod(a.shape[:-1])) if a.ndim > 1 else 1
    _, _, be = eng.count_errors(np.ascontiguousarray(a).reshape(-1), np.ascontiguousarray(b).reshape(-1), 31, rows)
    return be.astype(np.int64).reshape(a.shape[:-1])


def count_symbol_errors(first, second, engine=None):
    """sum(first != second), the symbol-error count of the reference's simulators."""
    eng = engine or get_engine()
    a, b = np.asarray(first), np.asarray(second)
    _, se, _ = eng.count_errors(a.reshape(-1), b.reshape(-1), 31, 1)
    return int(se[0])


def count_bits(n):
    """popcount (misc.py:449-476) -- host helper for small arrays."""
    n = np.asarray(n).astype(np.uint64)
    c = np.zeros(n.shape, dtype=np.int64)
    while np.any(n):
        c += (n & np.uint64(1)).astype(np.int64)
        n = n >> np.uint64(1)
    return c


def dB2Linear(valueIndB):
    """conversion.py:139-158."""
    return pow(10, np.asarray(valueIndB, dtype=float) / 10.0) if not np.isscalar(valueIndB) else pow(
        10, valueIndB / 10.0)


def linear2dB(valueInLinear):
    """conversion.py:161-180."""
    return 10.0 * np.log10(valueInLinear)


# ---- the small scalar / bit / bookkeeping helpers of util/conversion.py and util/misc.py ------------------------------
def dBm2Linear(valueIndBm):
    """conversion.py:183-202: dBm -> Watts."""
    return dB2Linear(valueIndBm) / 1000.0


def linear2dBm(valueInLinear):
    """conversion.py:205-226: Watts -> dBm."""
    return linear2dB(valueInLinear * 1000.0)


def SNR_dB_to_EbN0_dB(SNR, bits_per_symb):
    """conversion.py:282-302."""
    return SNR - 10.0 * np.log10(bits_per_symb)


def EbN0_dB_to_SNR_dB(EbN0, bits_per_symb):
    """conversion.py:305-323."""
    return EbN0 + 10.0 * np.log10(bits_per_symb)


def xor(a, b):
    """misc.py:297-324 (ints or integer arrays)."""
    return a ^ b


def binary2gray(num):
    """conversion.py:229-249."""
    return (num >> 1) ^ num


def gray2binary(num):
    """conversion.py:252-279 (codes of up to 16 bits, like the reference's four folding steps)."""
    num = num ^ (num >> 8)
    num = num ^ (num >> 4)
    num = num ^ (num >> 2)
    return num ^ (num >> 1)


def single_matrix_to_matrix_of_matrices(single_matrix, nrows=None, ncols=None):
    """conversion.py:22-136: cut a matrix into an object array of row blocks, column blocks, or K x K blocks."""
    single_matrix = np.asarray(single_matrix)
    if nrows is None:
        cols = np.hstack([0, np.cumsum(ncols)])
        out = np.zeros(len(ncols), dtype=np.ndarray)
        for t in range(len(ncols)):
            out[t] = single_matrix[:, cols[t]:cols[t + 1]]
        return out
    rows = np.hstack([0, np.cumsum(nrows)])
    K = len(nrows)
    if ncols is None:
        out = np.zeros(K, dtype=np.ndarray)
        for r in range(K):
            out[r] = single_matrix[rows[r]:rows[r + 1]]
        return out
    cols = np.hstack([0, np.cumsum(ncols)])
    out = np.zeros([K, K], dtype=np.ndarray)
    for r in range(K):
        for t in range(K):
            out[r, t] = single_matrix[rows[r]:rows[r + 1], cols[t]:cols[t + 1]]
    return out


def int2bits(n):
    """misc.py:417-446: bits needed to write n."""
    if n < 0:
        raise ValueError("int2bits: n must be greater then zero")
    return max(int(n).bit_length(), 1)


def pretty_time(time_in_seconds):
    """misc.py:258-294."""
    minutes, seconds = int(time_in_seconds) // 60, int(round(time_in_seconds % 60))
    hours, minutes = minutes // 60, minutes % 60
    if hours > 0:
        return "%sh:%02dm:%02ds" % (hours, minutes, seconds)
    if minutes > 0:
        return "%sm:%02ds" % (minutes, seconds)
    return "%.2fs" % time_in_seconds


def equal_dicts(a, b, ignore_keys):
    """misc.py:1120-1145."""
    ka, kb = set(a).difference(ignore_keys), set(b).difference(ignore_keys)
    return ka == kb and all(a[k] == b[k] for k in ka)


def calc_shannon_sum_capacity(sinrs):
    """misc.py:1204-1225 (linear SINRs)."""
    return float(np.sum(np.log2(1 + np.asarray(sinrs))))


def calc_unorm_autocorr(x):
    """misc.py:698-724."""
    R = np.correlate(x, x, "full")
    return R[R.size // 2:]


def calc_autocorr(x):
    """misc.py:727-752."""
    x2 = np.asarray(x) - np.mean(x)
    return calc_unorm_autocorr(x2) / (x2.size * float(np.var(x2)))


_CI_TABLE = {50: 0.674, 60: 0.842, 70: 1.036, 80: 1.282, 90: 1.645, 95: 1.960, 98: 2.326, 99: 2.576, 99.5: 2.807,
             99.8: 3.090, 99.9: 3.291}


def calc_confidence_interval(mean, std, n, P=95.0):
    """misc.py:807-867: normal-approximation interval for the confidence levels the reference tabulates."""
    half = _CI_TABLE[P] * std / np.sqrt(n)
    return mean - half, mean + half


def randn_c_RS(RS, *args):
    """misc.py:358-389: randn_c drawn from an explicit NumPy RandomState (host: the object's own stream is the point);
    RS None -> this module's randn_c (device)."""
    if RS is None:
        return randn_c(*args)
    return (1.0 / np.sqrt(2.0)) * (RS.randn(*args) + 1j * RS.randn(*args))
