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
