"""Stand-in for the `numba` import in the reference's util/misc.py (container-only).

TEST INFRASTRUCTURE.  Only used by oracle/make_golden.py and the container-only
oracle-vs-reference checks, so that /root/reference can be imported read-only.
`vectorize` falls back to numpy's element-wise wrapper: same results, no JIT.
"""
import numpy as _np


def vectorize(*signatures, **_kw):
    # used both bare (@numba.vectorize) and called (@numba.vectorize([...]))
    if len(signatures) == 1 and callable(signatures[0]):
        return _np.vectorize(signatures[0], otypes=[_np.int64])

    def wrap(func):
        return _np.vectorize(func)
    return wrap


def jit(*args, **_kw):
    if len(args) == 1 and callable(args[0]):
        return args[0]
    return lambda f: f


njit = jit
