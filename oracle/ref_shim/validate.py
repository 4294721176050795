"""Stand-in for configobj's `validate` module (container-only; see numba.py here).

The hot path never parses config files; the reference merely imports this at
module import time (simulations/configobjvalidation.py).
"""


class ValidateError(Exception):
    pass


class VdtTypeError(ValidateError):
    def __init__(self, value=None):
        super().__init__("bad type: %r" % (value,))


class VdtValueError(ValidateError):
    pass


class VdtValueTooSmallError(VdtValueError):
    def __init__(self, value=None):
        super().__init__("too small: %r" % (value,))


class VdtValueTooBigError(VdtValueError):
    def __init__(self, value=None):
        super().__init__("too big: %r" % (value,))


def is_float(value, min=None, max=None):
    v = float(value)
    if min is not None and v < float(min):
        raise VdtValueTooSmallError(value)
    if max is not None and v > float(max):
        raise VdtValueTooBigError(value)
    return v


def is_integer(value, min=None, max=None):
    v = int(value)
    if min is not None and v < int(min):
        raise VdtValueTooSmallError(value)
    if max is not None and v > int(max):
        raise VdtValueTooBigError(value)
    return v


class Validator:
    def __init__(self, functions=None):
        self.functions = dict(functions or {})
