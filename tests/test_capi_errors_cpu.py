"""CPU: the argument-error paths of the C ABI that need no device -- every entry point that takes a context refuses a NULL
one with an error code and a message (never a crash), the context constructor reports the missing device, the pure-host
entry points (mcle_build_demod_grid, mcle_device_count, mcle_last_error) work.  scripts/asan_host.sh runs this file on the
AddressSanitizer build of the library."""
import ctypes
from ctypes import POINTER, c_char_p, c_double, c_int, c_size_t, c_uint64, c_void_p

import numpy as np
import pytest

from pyphysim_amd import _lib

NO_CTX_NEEDED = {"mcle_last_error", "mcle_version", "mcle_device_count", "mcle_ctx_create", "mcle_build_demod_grid",
                 "mcle_comm_load"}
# a NULL context is a documented no-op for these two (include/mcle.h): they return MCLE_OK
NULL_IS_OK = {"mcle_ctx_destroy"}


def _zero_for(argtype):
    if argtype in (c_void_p, c_char_p) or (isinstance(argtype, type) and issubclass(argtype, ctypes._Pointer)):
        return None
    return argtype(0)


def test_every_context_entry_point_refuses_a_null_context():
    lib = _lib.load()
    checked = 0
    for name, (res, args) in sorted(_lib._PROTOS.items()):
        if name in NO_CTX_NEEDED or not args or args[0] is not c_void_p:
            continue
        rc = getattr(lib, name)(*[_zero_for(a) for a in args])
        if name in NULL_IS_OK:
            assert rc == 0, name
        else:
            assert rc != 0, "%s accepted a NULL context" % name
            msg = lib.mcle_last_error()
            assert msg and len(msg) > 3, name
        checked += 1
    assert checked >= 75


def test_context_constructor_without_a_device_or_with_a_bad_index():
    lib = _lib.load()
    n = c_int(-1)
    assert lib.mcle_device_count(ctypes.byref(n)) == 0 and n.value >= 0
    out = c_void_p()
    assert lib.mcle_ctx_create(0, None) != 0                                   # null output pointer
    if n.value == 0:
        assert lib.mcle_ctx_create(0, ctypes.byref(out)) != 0 and not out.value
        assert b"no HIP device" in lib.mcle_last_error()
    else:
        assert lib.mcle_ctx_create(n.value + 3, ctypes.byref(out)) != 0 and not out.value
        assert b"out of range" in lib.mcle_last_error()


def test_demod_grid_builder_on_the_host():
    """mcle_build_demod_grid is pure host code: 64-QAM -> a grid whose every cell lists its candidates, argument errors
    reported (M out of range, null pointers)."""
    lib = _lib.load()
    from pyphysim_amd.modulators import constellation
    c = np.ascontiguousarray(constellation("qam", 64)).view(np.float64)
    G = c_int(0)
    x0, y0, h = c_double(0), c_double(0), c_double(0)
    cells = (c_uint64 * (32 * 32))()
    rc = lib.mcle_build_demod_grid(c.ctypes.data_as(POINTER(c_double)), 64, ctypes.byref(G), ctypes.byref(x0),
                                   ctypes.byref(y0), ctypes.byref(h), cells)
    assert rc == 0 and 2 <= G.value <= 32 and h.value > 0
    assert all(cells[i] != 0 for i in range(G.value * G.value))
    assert lib.mcle_build_demod_grid(None, 64, ctypes.byref(G), ctypes.byref(x0), ctypes.byref(y0), ctypes.byref(h), cells) != 0
    assert lib.mcle_build_demod_grid(c.ctypes.data_as(POINTER(c_double)), 1, ctypes.byref(G), ctypes.byref(x0),
                                     ctypes.byref(y0), ctypes.byref(h), cells) != 0
    assert lib.mcle_build_demod_grid(c.ctypes.data_as(POINTER(c_double)), 512, ctypes.byref(G), ctypes.byref(x0),
                                     ctypes.byref(y0), ctypes.byref(h), cells) != 0
