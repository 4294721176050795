"""pyphysim_amd -- MI355X-native Monte Carlo link-level engine behind pyphysim's operator surface.

The arithmetic lives in hand-written HIP kernels (pyphysim_amd/csrc -> libmcle.so) reached
through the C ABI of include/mcle.h.  Importing this package does not touch the GPU; the first
compute call loads the library and fails loudly if it (or a gfx950 device) is missing.
"""
from . import _lib  # noqa: F401
from ._lib import McleError  # noqa: F401

__version__ = "0.1.0"
