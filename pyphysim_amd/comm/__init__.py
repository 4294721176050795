"""Host mirror of pyphysim.comm for the hot path: block diagonalisation and water-filling
(reference comm/blockdiagonalization.py, comm/waterfilling.py), executed by libmcle's HIP kernels."""
from . import blockdiagonalization, waterfilling

__all__ = ["blockdiagonalization", "waterfilling"]
