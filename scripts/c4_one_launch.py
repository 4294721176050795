#!/usr/bin/env python3
"""One launch of the MIMO-OFDM link over the flat channel (config 4's family) per fft:nt:nr:dtype given on the command line (64-QAM, prefix
16, full band, 25 dB, min-distance) -- the subject of a rocprofv3 --pmc pass (scripts/c4_pmc.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

for spec in sys.argv[1:]:
    fft, nt, nr, dt = spec.split(":")
    eng = Engine(0, dt)
    eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
    eng.run_mimo_ofdm(int(nt), int(nr), int(fft), 16, int(fft), 1, 10 ** -2.5, 1, 0, 1 << 16, method=_lib.DEMOD_MINDIST, dtype=dt)
    eng.sync()
    eng.close()
