#!/usr/bin/env python3
"""One launch of the frequency-selective MIMO-OFDM link per fft:dtype given on the command line (4 x 4, 64-QAM, five taps, slicer) --
the subject of a rocprofv3 --pmc pass (scripts/f1_pmc.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

p = 10.0 ** (-0.3 * np.arange(5))
p = p / p.sum()
for spec in sys.argv[1:]:
    fft, dt = spec.split(":")
    fft = int(fft)
    eng = Engine(0, dt)
    eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
    n = 1 << 16
    eng.run_mimo_ofdm_tdl(4, 4, fft, 16 if fft < 2048 else 32, fft, 1, 10 ** -2.5, p, list(range(5)), 1, 0, n, Ts=1.0 / (15e3 * fft),
                          method=_lib.DEMOD_QAM_SLICER, dtype=dt)
    eng.sync()
    eng.close()
