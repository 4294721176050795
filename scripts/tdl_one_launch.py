#!/usr/bin/env python3
"""One launch of config 3 per (fft_size, tdl_kernel option) given on the command line as fft:kernel[:dtype] ... -- the subject of a
rocprofv3 --pmc pass (scripts/tdl_hw_pmc.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.channels import discretize_profile  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

for spec in sys.argv[1:]:
    parts = spec.split(":")
    fft, kern, dt = int(parts[0]), int(parts[1]), (parts[2] if len(parts) > 2 else "f32")
    eng = Engine(0, dt)
    eng.set_constellation(constellation("qpsk", 4), _lib.CONST_GENERIC)
    Ts = 1.0 / (15e3 * fft)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    n = 1 << 18
    with eng.options(tdl_kernel=kern):
        eng.run_ofdm_tdl(fft, 16, fft, 1, 0.01, p_lin, d_idx, 1, 0, n, Fd=10.0, Ts=Ts, L=8, dtype=dt)
    eng.sync()
    eng.close()
