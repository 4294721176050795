#!/usr/bin/env python3
"""Realizations/s of the fused flat-fading MIMO application (apps/mimo/simulate_mimo.py) per scheme:
NSymbs = 200 symbols per layer, 16-QAM, SNR 15 dB, slicer demodulation."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

eng = Engine(0, "f32")
eng.set_constellation(constellation("qam", 16), _lib.CONST_QAM)
out = {}
for scheme, nt, nr in (("blast", 4, 4), ("blast", 2, 2), ("mrc", 1, 4), ("mrt", 4, 1), ("alamouti", 2, 2), ("svd", 4, 4),
                       ("gmd", 4, 4)):
    n = 262144
    cnt = eng.new_counters()
    run = lambda first: eng.run_mimo_flat(scheme, nt, nr, 200, 10 ** -1.5, 1, first, n, method=_lib.DEMOD_QAM_SLICER,
                                          counters=cnt)
    run(1 << 30)
    eng.sync()
    eng.timer_start()
    for s in range(3):
        run(s * n)
    ms = eng.timer_stop_ms() / 3
    c = eng.read_counters(cnt)
    layers = 1 if scheme in ("mrt", "alamouti") else nt
    out["%s_%dx%d" % (scheme, nr, nt)] = dict(realizations_per_s=n / (ms * 1e-3), kernel_ms=ms,
                                              ser=c["sym_errors"] / float(c["n_realizations"] * 200 * layers))
print(json.dumps(out))
