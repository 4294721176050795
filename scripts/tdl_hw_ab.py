#!/usr/bin/env python3
"""Config 3 at fft_size 2048: the two-wavefronts-per-realization kernel (csrc/siso_tdl_hw.hpp, default since round 6) against the
one-wavefront kernel (option tdl_kernel = 3) and the 1024 sibling, realizations/s and per-subcarrier ratio.  One JSON object on stdout
(profiles/r06/tdl_hw_ab.json)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.channels import discretize_profile  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

out = {}
for dt in ("f32", "f64"):
    eng = Engine(0, dt)
    eng.set_constellation(constellation("qpsk", 4), _lib.CONST_GENERIC)
    for fft, kern, name in ((1024, 2, "1024"), (2048, 2, "2048_half_wave"), (2048, 3, "2048_one_wave")):
        Ts = 1.0 / (15e3 * fft)
        p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
        n = (1 << 20) * 1024 // fft // (2 if dt == "f64" else 1)
        cnt = eng.new_counters()
        with eng.options(tdl_kernel=kern):
            run = lambda first: eng.run_ofdm_tdl(fft, 16, fft, 1, 0.01, p_lin, d_idx, 1, first, n, Fd=10.0, Ts=Ts, L=8, dtype=dt, counters=cnt)
            _t0, _k = __import__('time').time(), 0        # untimed launches until the device has been busy for 60 ms (timed cold, short
            while _k < 64 and (_k < 1 or __import__('time').time() - _t0 < 0.06):   # launches read 5 - 10 % low: DESIGN 6, round 6)
                run((1 << 30) + _k * n)
                eng.sync()
                _k += 1
            eng.timer_start()
            for s in range(4):
                run(s * n)
            ms = eng.timer_stop_ms() / 4
        c = eng.read_counters(cnt)
        out["%s_%s" % (dt, name)] = dict(realizations_per_launch=n, ms_per_launch=ms, realizations_per_s=n / (ms * 1e-3),
                                         ser=c["sym_errors"] / float(c["n_realizations"] * fft))
    for name in ("2048_half_wave", "2048_one_wave"):
        out["%s_%s" % (dt, name)]["per_subcarrier_vs_1024"] = 2.0 * out["%s_%s" % (dt, name)]["realizations_per_s"] / out[dt + "_1024"]["realizations_per_s"]
print(json.dumps(out, indent=1))
