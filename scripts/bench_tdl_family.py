#!/usr/bin/env python3
"""Realizations/s of config 3 (SISO OFDM over a 5-tap Jakes TDL channel, QPSK, cp 16, 20 dB) per fft_size and arithmetic: the
one-realization-per-wavefront kernel (csrc/siso_tdl_wave.hpp: k_run_ofdm_tdl_wave, default) next to the batched kernels it
replaced (context option tdl_kernel=1).  One JSON object on stdout (profiles/r04/tdl_family_rates.json)."""
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
    for fft in (256, 512, 1024, 2048):
        Ts = 1.0 / (15e3 * fft)
        p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
        n = min(1 << 21, max(1 << 17, (1 << 21) * 1024 // fft // (2 if dt == "f64" else 1)))
        row = {"realizations_per_launch": n}
        for name, kern in (("wave", 2), ("batched", 1)):
            cnt = eng.new_counters()
            with eng.options(tdl_kernel=kern):
                run = lambda first: eng.run_ofdm_tdl(fft, 16, fft, 1, 0.01, p_lin, d_idx, 1, first, n, Fd=10.0, Ts=Ts, L=8,
                                                     dtype=dt, counters=cnt)
                _t0, _k = __import__('time').time(), 0        # untimed launches until the device has been busy for 60 ms (timed cold, short
                while _k < 64 and (_k < 1 or __import__('time').time() - _t0 < 0.06):   # launches read 5 - 10 % low: DESIGN 6, round 6)
                    run((1 << 30) + _k * n)
                    eng.sync()
                    _k += 1
                eng.timer_start()
                for s in range(3):
                    run(s * n)
                ms = eng.timer_stop_ms() / 3
            c = eng.read_counters(cnt)
            row[name] = dict(realizations_per_s=n / (ms * 1e-3), ms_per_launch=ms, ser=c["sym_errors"] / float(c["n_realizations"] * fft))
        row["wave_over_batched"] = row["wave"]["realizations_per_s"] / row["batched"]["realizations_per_s"]
        out["%s_%d" % (dt, fft)] = row
print(json.dumps(out, indent=1))
