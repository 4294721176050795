#!/usr/bin/env python3
"""Realizations/s of config 4's complex64 family at EVERY geometry 1 <= Nt <= Nr <= 4 x fft_size 256 .. 2048 (64-QAM, prefix 16, full band,
25 dB, min-distance): one line per shape.  Used for the A/B of the wavefronts-per-SIMD bounds (MCLE_LIBRARY selects the build)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

DT = sys.argv[1] if len(sys.argv) > 1 else "f32"
eng = Engine(0, DT)
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
out = {}
for fft in (256, 512, 1024, 2048):
    for nr in (2, 3, 4):
        for nt in range(1, nr + 1):
            n = min(262144, max(16384, int(262144 * 4096 / (fft * nr)) // 8192 * 8192))
            cnt = eng.new_counters()
            run = lambda first: eng.run_mimo_ofdm(nt, nr, fft, 16, fft, 1, 10 ** -2.5, 1, first, n, method=_lib.DEMOD_MINDIST, dtype=DT, counters=cnt)
            _t0, _k = __import__('time').time(), 0        # untimed launches until the device has been busy for 60 ms (timed cold, short
            while _k < 64 and (_k < 1 or __import__('time').time() - _t0 < 0.06):   # launches read 5 - 10 % low: DESIGN 6, round 6)
                run((1 << 30) + _k * n)
                eng.sync()
                _k += 1
            eng.timer_start()
            for s in range(3):
                run(s * n)
            ms = eng.timer_stop_ms() / 3
            out["%dx%dx%d" % (fft, nt, nr)] = n / (ms * 1e-3)
print(json.dumps(out))
