#!/usr/bin/env python3
"""Realizations/s of the fused frequency-selective MIMO-OFDM link (k_run_mimo_ofdm_tdl_wave) per size and geometry, both
arithmetics: 64-QAM, cp 16 (32 at 2048), five Jakes taps at delays 0 .. 4 (Fd 10 Hz), SNR 25 dB, slicer.  One line per case."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
p = 10.0 ** (-0.3 * __import__("numpy").arange(5))
p = p / p.sum()
for dt in ("f32", "f64"):
    eng = Engine(0, dt)
    eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
    for fft, nt, nr in ((256, 4, 4), (512, 4, 4), (1024, 2, 2), (1024, 2, 4), (1024, 4, 4), (2048, 2, 2), (2048, 4, 4)):
        n = max(8192, int(98304 * 1024 * 4 / (fft * nr)) // 8192 * 8192)
        n = min(n, 196608)
        cnt = eng.new_counters()
        run = lambda first: eng.run_mimo_ofdm_tdl(nt, nr, fft, 16 if fft < 2048 else 32, fft, 1, 10 ** -2.5, p, list(range(5)), 1,
                                                  first, n, Ts=1.0 / (15e3 * fft), method=_lib.DEMOD_QAM_SLICER, dtype=dt, counters=cnt)
        _t0, _k = __import__('time').time(), 0        # untimed launches until the device has been busy for 60 ms (timed cold, short
        while _k < 64 and (_k < 1 or __import__('time').time() - _t0 < 0.06):   # launches read 5 - 10 % low: DESIGN 6, round 6)
            run((1 << 30) + _k * n)
            eng.sync()
            _k += 1
        eng.timer_start()
        for s in range(3):
            run(s * n)
        ms = eng.timer_stop_ms() / 3
        print("%s f1 %s fft %d %dx%d: %.3f ms per %d = %.4g /s" % (tag, dt, fft, nt, nr, ms, n, n / ms * 1e3))
