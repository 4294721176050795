#!/usr/bin/env python3
"""OFDM operator kernels outside the complex64 / 1024 case: complex128 (the drop-in default) and another FFT size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyphysim_amd.engine import Engine  # noqa: E402

for dt, fft, n in (("f64", 1024, 1 << 25), ("f64", 256, 1 << 25), ("f32", 512, 1 << 26), ("f32", 4096, 1 << 26)):
    eng = Engine(0, dt)
    x = eng.randn_c(n, 1, 2, device=True)
    size = 16 if dt == "f64" else 8
    cp = fft // 64
    nsym = n // fft

    def timed(fn):
        fn()
        eng.sync()
        eng.timer_start()
        for _ in range(5):
            fn()
        return eng.timer_stop_ms() / 5

    ms_m = timed(lambda: eng.ofdm_modulate(x, fft, cp, fft))
    t = eng.ofdm_modulate(x, fft, cp, fft)
    ms_d = timed(lambda: eng.ofdm_demodulate(t, fft, cp, fft))
    nbytes = n * size + nsym * (fft + cp) * size
    print("%s fft %4d: modulate %.4f ms (%.2f of 8 TB/s)  demodulate %.4f ms (%.2f)" % (
        dt, fft, ms_m, nbytes / ms_m / 1e6 / 8000, ms_d, nbytes / ms_d / 1e6 / 8000))
    del eng
