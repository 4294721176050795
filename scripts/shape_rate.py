"""usage: shape_rate.py fft nt nr f64_threads [grid_oversub ...] -- realizations/s of one shape of config 4's family (64-QAM, 25 dB; env DT=f32|f64, default f64)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
fft, nt, nr, thr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
DT = os.environ.get("DT", "f64")
eng = Engine(0, DT)
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
n = 262144
cnt = eng.new_counters()
for g in [int(v) for v in sys.argv[5:]] or [0]:
    with eng.options(f64_threads=thr, grid_oversub=g):
        run = lambda first: eng.run_mimo_ofdm(nt, nr, fft, 16, fft, 1, 10 ** -2.5, 1, first, n, method=_lib.DEMOD_MINDIST, dtype=DT, counters=cnt)
        run(1 << 30); eng.sync(); eng.timer_start()
        for s in range(5): run(s * n)
        ms = eng.timer_stop_ms() / 5
    print(DT, fft, nt, nr, "f64_threads", thr, "grid_oversub", g, "%.3f ms" % ms, "%.3e /s" % (n / ms * 1e3))
