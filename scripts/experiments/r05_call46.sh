#!/bin/bash
# round 5, call 46: k_run_mimo_ofdm_planar at 512 points with four receive antennas: two antennas per thread (256 threads, default)
# against four antennas per thread (128 threads, f64_threads = 128), both arithmetics, 4 x 4 and 2 x 4
export TMPDIR=/tmp
python - <<'PY'
import sys
sys.path.insert(0, '.')
from pyphysim_amd import _lib
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
for dt in ("f64", "f32"):
    eng = Engine(0, dt)
    eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
    n = 262144
    for NT in (4, 2):
        for rnd in (1, 2):
            for thr in (0, 128):
                cnt = eng.new_counters()
                with eng.options(f64_threads=thr):
                    run = lambda first: eng.run_mimo_ofdm(NT, 4, 512, 16, 512, 1, 10 ** -2.5, 1, first, n, method=_lib.DEMOD_MINDIST, dtype=dt, counters=cnt)
                    run(1 << 30); eng.sync(); eng.timer_start()
                    for s in range(4): run(s * n)
                    ms = eng.timer_stop_ms() / 4
                print("planar 512 %dx4 %s f64_threads=%d: %.3f ms per %d = %.4g /s" % (NT, dt, thr, ms, n, n / ms * 1e3))
PY
