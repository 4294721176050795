#!/bin/bash
# round 5, call 41: k_run_mimo_ofdm_planar at 2048 points, 2 x 4, both arithmetics: 1 024 threads (two antennas per thread; complex128
# spills 12 registers at its 128-register bound) against 512 threads (four antennas per thread, 256 registers)
export TMPDIR=/tmp
python - <<'PY'
import sys
sys.path.insert(0, '.')
from pyphysim_amd import _lib
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
for NT, dt in ((1, "f64"), (1, "f32")):
    eng = Engine(0, dt)
    eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
    n = 65536
    for rnd in (1, 2):
        for thr in (0, 512):
            cnt = eng.new_counters()
            with eng.options(f64_threads=thr):
                run = lambda first: eng.run_mimo_ofdm(NT, 4, 2048, 16, 2048, 1, 10 ** -2.5, 1, first, n, method=_lib.DEMOD_MINDIST, dtype=dt, counters=cnt)
                run(1 << 30); eng.sync(); eng.timer_start()
                for s in range(4): run(s * n)
                ms = eng.timer_stop_ms() / 4
            res = eng.read_counters(cnt) if hasattr(eng, "read_counters") else None
            print("planar 2048 %dx4 %s f64_threads=%d: %.3f ms per %d = %.4g /s" % (NT, dt, thr, ms, n, n / ms * 1e3))
PY
