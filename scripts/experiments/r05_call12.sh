#!/bin/bash
# round 5, call 12: f1 wavefront kernels after the 16-sample channel rounds (2048) and the un-hoisted sign pairs; the 2048-point
# planar kernels with four antennas per thread (512 threads, nothing spilled) against two (1 024 threads, 36 / 22 spilled)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mimo_tdl_wave.py -m gpu -q --timeout=600 -x 2>&1 | tail -4
for dt in f32 f64; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype $dt --config f1 --batch 98304 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f1 $dt', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
from pyphysim_amd import _lib
eng = Engine(0, "f64")
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
nv = 10 ** -2.5
for dt in ("f64", "f32"):
    for nt in (4, 3):
        for thr in (0, 512):
            with eng.options(f64_threads=thr):
                n = 65536
                c = eng.new_counters()
                eng.run_mimo_ofdm(nt, 4, 2048, 16, 2048, 1, nv, 7, 0, n, dtype=dt, counters=c); eng.sync()
                eng.timer_start()
                for s in range(4):
                    eng.run_mimo_ofdm(nt, 4, 2048, 16, 2048, 1, nv, 7, (s + 1) * n, n, dtype=dt, counters=c)
                ms = eng.timer_stop_ms() / 4
                r = eng.read_counters(c)
                print("planar 2048 %dx4 %s f64_threads=%d: %.3f ms per %d = %.4g /s  ser %.6f" % (nt, dt, thr, ms, n, n / ms * 1e3, r["sym_errors"] / (r["n_realizations"] * nt * 2048.0)))
PY
