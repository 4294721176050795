#!/bin/bash
# round 5, call 42: the planar family suites after the 2048 2 x 4 complex64 default moved to the 512-thread form (+ a direct check of
# that geometry against the generic-free oracle path: counts of the two forms must agree)
export TMPDIR=/tmp
python - <<'PY'
import sys
sys.path.insert(0, '.')
import numpy as np
from pyphysim_amd import _lib
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
for dt in ("f32", "f64"):
    eng = Engine(0, dt)
    eng.set_constellation(constellation("qam", 16), _lib.CONST_QAM)
    out = {}
    for thr in (0, 512, 1024):
        with eng.options(f64_threads=thr):
            res, se, be = eng.run_mimo_ofdm(2, 4, 2048, 16, 2048, 1, 10 ** -1.6, 77, 1000, 3000, method=_lib.DEMOD_MINDIST, dtype=dt, per_realization=True)
        out[thr] = (res["sym_errors"], res["bit_errors"], se.copy())
        print(dt, thr, res["sym_errors"], res["bit_errors"])
    a, b = out[512][2].astype(np.int64), out[1024][2].astype(np.int64)
    print(dt, "max per-realization difference between the forms:", int(np.max(np.abs(a - b))), "default equals",
          "512" if np.array_equal(out[0][2], out[512][2]) else ("1024" if np.array_equal(out[0][2], out[1024][2]) else "neither"))
PY
timeout 900 python -m pytest tests/test_gpu_planar_f32.py tests/test_gpu_f64_kernel.py tests/test_gpu_fuzz.py tests/test_gpu_oracle_depth.py -q --timeout=600 2>&1 | tail -3
