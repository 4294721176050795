#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ia_general.py tests/test_gpu_pipelines.py tests/test_gpu_slices.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -12
# the 2048 x 4x4 frequency-selective kernel before / after (one workgroup per CU either way)
python - <<'PY'
import numpy as np, json
from pyphysim_amd import _lib
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
from pyphysim_amd.channels import discretize_profile
eng = Engine(0, "f64")
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
Ts = 1.0 / (15e3 * 2048)
p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
for dt in ("f64", "f32"):
    n = 8192
    cnt = eng.new_counters()
    run = lambda first: eng.run_mimo_ofdm_tdl(4, 4, 2048, 16, 2048, 1, 10 ** -2.5, p_lin, d_idx, 1, first, n, Fd=10.0, Ts=Ts, L=8, dtype=dt, counters=cnt)
    run(1 << 30); eng.sync(); eng.timer_start()
    for s in range(3): run(s * n)
    ms = eng.timer_stop_ms() / 3
    c = eng.read_counters(cnt)
    print("f1 2048 4x4", dt, "%.4g /s" % (n / ms * 1e3), "%.3f ms" % ms, "ser %.5f" % (c["sym_errors"] / (c["n_realizations"] * 8192.0)))
PY
