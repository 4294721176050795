#!/bin/bash
# round 4, call 20: radix-16 kernel at Nt < 4 (1024, Nr = 4) -- parity and rate against the radix-4 family form (f64_threads=512)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f64_kernel.py -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|Error" | tail -5
python - <<'PY'
from pyphysim_amd import _lib
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
eng = Engine(0, "f64")
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
nv = 10 ** -2.5
for nt in (1, 2, 3):
    for thr in (0, 512):
        n = 262144
        cnt = eng.new_counters()
        with eng.options(f64_threads=thr):
            run = lambda first: eng.run_mimo_ofdm(nt, 4, 1024, 16, 1024, 1, nv, 1, first, n, dtype="f64", counters=cnt)
            run(1 << 30); eng.sync(); eng.timer_start()
            for s in range(3): run(s * n)
            ms = eng.timer_stop_ms() / 3
        c = eng.read_counters(cnt)
        print("1024 %dx4" % nt, "radix-16" if thr == 0 else "radix-4 ", "%.4g /s" % (n / ms * 1e3), "%.3f ms" % ms, "ser %.5f" % (c["sym_errors"] / (c["n_realizations"] * 1024.0 * nt)))
PY
