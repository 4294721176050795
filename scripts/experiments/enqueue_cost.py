# host-side cost of enqueuing one config-4 step (two launches through ctypes) vs the step itself; run on the GPU box
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pyphysim_amd import _lib
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
eng = Engine(0, "f32")
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
cnt = eng.new_counters()
run = lambda first: eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, 0.003, 1, first, 65536, mmse=True, method=_lib.DEMOD_QAM_SLICER, dtype="f32", counters=cnt)
for i in range(100): run(i * 65536)
eng.sync()
t0 = time.perf_counter()
for i in range(50): run(i * 65536)
t1 = time.perf_counter()
eng.sync()
t2 = time.perf_counter()
print("enqueue %.1f us per step, total %.3f ms per step" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e3))
