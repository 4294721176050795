import time, json, sys
sys.path.insert(0, "/root/repo")
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
from pyphysim_amd import _lib
eng = Engine(0, "f32")
eng.set_constellation(constellation("qam", 16), _lib.CONST_QAM)
out = {}
for solver in ("closed_form", "alt_min", "min_leakage", "max_sinr"):
    n = 262144
    eng.run_ia(200, 0.01, 1, 1 << 30, n, method=_lib.DEMOD_QAM_SLICER, solver=solver, max_iterations=60)
    eng.sync(); t0 = time.perf_counter()
    r = eng.run_ia(200, 0.01, 1, 0, n, method=_lib.DEMOD_QAM_SLICER, solver=solver, max_iterations=60)
    eng.sync(); dt = time.perf_counter() - t0
    out[solver] = dict(realizations_per_s=n / dt, ser=r["sym_errors"] / (n * 600.0), mean_iterations=r["ia_runned_iterations"] / n,
                       mean_sum_capacity=r["sum_capacity"] / n)
print(json.dumps(out))
