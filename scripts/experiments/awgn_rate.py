import time, torch
from pyphysim_amd.engine import Engine
from pyphysim_amd import _lib
from pyphysim_amd.modulators import constellation
eng = Engine()
eng.set_constellation(constellation("qam", 16), _lib.CONST_QAM)
for name, f in (("awgn 1e4 slicer", lambda a, n: eng.run_awgn(10000, 0.1, 1, a, n, method=_lib.DEMOD_QAM_SLICER, dtype="f32")),
                ("awgn 1e4 mindist", lambda a, n: eng.run_awgn(10000, 0.1, 1, a, n, dtype="f32")),
                ("rayleigh 1e4 slicer", lambda a, n: eng.run_flat_fading(10000, 0.1, 1, a, n, rayleigh_iid=True, L=0, method=_lib.DEMOD_QAM_SLICER, dtype="f32"))):
    n = 1 << 18
    for i in range(3): f(i * n, n)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(5): r = f((3 + i) * n, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(name, "%.3g realizations/s" % (n / dt), r["sym_errors"] / (n * 10000))
