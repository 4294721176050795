import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from pyphysim_amd.engine import Engine
from pyphysim_amd import _lib
from oracle import chains
import test_gpu_tdl_wave as t
eng=Engine(0,"f64")
kw=dict(t.CASES[3]); mod,M=kw.pop("mod"),kw.pop("M")
eng.set_constellation(chains.constellation(mod,M), _lib.CONST_GENERIC)
for dt in ("f32","f64"):
    for wave in (1,0):
        try:
            r=t._run(eng, 5, 8, dt, wave=wave, **kw)
            print(dt, wave, "ok", r[0]["sym_errors"])
        except Exception as e:
            print(dt, wave, "ERR", str(e)[:200])
