import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
from pyphysim_amd import _lib
from pyphysim_amd.channels import discretize_profile
eng=Engine(0,"f32"); eng.set_constellation(constellation("qpsk",4), _lib.CONST_GENERIC)
Ts=1/(15e3*1024); p,d=discretize_profile(np.array([0.,-3,-6,-9,-12]), np.arange(5)*Ts, Ts)
cnt=eng.new_counters()
for n in (419428, 104860, 262144):
    for o in (0,8,0,8,4,2,1):
        with eng.options(grid_oversub=o):
            run=lambda f: eng.run_ofdm_tdl(1024,16,1024,1,0.01,p,d,1,f,n,Fd=10.0,Ts=Ts,L=8,dtype="f32",counters=cnt)
            run(1<<30); eng.sync(); eng.timer_start()
            for s in range(5): run(s*n)
            ms=eng.timer_stop_ms()/5
        print(n, 'oversub', o, '%.3f ms' % ms, '%.4g /s' % (n/ms*1e3))
