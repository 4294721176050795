#!/usr/bin/env python3
"""Every fused pipeline with awkward realization counts (1, 63, 65, 1e6+3, 2^20), and 300 repeated launches
that must return the identical counter block."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
from pyphysim_amd import _lib
from pyphysim_amd.channels import discretize_profile
eng=Engine(0,"f32")
eng.set_constellation(constellation("qam",64), _lib.CONST_QAM)
Ts=1/(15e3*1024); p,d=discretize_profile(np.array([0.,-3,-6,-9,-12]), np.arange(5)*Ts, Ts)
t=time.time()
for count in (1, 63, 65, 1000003, 1<<20):
    r=eng.run_mimo_ofdm(4,4,1024,16,1024,1,0.003,1,5,count,method=_lib.DEMOD_QAM_SLICER); assert r["n_realizations"]+r["n_skipped"]==count, (count,r)
    r=eng.run_ofdm_tdl(1024,16,1024,1,0.01,p,d,1,5,count,Fd=10.0,Ts=Ts,L=8); assert r["n_realizations"]==count
    r=eng.run_ia(200,0.01,1,5,count); assert r["n_realizations"]+r["n_skipped"]==count
    r=eng.run_bd(3,2,500,1.0,0.03,1,5,count); assert r["n_realizations"]+r["n_skipped"]==count
    r=eng.run_mimo_flat("blast",4,4,200,0.03,1,5,count); assert r["n_realizations"]+r["n_skipped"]==count
    r=eng.run_flat_fading(100000 if count<100 else 1000,0.01,1,5,count); assert r["n_realizations"]==count
    r=eng.run_mimo_ofdm_tdl(4,4,1024,16,1024,1,0.003,p,d,1,5,min(count,200000),Fd=10.0,Ts=Ts,L=8); assert r["n_realizations"]+r["n_skipped"]==min(count,200000)
print("stress ok", time.time()-t)
# repeat many launches to look for leaks / drift
import resource
a=eng.run_mimo_ofdm(4,4,1024,16,1024,1,0.003,1,0,4096)
for i in range(300):
    b=eng.run_mimo_ofdm(4,4,1024,16,1024,1,0.003,1,0,4096)
    assert a==b
print("repeat ok", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)
