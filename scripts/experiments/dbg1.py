import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from pyphysim_amd.engine import Engine
import test_gpu_mimo_tdl_wave as t
eng=Engine(0,"f64")
t._set(eng,"qam",64)
for kw in (dict(), dict(nt=2,nr=3,fft_size=512,Ts=1e-6)):
    for k in (0,1,2):
        try:
            res,se,be=t._run(eng,7,16,"f64",kernel=k,snr_db=300.0,**kw)
            print(kw,k,res["sym_errors"],se)
        except Exception as e: print(kw,k,"ERR",str(e)[:80])
    w=t._oracle(7,16,"qam",64,snr_db=300.0,**kw)
    print("oracle",w[0])
