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

# round 6: the complex128 kernels of the round (full-wave / part-wave family, packed walks) and the complex64 packed walks --
# awkward counts, and split invariance: the counters of [0, n) equal the sums over [0, k) and [k, n) for ragged k
def same_sum(fn, n, k):
    a, b, c = fn(0, n), fn(0, k), fn(k, n - k)
    for key in ("sym_errors", "bit_errors", "sym_errors_sq", "bit_errors_sq", "n_realizations", "n_skipped"):
        assert a[key] == b[key] + c[key], (key, a[key], b[key], c[key])
    assert a["n_realizations"] + a["n_skipped"] == n
t = time.time()
for dt in ("f64", "f32"):
    for fft, nt in ((256, 4), (256, 2), (512, 4), (1024, 4), (2048, 4)):
        n = 40961 if fft < 2048 else 8195
        for k in (1, 7, n // 3, n - 1):
            same_sum(lambda f, c: eng.run_mimo_ofdm(nt, nt, fft, 16, fft, 1, 0.003, 9, f, c, method=_lib.DEMOD_MINDIST, dtype=dt), n, k)
    for ns in (128, 130, 200, 1000):
        for k in (1, 15, 17, 5000):
            same_sum(lambda f, c: eng.run_ia(ns, 0.01, 9, f, c, dtype=dt), 20011, k)
            same_sum(lambda f, c: eng.run_bd(3, 2, ns, 1.0, 0.03, 9, f, c, dtype=dt), 10007, k)
    for count in (1, 63, 65, 1000003):
        r = eng.run_ia(200, 0.01, 1, 5, count, dtype=dt); assert r["n_realizations"] + r["n_skipped"] == count
        r = eng.run_bd(3, 2, 500, 1.0, 0.03, 1, 5, count, dtype=dt); assert r["n_realizations"] + r["n_skipped"] == count
        r = eng.run_mimo_ofdm(4, 4, 256, 16, 256, 1, 0.003, 1, 5, count, dtype=dt); assert r["n_realizations"] + r["n_skipped"] == count
        r = eng.run_mimo_ofdm(2, 2, 256, 16, 256, 1, 0.003, 1, 5, count, dtype=dt); assert r["n_realizations"] + r["n_skipped"] == count
print("round-6 stress ok", time.time() - t)

# round 6, last day: the kernels and grid rules added then -- config 3 at 2048 with two wavefronts per realization (odd counts: the last
# pair's second slot is not a realization), delays beyond the prefix in the wavefront kernels, the small shapes whose grids give a
# workgroup hundreds of realizations, the parked-coefficient f1 kernels off 1024 points; awkward counts and split invariance
t = time.time()
for dt in ("f64", "f32"):
    for fft in (256, 512, 1024, 2048):
        Tsn = 1 / (15e3 * fft)
        pn, dn = discretize_profile(np.array([0., -3, -6, -9, -12]), np.arange(5) * Tsn, Tsn)
        eng.set_constellation(constellation("qpsk", 4), _lib.CONST_GENERIC)
        for n, k in ((1, 0), (2, 1), (3, 1), (40961, 7), (40961, 20480), (1000003 if fft <= 512 else 100003, 65)):
            if k:
                same_sum(lambda f, c: eng.run_ofdm_tdl(fft, 16, fft, 1, 0.01, pn, dn, 9, f, c, Fd=10.0, Ts=Tsn, L=8, dtype=dt), n, k)
            else:
                r = eng.run_ofdm_tdl(fft, 16, fft, 1, 0.01, pn, dn, 9, 0, n, Fd=10.0, Ts=Tsn, L=8, dtype=dt); assert r["n_realizations"] == n
        pi_, di_ = discretize_profile(np.array([0., -3, -6]), np.array([0, 7, 40]) * Tsn, Tsn)      # a delay beyond the prefix
        same_sum(lambda f, c: eng.run_ofdm_tdl(fft, 16, fft, 3, 0.01, pi_, di_, 9, f, c, Fd=10.0, Ts=Tsn, L=8, dtype=dt), 10007, 333)
        eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
        for nt, nr in ((1, 2), (2, 2), (2, 3), (2, 4), (4, 4)):
            n = 20011 if fft <= 512 else 4099
            same_sum(lambda f, c: eng.run_mimo_ofdm(nt, nr, fft, 16, fft, 1, 0.003, 9, f, c, method=_lib.DEMOD_MINDIST, dtype=dt), n, n // 3)
            same_sum(lambda f, c: eng.run_mimo_ofdm_tdl(nt, nr, fft, 16, fft, 1, 0.003, pn, dn, 9, f, c, Fd=10.0, Ts=Tsn, L=8, dtype=dt), min(n, 6007), 17)
        same_sum(lambda f, c: eng.run_mimo_ofdm_tdl(2, 3, fft, 4, fft, 2, 0.003, pi_, di_, 9, f, c, Fd=10.0, Ts=Tsn, L=8, dtype=dt), 2003, 1000)
print("last-day stress ok", time.time() - t)
