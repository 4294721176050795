"""round 5: what the sector certificate buys -- 8-PSK / 16-PSK through config 3's link (OFDM 1024 over the 5-tap TDL channel) and
the AWGN link, with the certificate (default) and without (option demod_nocert: candidate grid), both arithmetics"""
import sys; sys.path.insert(0, '.')
import numpy as np
from pyphysim_amd.engine import Engine
from pyphysim_amd.modulators import constellation
from pyphysim_amd.channels import discretize_profile
from pyphysim_amd import _lib
eng = Engine(0, "f64")
Ts = 1.0 / (15e3 * 1024)
p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
for M in (8, 16):
    eng.set_constellation(constellation("psk", M), _lib.CONST_GENERIC)
    for dt in ("f64", "f32"):
        for nocert in (1, 0):
            with eng.options(demod_nocert=nocert):
                n = 262144
                c = eng.new_counters()
                eng.run_ofdm_tdl(1024, 16, 1024, 1, 0.01, p_lin, d_idx, 5, 0, n, Fd=10.0, Ts=Ts, L=8, dtype=dt, counters=c); eng.sync()
                eng.timer_start()
                for s in range(4):
                    eng.run_ofdm_tdl(1024, 16, 1024, 1, 0.01, p_lin, d_idx, 5, (s + 1) * n, n, Fd=10.0, Ts=Ts, L=8, dtype=dt, counters=c)
                ms = eng.timer_stop_ms() / 4
                r = eng.read_counters(c)
                m2 = 1 << 22
                c2 = eng.new_counters()
                eng.run_awgn(1000, 0.02, 5, 0, 4096, dtype=dt, counters=c2); eng.sync()
                eng.timer_start()
                for s in range(4):
                    eng.run_awgn(10000, 0.02, 5, s * 65536, 65536, dtype=dt, counters=c2)
                ms2 = eng.timer_stop_ms() / 4
                print("%d-PSK %s nocert=%d: OFDM-TDL %.4g realizations/s (ser %.5f) | AWGN 1e4 symbols %.4g realizations/s" % (
                    M, dt, nocert, n / ms * 1e3, r["sym_errors"] / (r["n_realizations"] * 1024.0), 65536 / ms2 * 1e3))
