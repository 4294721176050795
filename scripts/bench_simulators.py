#!/usr/bin/env python3
"""End-to-end rate through the SimulationRunner surface (simulate() incl. Python bookkeeping, counter read-back and
Result construction) next to the bare kernel rate of bench.py: realizations/s per simulator, one SNR point."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import simulators  # noqa: E402

CASES = [
    ("AwgnSimulator (config 1)", lambda: simulators.AwgnSimulator(SNR=[10.0], M=16, NSymbs=10000, rep_max=1 << 21,
                                                                batch_size=1 << 19)),
    ("MimoOfdmSimulator (config 4)", lambda: simulators.MimoOfdmSimulator(SNR=[25.0], M=64, rep_max=1 << 21, batch_size=65536)),
    ("OfdmTdlSimulator (config 3)", lambda: simulators.OfdmTdlSimulator(SNR=[20.0], rep_max=1 << 22, batch_size=131072)),
    ("FlatFadingSimulator (config 2)", lambda: simulators.FlatFadingSimulator(SNR=[20.0], M=64, rep_max=1 << 15, batch_size=4096)),
    ("IaSimulator (config 5)", lambda: simulators.IaSimulator(SNR=[20.0], rep_max=1 << 23, batch_size=262144)),
    ("MimoOfdmTdlSimulator (8f.1)", lambda: simulators.MimoOfdmTdlSimulator(
        SNR=[25.0], M=64, Nt=4, Nr=4, fft_size=1024, cp_size=16, num_ofdm_symbols=1, Fd=10.0, Ts=1.0 / (15e3 * 1024),
        tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0), tap_delays=[i / (15e3 * 1024) for i in range(5)],
        rep_max=98304 * 8, batch_size=98304)),
    ("BdSimulator (8f.3, comp_BD app)", lambda: simulators.BdSimulator(SNR=[15.0], noise_var=1.0, rep_max=1 << 22,
                                                                     batch_size=131072)),
    ("MimoSimulator blast 4x4", lambda: simulators.MimoSimulator(SNR=[15.0], scheme="blast", Nt=4, Nr=4, rep_max=1 << 23,
                                                               batch_size=262144)),
]
out = {}
for name, make in CASES:
    sim = make()
    sim.rep_max = min(sim.rep_max, 4 * sim.batch_size)
    sim.simulate()                      # warm-up (library load, twiddles, first launches)
    sim = make()
    t0 = time.perf_counter()
    sim.simulate()
    dt = time.perf_counter() - t0
    out[name] = dict(realizations_per_s=sim.rep_max / dt, seconds=dt, rep_max=sim.rep_max,
                     ser=sim.results.get_result_values_list("ser")[0])
print(json.dumps(out, indent=1))
