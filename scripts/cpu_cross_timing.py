#!/usr/bin/env python3
"""CONTAINER-ONLY: time the reference's own chains (imported read-only from /root/reference through
oracle/make_golden.py's runners) against the NumPy oracle port (oracle/chains.py) on the same seeds, one thread.

BASELINE.md section 3.1 asks for this one-off cross-timing: bench.py's cpu_baseline leg times the port on the GPU
box (the reference cannot travel), so the port must not be a faster or slower stand-in.  Writes
profiles/cpu_cross_timing.json; bench.py quotes its `summary` next to cpu_baseline.
"""
import json
import os
import sys
import time

for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)
import make_golden as mg  # noqa: E402  (imports the reference)
from oracle import chains  # noqa: E402

CASES = {
    "c4": ("c4_mimo_ofdm", dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1,
                                snr_db=25.0, mmse=True), 300),
    "c3": ("c3_ofdm_tdl", dict(mod="qpsk", M=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=20.0,
                               Fd=10.0, Ts=1.0 / (15e3 * 1024), L=8, tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0),
                               tap_delays_samples=(0, 1, 2, 3, 4)), 60),
    "c2": ("c2_flat_jakes", dict(mod="qam", M=64, N=100000, snr_db=20.0, Fd=100.0, Ts=1e-3, L=8), 8),
    "c1": ("c1_awgn", dict(mod="qam", M=16, N=10000, snr_db=10.0), 300),
}


def rate(fn, n):
    fn(0)
    t0 = time.perf_counter()
    for r in range(1, n + 1):
        fn(r)
    return n / (time.perf_counter() - t0)


out = {}
for cfg, (name, kw, n) in CASES.items():
    ref = rate(lambda r: mg.run_ref(name, kw, 1000 + r), n)
    port = rate(lambda r: mg.ORACLE[name](chains.LegacyRng(1000 + r), **kw), n)
    port_philox = rate(lambda r: mg.ORACLE[name](chains.PhiloxRng(20260927, r), **kw), n)
    out[cfg] = {"reference_realizations_per_s": ref, "port_legacy_rng_realizations_per_s": port,
                "port_philox_rng_realizations_per_s": port_philox, "port_over_reference": port_philox / ref,
                "realizations_timed": n}
    print(cfg, out[cfg], flush=True)
c4 = out["c4"]
doc = {"host": "build container (no GPU), 1 thread", "chains": out,
       "summary": "reference %.0f /s vs port %.0f /s per core on config 4 in the build container (ratio %.2f; the "
                  "port draws from NumPy Philox instead of MT19937)" % (
                      c4["reference_realizations_per_s"], c4["port_philox_rng_realizations_per_s"],
                      c4["port_over_reference"])}
json.dump(doc, open(os.path.join(REPO, "profiles", "cpu_cross_timing.json"), "w"), indent=1)
print(doc["summary"])
