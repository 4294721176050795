#!/usr/bin/env python3
"""HBM throughput of the per-operator (staged) kernels on device-resident arrays: algorithmic bytes
(inputs read once + outputs written once) / HIP-event time.  Run on the GPU box:
    python scripts/bench_operators.py > gpurun_out/operators.json
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

DT = "f64" if "--dtype=f64" in sys.argv or os.environ.get("BENCH_OPS_DTYPE") == "f64" else "f32"
CB = 16 if DT == "f64" else 8     # bytes per complex sample
eng = Engine(0, DT)
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
N = (1 << 25) if DT == "f64" else (1 << 26)   # 512 MiB of samples either way, past the 256 MiB Infinity Cache
REP = 5
rows = []


def timed(name, fn, nbytes):
    fn()
    eng.sync()
    eng.timer_start()
    for _ in range(REP):
        fn()
    ms = eng.timer_stop_ms() / REP
    rows.append({"kernel": name, "ms": ms, "algorithmic_GB": nbytes / 1e9, "GBps": nbytes / ms / 1e6,
                 "frac_of_8TBps": nbytes / ms / 1e6 / 8000.0})


ONLY = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]


def jakes_rows(cases):
    """The Jakes generator alone: S streams of N / S samples, L rays, written once (no input but 2 L S phases)."""
    rs = np.random.RandomState(5)
    for L, S in cases:
        n = N // S
        phi, psi = rs.uniform(0, 2 * np.pi, (L, S)), rs.uniform(0, 2 * np.pi, (L, S))
        timed("jakes_generate L=%d, %d streams (-> %d B)" % (L, S, CB),
              lambda: eng.jakes_generate(phi, psi, 100.0, 1e-3, 1e-3, n, device=True), N * CB)


if ONLY == ["jakes"]:
    jakes_rows(((8, 16), (16, 16), (12, 16)))
    print(json.dumps({"device": eng.device_name, "dtype": DT, "bytes_per_sample": CB, "n_symbols": N, "rows": rows}, indent=1))
    sys.exit(0)
idx = eng.rand_symbols(N, 64, 1, 2, device=True)
tx = eng.modulate(idx)
noise = eng.randn_c(N, 1, 2, device=True)
rx = eng.awgn_add(tx, noise, 0.01)
cnt = eng.new_counters()          # device-resident: the counting kernels are timed without a read-back
timed("modulate (4 B idx -> 8 B sample)", lambda: eng.modulate(idx), N * (4 + CB))
timed("awgn_add (8+8 -> 8)", lambda: eng.awgn_add(tx, noise, 0.01), N * 3 * CB)
timed("demodulate slicer (8 -> 4)", lambda: eng.demodulate(rx, method=_lib.DEMOD_QAM_SLICER), N * (4 + CB))
timed("demodulate mindist M=64 (8 -> 4)", lambda: eng.demodulate(rx), N * (4 + CB))
timed("demod_count slicer (8 + 4 -> counters)", lambda: eng.demod_count(rx, idx, n_real=1024, method=_lib.DEMOD_QAM_SLICER, counters=cnt), N * (4 + CB))
dec = eng.demodulate(rx, method=_lib.DEMOD_QAM_SLICER)
timed("count_errors (4 + 4 -> counters)", lambda: eng.count_errors(idx, dec, 6, n_real=1024, counters=cnt), N * 8)
timed("cdiv (8+8 -> 8)", lambda: eng.cdiv(rx, tx), N * 3 * CB)
timed("randn_c Philox (-> 8)", lambda: eng.randn_c(N, 1, 2, device=True), N * CB)
jakes_rows(((8, 16), (16, 16)))
nsym = N // 1024
timed("ofdm_modulate 1024+16 (8 -> 8.1)", lambda: eng.ofdm_modulate(tx, 1024, 16, 1024), N * CB + nsym * 1040 * CB)
t = eng.ofdm_modulate(tx, 1024, 16, 1024)
timed("ofdm_demodulate 1024+16", lambda: eng.ofdm_demodulate(t, 1024, 16, 1024), N * CB + nsym * 1040 * CB)
b = 4096
ns = N // (4 * b)
X = tx.reshape(b, 4, ns)
H = eng.randn_c(b * 16, 3, 4, device=True).reshape(b, 4, 4)
timed("mimo_channel 4x4 (H X)", lambda: eng.mimo_channel(H, X), b * 4 * ns * 2 * CB)
G, _ = eng.blast_filter(H, 0.01)
timed("blast_decode 4x4 (G Y)", lambda: eng.blast_decode(G, X), b * 4 * ns * 2 * CB)
timed("blast_encode (transpose)", lambda: eng.blast_encode(tx, 4, batch=b), N * 2 * CB)
print(json.dumps({"device": eng.device_name, "dtype": DT, "bytes_per_sample": CB, "n_symbols": N, "rows": rows}, indent=1))
